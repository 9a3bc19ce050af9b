#!/usr/bin/env python3
"""Time the fused render kernel (prt_render) on draw()-style pictures and, beside it, the numpy
oracle of the renderers on a bounded pixel sample of the same picture.

    python tools/render_bench.py [--width 640 2048 8192] [--view xy] [--cpu-pixels 200000]

Prints one JSON line per width: pixels, GPU ms per frame (torch events around `reps` launches on
the current stream), Mpixel/s, bytes written per pixel, and the oracle's pixel rate on 1 core."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, nargs="+", default=[640, 2048, 8192])
    ap.add_argument("--view", default="xy")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-pixels", type=int, default=200_000)
    args = ap.parse_args()

    import torch

    import helpers
    import scenes
    from oracle import render_oracle as ro
    from pyrayt_amd import engine
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    components, _ = scenes.config3(api, 8)
    device = torch.device("cuda", 0)
    ds = engine.DeviceScene.from_components(components)
    flat = helpers.flat_scene(SceneSnapshot(components))
    for width in args.width:
        camera, light, _ = api.cg.renderers.view_of(components, args.view, resolution=width)
        h, v = camera.get_resolution()
        ds.render(camera, device, light=light, keep_hits=True)  # warm up, uploads the scene
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(args.reps):
            rgba, t, surf = ds.render(camera, device, light=light, keep_hits=True)
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / args.reps
        # oracle on a contiguous band of rows (same work per pixel)
        rows = max(1, min(v, args.cpu_pixels // h))
        rays = engine.camera_rays(camera, device)[:, : rows * h].cpu().numpy().reshape(2, 4, -1)
        t0 = time.perf_counter()
        ot, osurf = ro.nearest_hits(flat, rays)
        ro.shaded_canvas(flat, ds.snapshot.gooch_table(), rays, ot, osurf, np.asarray(light, float), h, rows)
        cpu_s = time.perf_counter() - t0
        assert np.array_equal(osurf, surf[: rows * h].cpu().numpy())
        print(json.dumps({
            "picture": f"{h}x{v}", "pixels": h * v, "surfaces": int(len(flat["prim_type"])),
            "gpu_ms_per_frame": round(ms, 4), "gpu_mpixel_per_s": round(h * v / ms / 1e3, 1),
            "bytes_written_per_pixel": 48, "write_gb_per_s": round(48 * h * v / ms / 1e6, 1),
            "oracle_pixels": rows * h, "oracle_mpixel_per_s": round(rows * h / cpu_s / 1e6, 3),
            "hit_fraction": round(float((surf >= 0).float().mean()), 4),
        }))
    ds.close()


if __name__ == "__main__":
    main()
