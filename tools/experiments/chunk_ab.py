#!/usr/bin/env python3
"""Chunked traces against one-chain traces (scene option chunks = 0 / 1), interleaved on one box: blocking prt_trace calls
of BASELINE config 2 with rotating ray sets -- what one RayTracer.trace_device() costs in a running design loop -- on the
null stream and on a stream of the caller's own, and the library's own loop with two traces in flight.

usage: tools/chunk_ab.py [--rays N] [--steps K] [--reps R]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import scenes  # noqa: E402
from pyrayt_amd import engine  # noqa: E402
from pyrayt_amd.g3d.objects import CountedObject  # noqa: E402
from pyrayt_amd.scene import SceneSnapshot  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workload", default="config2")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sets = []
    for seed in range(4):
        CountedObject.reset_ids()
        base = {"config2": 1234, "config3": 7, "config5": 11}.get(args.workload, 0)
        if args.workload == "config4":
            parts, rays = scenes.config4(scenes.product_api(), args.rays // 8)
        else:
            parts, rays = scenes.SCENES[args.workload](scenes.product_api(), args.rays, seed=base + seed)
        sets.append(torch.from_numpy(np.ascontiguousarray(rays)).to(dev))
    n = sets[0].shape[1]
    limit = 10
    block = torch.empty((15, n * limit), dtype=torch.float64, device=dev)
    block2 = torch.empty_like(block)
    side = torch.cuda.Stream(dev)
    scenes_by = {label: engine.DeviceScene(SceneSnapshot(parts), options={"chunks": value})
                 for label, value in (("one chain", 1), ("two chunks", 0))}
    results = {}
    for rep in range(args.reps):
        for label, scene in scenes_by.items():
            for where in ("null stream", "own stream"):
                stream = side if where == "own stream" else torch.cuda.current_stream(dev)
                with torch.cuda.stream(stream):
                    for k in range(20):
                        scene.trace(sets[k % 4], limit, out=block)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for k in range(args.steps):
                        rows, counts = scene.trace(sets[k % 4], limit, out=block)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / args.steps * 1e3
                results.setdefault((label, "blocking, " + where), []).append((ms, scene.trace_stats()["variant"]))
            batch = engine.TraceBatch(scene, [sets[k % 4] for k in range(args.steps)], limit, depth=2, outs=[block, block2],
                                      flags=engine.TRACE_NO_TIMING)
            batch.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch.run()
            torch.cuda.synchronize()
            results.setdefault((label, "two traces in flight"), []).append(((time.perf_counter() - t0) / args.steps * 1e3, 0))
    rows = int(sum(counts))
    for (label, how), vals in sorted(results.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        best = min(v[0] for v in vals)
        print(f"{how:28s} {label:12s} " + "  ".join(f"{v[0]:.4f}" for v in vals) + f"  ms per trace   best {rows / best / 1e-3:.4e} rows/s"
              + (f"   (variant {vals[-1][1]})" if vals[-1][1] else ""))


if __name__ == "__main__":
    main()
