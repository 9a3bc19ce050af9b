#!/usr/bin/env python3
"""Component cull steps and their hierarchy: prt_propagate (k_hit), a 1-generation prt_trace and the whole
trace over a train of N biconvex lenses + a detector, 1M rays starting in front of the first lens.

  default            cull step per component + hierarchy of group steps (from 8 components on); the groups are
                     formed by position when the list order is not already tight (prt_scene.hpp)
  list_order_groups  the hierarchy over runs of consecutive components (the round-2 form)
  no_groups          one cull step per component, no hierarchy
  no_cull            no cull steps

usage (GPU box): python tools/cull_scaling.py [--shuffle] [--counts 2 8 32] > profiles/r3/cull_scaling.txt
--shuffle lists the lenses in random order (the detector somewhere in between) and traces the beam from
both ends."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot

ap = argparse.ArgumentParser()
ap.add_argument("--shuffle", action="store_true")
ap.add_argument("--counts", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32])
ap.add_argument("--rays", type=int, default=1_000_000)
args = ap.parse_args()

n = args.rays
forward = scenes.cone_rays(n, (-3.0, 0.0, 0.0), 3.0, 5)
beams = [("from the front", forward)]
print(f"# {n} rays, lens train {'listed in random order' if args.shuffle else 'listed along the axis'}")
print(f"{'lenses':>6s} {'beam':>15s} {'form':>18s} {'prims':>5s} {'cull steps':>10s} {'by position':>11s} {'propagate us':>12s} "
      f"{'generation 0 us':>15s} {'whole trace ms':>14s} {'rows':>10s}")
for count in args.counts:
    order = np.random.default_rng(100 + count).permutation(count) if args.shuffle else np.arange(count)
    if args.shuffle:
        back = scenes.cone_rays(n, (count + 0.5, 0.0, 0.0), 3.0, 6)
        back[4] *= -1.0
        beams = [("from the front", forward), ("from the back", back)]
    for beam_name, rays in beams:
        rays_dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
        reference = None
        for form in ("default", "list_order_groups", "no_groups", "no_cull"):
            if form == "list_order_groups" and not (args.shuffle and count >= 8):
                continue
            if form == "no_groups" and count < 8:
                continue
            CountedObject.reset_ids()
            parts = [pyrayt.components.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in order]
            detector = pyrayt.components.baffle((2, 2)).move_x(1.0 * count + 1) if not args.shuffle else \
                pyrayt.components.baffle((2, 2)).move_x(-4.0)
            parts.insert(len(parts) // 3 if args.shuffle else len(parts), detector)
            if args.shuffle:  # a second detector behind the far end: both beams end on one
                parts.insert(2 * len(parts) // 3, pyrayt.components.baffle((2, 2)).move_x(1.0 * count + 1))
            ds = engine.DeviceScene(SceneSnapshot(parts), options={} if form == "default" else {form: 1})
            for _ in range(3): t, surf = ds.propagate(rays_dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): t, surf = ds.propagate(rays_dev)
            e1.record(); torch.cuda.synchronize()
            if reference is None:
                reference = surf.clone()
            same = bool(torch.equal(surf, reference))
            best = 1e9
            for _ in range(5):
                ds.trace(rays_dev, 1)
                best = min(best, ds.trace_stats()["kernel_ms"])
            full = 1e9
            for _ in range(3):
                rows, counts = ds.trace(rays_dev, 4 * count + 4)
                full = min(full, ds.trace_stats()["kernel_ms"])
            info = ds.info()
            print(f"{count:6d} {beam_name:>15s} {form:>18s} {info['primitives']:5d} {info['cull_steps']:10d} "
                  f"{info['spatial_groups']:11d} {e0.elapsed_time(e1) / 10 * 1000:12.1f} {best * 1000:15.1f} {full:14.3f} "
                  f"{sum(counts):10d}{'' if same else '  IDS DIFFER'}", flush=True)
            ds.close()
