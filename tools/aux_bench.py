#!/usr/bin/env python3
"""The kernels either side of the trace against the HBM roofline: device-side sources (prt_generate_rays:
104 B written per ray) and the frame placement kernel (prt_place_rows: 120 B read + 120 B written per row)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import pyrayt_amd as pyrayt
from pyrayt_amd import distributed as pdist, engine


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dev = torch.device("cuda", 0)
for name, src in (("ConeOfRays", pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)),
                  ("LineOfRays", pyrayt.components.LineOfRays(spacing=1)),
                  ("Lamp", pyrayt.components.Lamp(1, 1, 20))):
    for n in (1_000_000, 8_000_000):
        us = timed(lambda: engine.generate_rays([src], n, dev))
        print(f"prt_generate_rays {name:12s} {n:9d} rays: {us:8.1f} us  {n * 104 / us / 1e3:7.1f} GB/s written "
              f"(incl. the torch.empty of the ray set)")

lib = engine.library()
for world, per_rank in ((2, 1_500_000), (8, 375_000), (8, 3_000_000)):
    limit = 10
    matrix = torch.zeros((world, limit), dtype=torch.int64)
    matrix[:, :3] = per_rank // 3
    widest = int(matrix.sum(dim=1).max())
    _, _, total = pdist.placement(matrix)
    blocks = torch.rand((world, 15, widest), dtype=torch.float64, device=dev)
    us = timed(lambda: pdist._place_on_device(blocks, matrix, limit, total))
    print(f"prt_place_rows {world} ranks x {widest} rows -> {total} rows: {us:8.1f} us  "
          f"{total * 240 / us / 1e3:7.1f} GB/s read + written (incl. output allocation and the stream sync of the wrapper)")
