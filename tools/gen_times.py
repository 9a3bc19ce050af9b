#!/usr/bin/env python3
"""Per-generation cost of a workload: trace with generation_limit = 1..G and difference the
GPU time of the generation kernels (prt_trace_stats).  usage: gen_times.py [workload] [rays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import scenes
from pyrayt_amd import engine

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
api = scenes.product_api()
comps, rays = scenes.SCENES[name](api, n)
dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
ds = engine.DeviceScene.from_components(comps)
print(name, ds.info())
prev = 0.0
for limit in range(1, 11):
    best = 1e9
    for _ in range(5):
        rows, counts = ds.trace(dev, limit)
        st = ds.trace_stats()
        best = min(best, st["kernel_ms"])
    live = counts[-1] if counts else 0
    print(f"limit {limit}: kernel {best:.4f} ms (+{best - prev:.4f})  rows/gen {counts}")
    if len(counts) < limit or live == 0:
        break
    prev = best
