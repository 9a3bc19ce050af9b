#!/usr/bin/env python3
"""What lean segments cost a ray set that does not qualify (ids in random order: every wave hands on its rows, and reads them
after the head of its segment has told it to): config 2, 1M rays, synchronous traces, ms per trace.
    PRT_LIB=... python3 tools/experiments/nonlean_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot

parts, rays = scenes.config2(scenes.product_api(), 1_000_000)
mixed = rays.copy()
mixed[12] = np.random.default_rng(1).permutation(rays.shape[1])
ds = engine.DeviceScene(SceneSnapshot(parts))
out = torch.empty((15, 10_000_000), dtype=torch.float64, device="cuda")
for label, block in (("ids counting up (lean)", rays), ("ids shuffled (rows)", mixed)):
    dev = torch.from_numpy(block).cuda()
    for _ in range(30):
        ds.trace(dev, 10, out=out)
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            ds.trace(dev, 10, out=out)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e3)
    print(f"{os.path.basename(engine.LIB_PATH):24s} {label:26s} ms per trace {min(best):.4f}  kernel ms {ds.trace_stats()['kernel_ms']:.4f}")
