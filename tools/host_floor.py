#!/usr/bin/env python3
"""Host-side floor of a trace: config 2 with so few rays that the kernels are pure latency."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot
for n in (256, 16384, 131072):
    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    dev = torch.from_numpy(rays).cuda()
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda")
    for _ in range(20): ds.trace(dev, 10, out=block)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); k = 0.0
    for _ in range(200):
        ds.trace(dev, 10, out=block); k += ds.trace_stats()["kernel_ms"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"n={n:7d}: step {dt*1e6:7.1f} us, generation kernels {k/200*1e3:7.1f} us, outside {dt*1e6 - k/200*1e3:6.1f} us")
    ds.close()
