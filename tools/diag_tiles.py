#!/usr/bin/env python3
"""Which launches of a repeated trace run dense / with their absorbed rays kept / with a look-back, and what each trace costs:
    python tools/diag_tiles.py [workload] [rays] [traces] [fresh|same] [flags]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from pyrayt_amd import engine
import torch

import scenes
from pyrayt_amd.scene import SceneSnapshot

workload = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
fresh = len(sys.argv) > 4 and sys.argv[4] == "fresh"  # every trace from another buffer (same rays)
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
parts, rays = getattr(scenes, workload)(scenes.product_api(), n)
ds = engine.DeviceScene(SceneSnapshot(parts))
dev = torch.from_numpy(rays).cuda()
keys = ("dense_launches", "sparse_keep_launches", "speculation_misses")
before = ds.telemetry()
for k in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fresh:
        dev = dev.clone()
    out = ds.trace(dev, 10, flags=flags)
    st = ds.trace_stats()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    now = ds.telemetry()
    print(k, "%.3f ms" % ms, "kernel %.3f ms" % st["kernel_ms"], out[1], {key: now[key] - before[key] for key in keys})
    before = now
