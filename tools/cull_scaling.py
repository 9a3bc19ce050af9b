#!/usr/bin/env python3
"""Does the component cull step pay?  prt_propagate (k_hit) and a 1-generation prt_trace over a
train of N biconvex lenses + a detector, 1M rays starting in front of the first lens.
Run twice: as is, and with PRT_NO_CULL=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot

n = 1_000_000
rays = scenes.cone_rays(n, (-3.0, 0.0, 0.0), 3.0, 5)
rays_dev = torch.from_numpy(rays).cuda()
for count in (1, 2, 4, 8, 16, 32):
    parts = [pyrayt.components.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in range(count)]
    parts.append(pyrayt.components.baffle((2, 2)).move_x(1.0 * count + 1))
    ds = engine.DeviceScene(SceneSnapshot(parts))
    for _ in range(3): ds.propagate(rays_dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ds.propagate(rays_dev)
    e1.record(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        ds.trace(rays_dev, 1)
        best = min(best, ds.trace_stats()["kernel_ms"])
    full = 1e9
    for _ in range(3):
        rows, counts = ds.trace(rays_dev, 4 * count + 4)
        full = min(full, ds.trace_stats()["kernel_ms"])
    info = ds.info()
    print(f"{count:3d} lenses ({info['primitives']:3d} prims, {info['cull_steps']:2d} cull steps): propagate "
          f"{e0.elapsed_time(e1) / 10 * 1000:7.1f} us, generation 0 {best * 1000:7.1f} us, whole trace "
          f"{full:8.3f} ms for {sum(counts)} rows")
    ds.close()
