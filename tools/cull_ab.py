import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot
n = 1_000_000
rays = torch.from_numpy(scenes.cone_rays(n, (-3.0, 0.0, 0.0), 3.0, 5)).cuda()
for opts in ({}, {"one_direction": 1}):
    parts = [pyrayt.components.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in range(32)]
    parts.append(pyrayt.components.baffle((2, 2)).move_x(33.0))
    ds = engine.DeviceScene(SceneSnapshot(parts), options=opts)
    for _ in range(3): ds.propagate(rays)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ds.propagate(rays)
    e1.record(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(8):
        ds.trace(rays, 1); best = min(best, ds.trace_stats()["kernel_ms"])
    full = 1e9
    for _ in range(4):
        ds.trace(rays, 140); full = min(full, ds.trace_stats()["kernel_ms"])
    print(os.path.basename(os.environ.get("PRT_LIB", "libprt_hip.so")), opts, f"propagate {e0.elapsed_time(e1)/20*1000:.1f} us  gen0 {best*1000:.1f} us  trace {full:.3f} ms", flush=True)
