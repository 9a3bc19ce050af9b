#!/usr/bin/env python3
"""Target for rocprofv3 counter passes: prt_propagate over scenes of 1, 2, 4 bare planes and the
lens (+ plane), to separate the fixed cost of the hit kernel from the per-step cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot

_, rays = scenes.config2(scenes.product_api(), 1_000_000)
dev = torch.from_numpy(rays).cuda()
def det(k): return pyrayt.components.baffle((1, 1)).move_x(1 + k)
def lens(): return pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
cases = [[det(0)], [det(0), det(1)], [det(k) for k in range(4)], [lens()], [lens(), det(0)]]
for parts in cases:
    ds = engine.DeviceScene(SceneSnapshot(parts))
    for _ in range(3):
        ds.propagate(dev)
    torch.cuda.synchronize()
    ds.close()
