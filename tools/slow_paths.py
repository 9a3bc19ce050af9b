#!/usr/bin/env python3
"""How many rays of each fixture take the exact (slow) paths behind the engine's shortcuts.
Needs the counting build: make -C pyrayt_amd/csrc libprt_hip_count.so ; run on the GPU box with
PRT_LIB=pyrayt_amd/csrc/libprt_hip_count.so python tools/slow_paths.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import helpers
from test_gpu_parity import device_scene
from pyrayt_amd import engine

lib = engine.library()
names = sys.argv[1:] or ["config2", "config3", "config4", "config5", "mirrors_and_stops", "stopped_lens",
                         "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "stale_box"]
out = (ctypes.c_ulonglong * 4)()
lib.prt_debug_slow_counters(out, 1)
print(f"{'fixture':18s} {'rays':>7s} {'node tests':>11s} {'exact box':>10s} {'cull tests':>11s} {'culled':>8s}")
for name in names:
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx))
    rays = torch.from_numpy(np.ascontiguousarray(fx["rays0"])).to("cuda:0")
    ds.trace(rays, int(fx["generation_limit"]))
    torch.cuda.synchronize()
    lib.prt_debug_slow_counters(out, 1)
    print(f"{name:18s} {rays.shape[1]:7d} {out[0]:11d} {out[1]:10d} {out[2]:11d} {out[3]:8d}")
    ds.close()
