#!/usr/bin/env python3
"""Dump the per-wave phase stamps of generation 0 from the timing build (make -C pyrayt_amd/csrc
libprt_hip_timing.so) for tools/lookback_analysis.py:
    PRT_LIB=pyrayt_amd/csrc/libprt_hip_timing.so python tools/wave_stamps.py stamps.bin [rays] [workload] [nohints|hints]\n(-DPRT_TIMING_GEN=<g> picks the generation that is stamped)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import scenes
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot

n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
workload = sys.argv[3] if len(sys.argv) > 3 else "config2"
flags = engine.TRACE_NO_HINTS if (len(sys.argv) <= 4 or sys.argv[4] == "nohints") else 0
parts, rays = getattr(scenes, workload)(scenes.product_api(), n)
ds = engine.DeviceScene(SceneSnapshot(parts))
dev = torch.from_numpy(rays).cuda()
for _ in range(3):
    ds.trace(dev, 10, flags=flags)  # (default: the general path, every generation with its look-back)
torch.cuda.synchronize()
count = 16384 * 4 * 8
out = np.zeros(count, dtype=np.int64)
lib = engine.library()
lib.prt_debug_wave_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
rc = lib.prt_debug_wave_stamps(out.ctypes.data, count)
assert rc == 0, rc
out.tofile(sys.argv[1])
print("wrote", sys.argv[1])
