#!/usr/bin/env python3
"""Where the host time of a trace goes (experiment build libprt_hip_hostprof.so, PRT_LIB points at it):
stamps inside prt_trace + the python wrapper around it + a bare ctypes call with prebuilt arguments."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PRT_LIB", os.path.join(ROOT, "pyrayt_amd", "csrc", "libprt_hip_hostprof.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot
lib = engine.library()
names = ["", "enter->loop", "settle", "ev0 recorded", "first launch", "all launched", "epoch seen", "exit"]
for n in (256, 131072, 1000000):
    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    dev = torch.from_numpy(rays).cuda()
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda")
    for _ in range(20): ds.trace(dev, 10, out=block)
    torch.cuda.synchronize()
    out = (ctypes.c_double * 9)()
    lib.prt_debug_host_profile(out); base = list(out)
    t0 = time.perf_counter(); k = 0.0
    for _ in range(200):
        ds.trace(dev, 10, out=block)
    dt = (time.perf_counter() - t0) / 200
    lib.prt_debug_host_profile(out)
    cnt = out[8] - base[8]
    stamps = [(out[i] - base[i]) / cnt for i in range(8)]
    print(f"n={n}: python step {dt*1e6:.1f} us; inside prt_trace (cumulative us): " +
          ", ".join(f"{names[i]} {stamps[i]:.1f}" for i in range(1, 8)))
    # bare ctypes call
    work = ds._work; counts = (ctypes.c_int64 * 10)()
    args = (ds.handle, 0, dev.data_ptr(), n, dev.stride(0), 10, float(engine.DEFAULT_RAY_OFFSET), block.data_ptr(),
            block.shape[1], counts, work.data_ptr(), 0, engine._stream_ptr(torch, dev.device))
    for _ in range(5): lib.prt_trace(*args)
    t0 = time.perf_counter()
    for _ in range(200): lib.prt_trace(*args)
    dt2 = (time.perf_counter() - t0) / 200
    print(f"          bare ctypes call {dt2*1e6:.1f} us; kernel_ms {ds.trace_stats()['kernel_ms']*1e3:.1f} us")
    ds.close()
