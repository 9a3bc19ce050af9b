#!/usr/bin/env python3
"""Would independent half-size traces on separate streams overlap well enough to beat one
full-size trace?  K threads, each with its own DeviceScene + torch stream, trace n/K rays
concurrently (ctypes drops the GIL inside prt_trace); compare with one trace of n rays."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine

n = 1_000_000
api = scenes.product_api()
comps, rays = scenes.config2(api, n)
dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()

def bench_single(reps=30):
    ds = engine.DeviceScene.from_components(comps)
    for _ in range(5): ds.trace(dev, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ds.trace(dev, 10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    ds.close(); return dt

def bench_split(k, reps=30):
    parts = [dev[:, i * n // k:(i + 1) * n // k].contiguous() for i in range(k)]
    scenes_k = [engine.DeviceScene.from_components(comps) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    barrier = threading.Barrier(k + 1)
    def worker(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(5): scenes_k[i].trace(parts[i], 10)
            barrier.wait()
            for _ in range(reps): scenes_k[i].trace(parts[i], 10)
            barrier.wait()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
    for t in threads: t.start()
    barrier.wait(); torch.cuda.synchronize(); t0 = time.perf_counter()
    barrier.wait(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    for t in threads: t.join()
    for s in scenes_k: s.close()
    return dt

print(f"single 1M trace          {bench_single()*1e3:.4f} ms")
for k in (2, 3, 4):
    print(f"{k} concurrent {n//k}-ray traces {bench_split(k)*1e3:.4f} ms per 1M rays")
print(f"single 1M trace          {bench_single()*1e3:.4f} ms")
