import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
api = scenes.product_api()
for name, n in (("config2", 1_000_000), ("config3", 1_000_000), ("config3", 4_000_000), ("config5", 2_000_000)):
    comps, rays = scenes.SCENES[name](api, n)
    dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    ds = engine.DeviceScene.from_components(comps)
    for flags in (0, 2):
        for _ in range(3):
            ds.trace(dev, 10, flags=flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            rows, counts = ds.trace(dev, 10, flags=flags)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        st = ds.trace_stats()
        print(f"{name} n={n} flags={flags}: wall {wall:.3f} ms, kernel {st['kernel_ms']:.3f} ms, launches {st['kernel_launches']}, gens {len(counts)}")
    ds.close()
