#!/usr/bin/env python3
"""A/B of the nearest-hit kernels (prt_propagate) on one box: lane-per-ray vs K lanes per ray, steps
through the scalar cache vs staged in LDS.  usage: hit_ab.py [scene rays]..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject

jobs = [("config2", 1_000_000), ("config3", 4_000_000)]
variants = ["", "lds", "lanes4", "lanes4,lds", "lanes8", "lanes8,lds", "lanes16"]
for name, n in jobs:
    CountedObject.reset_ids()
    comps, rays = scenes.SCENES[name](scenes.product_api(), n)
    dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    ds = engine.DeviceScene.from_components(comps)
    base = None
    for rep in range(2):
        for v in variants:
            os.environ["PRT_HIT_VARIANT"] = v
            for _ in range(3):
                t, surf = ds.propagate(dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                t, surf = ds.propagate(dev)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            if v == "":
                base = surf.clone()
            same = bool(torch.equal(surf, base))
            print(f"{name:8s} {n:8d} rays  variant {v or 'lane-per-ray (scalar steps)':28s} {us:9.1f} us per propagate  ids equal: {same}", flush=True)
    ds.close()
