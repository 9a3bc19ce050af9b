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
    base = None
    scene_of = {}
    for v in variants:  # one compiled scene per kernel variant (prt_scene_options.hit_lanes / hit_staged)
        lanes = [int(part[5:]) for part in v.split(",") if part.startswith("lanes")]
        scene_of[v] = engine.DeviceScene.from_components(comps, options={"hit_lanes": lanes[0] if lanes else 0,
                                                                         "hit_staged": int("lds" in v)})
    for rep in range(2):
        for v in variants:
            ds = scene_of[v]
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
    for ds in scene_of.values():
        ds.close()
