#!/usr/bin/env python3
"""Run only the nearest-hit kernel (prt_propagate) on BASELINE config 2, 1M rays, `reps` times --
a target for rocprofv3 counter passes on the hit phase alone.  usage: hit_only.py [reps] [scene]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else "config2"
comps, rays = scenes.SCENES[name](scenes.product_api(), 1_000_000)
dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
ds = engine.DeviceScene.from_components(comps)
for _ in range(reps):
    ds.propagate(dev)
torch.cuda.synchronize()
ds.close()
