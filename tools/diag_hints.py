import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot
CountedObject.reset_ids()
parts, rays = scenes.config2(scenes.product_api(), int(sys.argv[1]) if len(sys.argv) > 1 else 50000)
ds = engine.DeviceScene(SceneSnapshot(parts))
dev = torch.from_numpy(rays).cuda()
for limit in (10, 10, 10, 10, 1, 1):
    rows, counts = ds.trace(dev, limit)
    print(limit, counts, ds.telemetry(), ds.trace_stats())
