#!/usr/bin/env python3
"""How k_hit (prt_propagate) time scales with scene content: 1M rays, various scenes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import g3d as cg, engine
from pyrayt_amd.scene import SceneSnapshot

n = 1_000_000
_, rays = scenes.config2(scenes.product_api(), n)
rays_dev = torch.from_numpy(rays).cuda()
m = pyrayt.materials
def lens(): return pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
def det(): return pyrayt.components.baffle((1, 1)).move_x(1)
cases = {
    "plane": lambda: [det()],
    "2 planes": lambda: [det(), det().move_x(1)],
    "4 planes": lambda: [det(), det().move_x(1), det().move_x(2), det().move_x(3)],
    "sphere": lambda: [cg.Sphere(0.5, material=m.mirror)],
    "2 spheres": lambda: [cg.Sphere(0.5, material=m.mirror), cg.Sphere(0.4, material=m.mirror).move_x(2)],
    "sphere&sphere": lambda: [cg.csg.intersect(cg.Sphere(2, material=m.mirror).move_x(1.9), cg.Sphere(2, material=m.mirror).move_x(-1.9))],
    "lens": lambda: [lens()],
    "lens+plane": lambda: [lens(), det()],
}
for name, make in cases.items():
    ds = engine.DeviceScene(SceneSnapshot(make()))
    for _ in range(3): ds.propagate(rays_dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ds.propagate(rays_dev)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1) / 10 * 1000:7.1f} us per 1M-ray propagate")
    ds.close()
