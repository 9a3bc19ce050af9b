#!/usr/bin/env python3
"""Rays of a fuzz seed where prt_propagate and the C oracle disagree.  usage: diag_fuzz.py seed..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import helpers, scenes
from oracle import c_oracle
from test_gpu_fuzz import random_component
from pyrayt_amd.engine import DeviceScene
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot
np.set_printoptions(precision=17, linewidth=220)
for seed in map(int, sys.argv[1:]):
    api = scenes.product_api()
    rng = np.random.default_rng(1000 + seed)
    CountedObject.reset_ids()
    parts = []
    for _ in range(rng.integers(1, 5)):
        comp = random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 4)))
        comp.move(*rng.uniform(-2.0, 2.0, 3))
        parts.append(comp)
    rays = scenes.random_rays(20_000, seed=5000 + seed, box=4.0, wavelength=0.55)
    rays[10] = rng.uniform(0.4, 0.8, rays.shape[1])
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    for knob in ("", "PRT_NO_CHAIN", "PRT_NO_CULL"):
        if knob: os.environ[knob] = "1"
        ds = DeviceScene(snap)
        t, surf = ds.propagate(torch.from_numpy(rays).to("cuda:0"))
        if knob: del os.environ[knob]
        t, surf = t.cpu().numpy(), surf.cpu().numpy()
        wt, ws = c_oracle.propagate(flat, rays)
        bad = np.nonzero(surf != ws)[0]
        print(f"seed {seed} [{knob or 'default'}] info {ds.info()} mismatches {len(bad)}")
        for i in bad[:6]:
            print(f"  ray {i}: hip t={t[i]!r} s={surf[i]} | oracle t={wt[i]!r} s={ws[i]}  o={rays[0:3,i]!r} d={rays[4:7,i]!r}")
    print("node ops", flat["node_op"], "prim types", flat["prim_type"], "roots", flat["roots"])
