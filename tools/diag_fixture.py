#!/usr/bin/env python3
"""A golden scene fixture, verbosely: rows the HIP engine and the reference disagree on, per program form.
usage: diag_fixture.py <scene_name.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import helpers
from pyrayt_amd.engine import DeviceScene
from test_gpu_parity import FixtureSnapshot

fx = helpers.load(sys.argv[1])
want = fx["frame"]
rays = fx["rays0"]
limit = int(fx["generation_limit"])
for env in ({}, {"no_cull": 1}, {"no_chain": 1}, {"no_cull": 1, "no_chain": 1}):
    ds = DeviceScene(FixtureSnapshot(helpers.scene_of(fx)), options=env)
    rows, counts = ds.trace(torch.from_numpy(rays).cuda(), limit)
    got = rows.cpu().numpy().T
    key = lambda f: {(int(r[0]), int(r[4])): r for r in f}
    g, w = key(got), key(want)
    missing = sorted(set(w) - set(g)); extra = sorted(set(g) - set(w))
    wrong = sorted(k for k in set(g) & set(w) if g[k][5] != w[k][5])
    print(env, "rows", got.shape[0], "want", want.shape[0], "missing", missing[:8], "extra", extra[:8], "other surface", wrong[:8])
    for gen, rid in (missing + extra + wrong)[:6]:
        col = int(np.nonzero(rays[12] == rid)[0][0])
        print("   gen", gen, "ray", rid, "o", rays[0:3, col].tolist(), "d", rays[4:7, col].tolist(),
              "want surf", w.get((gen, rid), [None] * 6)[5], "got surf", g.get((gen, rid), [None] * 6)[5])
    ds.close()
