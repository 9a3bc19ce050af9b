#!/usr/bin/env python3
"""Generation-by-generation comparison of prt_propagate with a scene fixture's reference intermediates
(t_g, surf_g on the reference's own ray states): prints the rays that differ.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import helpers
from test_gpu_parity import device_scene

np.set_printoptions(precision=17, linewidth=200)
for name in sys.argv[1:]:
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx))
    print(name, ds.info())
    rays = fx["rays0"]
    for g in range(int(fx["n_generations"])):
        t, surf = ds.propagate(torch.from_numpy(np.ascontiguousarray(rays)).to("cuda:0"))
        t, surf = t.cpu().numpy(), surf.cpu().numpy()
        want_t, want_s = fx[f"t_{g}"], fx[f"surf_{g}"]
        bad = np.nonzero((surf != want_s) | ~(np.isclose(t, want_t, rtol=0, atol=1e-6) | (np.isinf(t) & np.isinf(want_t))))[0]
        print(f"  generation {g}: {len(surf)} rays, {len(bad)} differ")
        for i in bad[:12]:
            print(f"    ray {i} id {rays[12, i]:.0f}: got t={t[i]!r} surf={surf[i]}  want t={want_t[i]!r} surf={want_s[i]}")
            print(f"      o={rays[0:3, i]!r} d={rays[4:7, i]!r}")
        if f"next_{g}" not in fx:
            break
        rays = fx[f"next_{g}"]
