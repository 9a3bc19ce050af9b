#!/usr/bin/env python3
"""One seed of tests/test_gpu_fuzz.py::test_random_scene, verbosely: which rays differ between the HIP engine
and the C oracle, under which program forms.  usage: diag_fuzz.py <seed> [out.npz]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import helpers, scenes, test_gpu_fuzz as fz
from oracle import c_oracle, prt_oracle
from pyrayt_amd.engine import DeviceScene
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot

seed = int(sys.argv[1])
parts, rays, rng, short, odd = fz.build_random_scene(seed)
snap = SceneSnapshot(parts)
flat = helpers.flat_scene(snap)
want_t, want_surf = c_oracle.propagate(flat, rays)
np_t, np_surf = prt_oracle.propagate(flat, rays)
print("numpy oracle == C oracle:", np.array_equal(np_surf, want_surf))
for env in ({}, {"no_cull": 1}, {"no_implied": 1}, {"no_cull": 1, "no_chain": 1, "no_implied": 1}):
    ds = DeviceScene(snap, options=env)
    t, surf = ds.propagate(torch.from_numpy(rays).cuda())
    surf, t = surf.cpu().numpy(), t.cpu().numpy()
    bad = np.nonzero(surf != want_surf)[0]
    print(env, "mismatching rays:", bad[:10], [(int(surf[i]), int(want_surf[i]), t[i], want_t[i]) for i in bad[:5]])
    ds.close()
if len(sys.argv) > 2:
    np.savez(sys.argv[2], rays=rays, prims=snap.prims, nodes=snap.nodes, roots=snap.roots, materials=snap.materials)
for i in np.nonzero(surf != want_surf)[0][:3]:
    print("ray", i, "o", rays[0:4, i].tolist(), "d", rays[4:8, i].tolist(), "|d|", np.linalg.norm(rays[4:7, i]))

# per component: the engine's hit list against the numpy oracle's, for the first mismatching ray
bad = np.nonzero(surf != want_surf)[0]
if len(bad):
    i = int(bad[0])
    ds = DeviceScene(snap)
    one = np.ascontiguousarray(rays[:, i:i + 1])
    for root in range(len(flat["roots"])):
        hits, ids = ds.intersect(root, torch.from_numpy(one).cuda())
        want_h, want_i = prt_oracle.component_hits(flat, root, one[:8].reshape(2, 4, 1))
        print("component", root, "\n   engine", hits.cpu().numpy().ravel(), ids.cpu().numpy().ravel(),
              "\n   oracle", np.asarray(want_h).ravel(), np.asarray(want_i).ravel())
    ds.close()
