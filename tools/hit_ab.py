#!/usr/bin/env python3
"""A/B of the nearest-hit kernels (prt_propagate) on one box: one ray per lane vs K lanes per ray (wavefront
shuffle min-reduce over (t, component order)), steps through the scalar cache vs staged in LDS, with and
without component cull steps.  Scenes: the BASELINE configs 2 / 3 (2 and 5 components) and the regime
the K-lanes kernels are for -- many components: a 33-component lens train (coherent beam) and a 10 x 10
grid of parts under incoherent rays.

usage (GPU box): python tools/hit_ab.py [scene ...] > profiles/r3/hit_variants.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
import pyrayt_amd as pyrayt
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject


def train33(n):
    c = pyrayt.components
    parts = [c.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in range(32)]
    parts.append(c.baffle((2, 2)).move_x(33.0))
    return parts, scenes.cone_rays(n, (-3.0, 0.0, 0.0), 3.0, 5)


def grid100(n):
    """test_incoherent_rays_over_a_grid_of_parts' scene, 10 x 10, rays from everywhere"""
    api = scenes.product_api()
    cg, m = api.cg, api.materials
    parts = []
    for ix in range(10):
        for iy in range(10):
            kind = (ix + iy) % 3
            if kind == 0:
                part = cg.Sphere(0.35, material=m.mirror)
            elif kind == 1:
                part = api.components.biconvex_lens(1.5, 1.5, 0.2, aperture=0.7).rotate_z(15 * ix)
            else:
                part = cg.Cuboid.from_sides(0.5, 0.4, 0.6, material=m.glass["SF2"]).rotate_x(20 * iy)
            parts.append(part.move(1.2 * ix - 5.4, 1.2 * iy - 5.4, 0.3 * (ix - iy)))
    return parts, scenes.random_rays(n, 29, box=7.5, degenerate=True)


JOBS = {"config2": lambda: scenes.SCENES["config2"](scenes.product_api(), 1_000_000),
        "config3": lambda: scenes.SCENES["config3"](scenes.product_api(), 4_000_000),
        "train33": lambda: train33(1_000_000), "grid100": lambda: grid100(1_000_000)}
variants = [("one ray per lane (scalar steps)", {}), ("one ray per lane, LDS-staged", {"hit_staged": 1}),
            ("4 lanes per ray", {"hit_lanes": 4}), ("4 lanes per ray, LDS-staged", {"hit_lanes": 4, "hit_staged": 1}),
            ("8 lanes per ray", {"hit_lanes": 8}), ("8 lanes per ray, LDS-staged", {"hit_lanes": 8, "hit_staged": 1}),
            ("16 lanes per ray", {"hit_lanes": 16})]
for name in (sys.argv[1:] or list(JOBS)):
    CountedObject.reset_ids()
    comps, rays = JOBS[name]()
    dev = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    for cull_label, cull in (("cull steps", {}), ("no cull steps", {"no_cull": 1})):
        if cull and len(comps) < 3:
            continue
        base = None
        for label, opts in variants:
            try:
                ds = engine.DeviceScene.from_components(comps, options=dict(opts, **cull))
                for _ in range(2):
                    t, surf = ds.propagate(dev)
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001  (e.g. a 100-part program does not fit the LDS stage)
                print(f"{name:8s} {len(comps):4d} components {cull_label:14s} {label:32s} not available: {str(exc)[:60]}", flush=True)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                t, surf = ds.propagate(dev)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            if base is None:
                base = surf.clone()
            print(f"{name:8s} {len(comps):4d} components {cull_label:14s} {label:32s} {us:10.1f} us per propagate of {dev.shape[1]} rays"
                  f"   ids equal: {bool(torch.equal(surf, base))}", flush=True)
            ds.close()
