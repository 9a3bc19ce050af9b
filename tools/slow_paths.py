#!/usr/bin/env python3
"""How often the shipping library's shortcuts fall through to their exact paths, per fixture and for the
BASELINE configs at size: a trace with PRT_TRACE_COUNT_PATHS (include/prt.h) counts, per ray-generation,
the rays that are not well formed (they take no shortcut at all) and, per CSG node evaluation under an
implied cull box, how many had survivors and how many of those evaluated upstream's box test exactly;
prt_trace_telemetry returns the totals.  (A counted trace runs on the three-kernel path.)

usage (GPU box): python tools/slow_paths.py [fixture ...] > profiles/r3/slow_paths.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import helpers
import scenes
from test_gpu_parity import device_scene
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot

KEYS = ("rays_not_well_formed", "implied_box_nodes", "exact_box_tests")


def counted(ds, rays, limit):
    before = ds.telemetry()
    rows, counts = ds.trace(rays, limit, flags=engine.TRACE_COUNT_PATHS)
    torch.cuda.synchronize()
    after = ds.telemetry()
    assert after["counted_traces"] == before["counted_traces"] + 1
    st = ds.trace_stats()
    return st["ray_generations"], [after[k] - before[k] for k in KEYS]


names = sys.argv[1:] or ["config2", "config3", "config4", "config5", "mirrors_and_stops", "stopped_lens",
                         "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a",
                         "adv_bench_a", "stale_box"]
print(f"{'scene':28s} {'ray-generations':>15s} {'not well formed':>15s} {'implied-box nodes':>17s} {'exact box tests':>15s}")
for name in names:
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx))
    rays = torch.from_numpy(np.ascontiguousarray(fx["rays0"])).to("cuda:0")
    total, (bad, nodes, exact) = counted(ds, rays, int(fx["generation_limit"]))
    print(f"{'fixture ' + name:28s} {total:15d} {bad:15d} {nodes:17d} {exact:15d}")
    ds.close()
for name, n in (("config2", 1_000_000), ("config3", 4_000_000), ("config4", 8_000_000), ("config5", 2_000_000)):
    CountedObject.reset_ids()
    if name == "config4":
        parts, rays = scenes.config4(scenes.product_api(), n // 8)
    else:
        parts, rays = scenes.SCENES[name](scenes.product_api(), n)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    total, (bad, nodes, exact) = counted(ds, torch.from_numpy(rays).to("cuda:0"), 10)
    print(f"{'BASELINE ' + name + f' {n} rays':28s} {total:15d} {bad:15d} {nodes:17d} {exact:15d}")
    ds.close()
