#!/usr/bin/env python3
"""Result-sink reductions against their roofline: prt_frame_reduce over the record block of a BASELINE
trace (config 2: 3M rows, one group and per-source groups; config 4: 8 sources).  Algorithmic bytes:
the eight columns the pass reads (surface, generation, id, y1, z1, x0 / y0 / tilts ... see prt_frame.hpp)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes
from pyrayt_amd import engine
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot

lib = engine.library()
COLUMNS_READ = 11  # surface, generation, id, y1, z1, x0, y0, x_tilt, y_tilt, wavelength, intensity


def run(name, parts, rays, rays_per_source, n_groups, surface):
    ds = engine.DeviceScene(SceneSnapshot(parts))
    rows, counts = ds.trace(torch.from_numpy(rays).cuda(), 10)
    out = torch.empty((n_groups, 9), dtype=torch.float64, device="cuda")  # nine sums per group
    st = engine._stream_ptr(torch, rows.device)
    nan = float("nan")

    def call():
        engine._check(lib.prt_frame_reduce(0, rows.data_ptr(), rows.stride(0), rows.shape[1],
                                           nan if surface is None else float(surface), nan,
                                           float(rays_per_source), n_groups, None, out.data_ptr(), st))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    n_rows = rows.shape[1]
    gbs = n_rows * 8 * COLUMNS_READ / (us * 1e-6) / 1e9
    print(f"{name}: {n_rows} rows, {n_groups} group(s), surface filter {surface}: {us:8.1f} us per pass, "
          f"{gbs:7.1f} GB/s of the {COLUMNS_READ} columns it reads (count check {float(out[:, 0].sum()):.0f})")
    ds.close()


CountedObject.reset_ids()
parts, rays = scenes.config2(scenes.product_api(), 1_000_000)
det = parts[1].get_id()
run("config2 all rows, one group", parts, rays, 0, 1, None)
run("config2 detector rows, one group", parts, rays, 0, 1, det)
run("config2 detector rows, 100 sources", parts, rays, 10_000, 100, det)
CountedObject.reset_ids()
parts, rays = scenes.config4(scenes.product_api(), 1_000_000)
run("config4 all rows, 8 sources", parts, rays, 1_000_000, 8, None)
run("config4 all rows, 4000 groups (global atomics)", parts, rays, 2_000, 4000, None)
