#!/usr/bin/env python3
"""The on-device part of the frame re-assembly at full size, on one GPU: a one-rank RCCL communicator
(ncclAllGather of a rank to itself = a device copy into the staging area) + the placement kernel, over the
2 999 991 rows of a 1M-ray config-2 trace.  What an 8-GPU gather adds to this is the xGMI transfer
(7/8 x 360 MB into every GPU); what it cannot go below is this.

usage (GPU box): python tools/gather_bench.py >> profiles/r3/aux_kernels.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

import scenes
from pyrayt_amd import distributed as pdist
from pyrayt_amd import engine
from pyrayt_amd.scene import SceneSnapshot

n, limit = 1_000_000, 10
parts, rays = scenes.config2(scenes.product_api(), n, seed=1234)
ds = engine.DeviceScene(SceneSnapshot(parts))
block = torch.empty((15, n * limit), dtype=torch.float64, device="cuda:0")
rows, counts = ds.trace(torch.from_numpy(rays).cuda(), limit, out=block)
comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
matrix = comm.gather_counts(counts, limit)
for _ in range(3):
    out = comm.gather_rows(rows, matrix, limit, reuse=True)
torch.cuda.synchronize()
assert torch.equal(out, rows)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    out = comm.gather_rows(rows, matrix, limit, reuse=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = rows.shape[1] * 120
print(f"frame re-assembly on one rank (prt_allgather_rows: 15 ncclAllGather to self + k_place_rows), {rows.shape[1]} rows = "
      f"{nbytes / 1e6:.0f} MB: {ms:.3f} ms per frame = {4 * nbytes / ms / 1e6:.0f} GB/s over the 4 x {nbytes / 1e6:.0f} MB it moves "
      f"(block -> staging -> frame)")
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(reps):
    matrix = comm.gather_counts(counts, limit)
t1.record(); torch.cuda.synchronize()
print(f"count exchange (prt_allgather_counts: H2D + ncclAllGather + D2H + stream synchronisation): {t0.elapsed_time(t1) / reps * 1e3:.1f} us")
comm.close()
