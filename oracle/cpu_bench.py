"""CPU baseline of the north-star workload on all host cores (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

The C restatement of the reference's path (oracle/prt_oracle.c, pinned to the reference's goldens by
tests/test_c_oracle.py) traced by P processes over contiguous id ranges of the same seeded 1M-ray
job -- the same sharding the GPU ranks use (rays are independent: SURVEY.md section 8e).  bench.py's
`cpu_baseline` leg runs this file as a child process (a fresh interpreter that never touches the GPU)
and reports its figure as kind "port-c" next to the single-thread numpy port.

    python -m oracle.cpu_bench --rays 1000000 --procs 64 [--repeat 20] [--workload config2] [--limit 10]

prints one JSON line: {"rows": R, "seconds": S, "procs": P, "rows_per_s": R / S}.  The timed region is
the traces alone: every worker builds its scene and ray slice first and waits at a barrier; with
--repeat K every worker traces its slice K times (a 1M-ray job split over a hundred cores is only
milliseconds of work per core otherwise).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def _worker(rank, procs, workload, n_rays, limit, repeat, barrier, out):
    import numpy as np

    import helpers
    import scenes
    from oracle import c_oracle
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    if workload == "config2":
        parts, rays = scenes.config2(scenes.product_api(), n_rays, seed=1234)
    elif workload == "config4":
        parts, rays = scenes.config4(scenes.product_api(), n_rays // 8)
    else:
        parts, rays = scenes.SCENES[workload](scenes.product_api(), n_rays)
    n = rays.shape[1]
    lo, hi = rank * n // procs, (rank + 1) * n // procs
    mine = np.ascontiguousarray(rays[:, lo:hi])
    del rays
    flat = helpers.flat_scene(SceneSnapshot(parts))
    barrier.wait()
    t0 = time.perf_counter()
    rows = 0
    for _ in range(repeat):
        frame, _ = c_oracle.trace(flat, mine, limit)
        rows += frame.shape[0]
    t1 = time.perf_counter()
    out.put((rank, rows, t0, t1))


def run(workload, n_rays, limit, procs, repeat=1):
    ctx = mp.get_context("fork")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    workers = [ctx.Process(target=_worker, args=(r, procs, workload, n_rays, limit, repeat, barrier, out)) for r in range(procs)]
    for w in workers:
        w.start()
    results = [out.get() for _ in workers]
    for w in workers:
        w.join()
    rows = sum(r[1] for r in results)
    seconds = max(r[3] for r in results) - min(r[2] for r in results)  # perf_counter is system-wide monotonic
    return {"rows": rows, "seconds": seconds, "procs": procs, "repeat": repeat, "rows_per_s": rows / seconds}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=1)
    a = ap.parse_args()
    print(json.dumps(run(a.workload, a.rays, a.limit, a.procs, a.repeat)), flush=True)
