#!/usr/bin/env python3
"""What a record plan buys on BASELINE config 2 (1M rays, four seeded ray sets in rotation): the whole frame, the rows
of the detector only, and sums only -- each as blocking prt_trace calls on one stream and as a batch with two traces
in flight (the bench line's regime), with the bytes each form has to move per ray beside it.

usage: tools/plan_bench.py [--rays N] [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import scenes  # noqa: E402
from pyrayt_amd import engine  # noqa: E402
from pyrayt_amd.g3d.objects import CountedObject  # noqa: E402
from pyrayt_amd.scene import SceneSnapshot  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sets = []
    for seed in (1234, 1235, 1236, 1237):
        CountedObject.reset_ids()
        parts, rays = scenes.config2(scenes.product_api(), args.rays, seed=seed)
        sets.append(torch.from_numpy(np.ascontiguousarray(rays)).to(dev))
    scene = engine.DeviceScene(SceneSnapshot(parts))
    detector = parts[1].get_id()
    limit = 10
    plans = {
        "whole frame (no plan)": None,
        "rows of the detector": engine.RecordPlan(surfaces=(detector,), rows=True, generation_limit=limit),
        "rows of the detector + sums": engine.RecordPlan(surfaces=(detector,), rows=True, stats=True, generation_limit=limit),
        "sums only (detector)": engine.RecordPlan(surfaces=(detector,), rows=False, stats=True, generation_limit=limit),
        "sums only (every surface)": engine.RecordPlan(rows=False, stats=True, generation_limit=limit),
    }
    out = {}
    block = torch.empty((15, args.rays * limit), dtype=torch.float64, device=dev)
    block2 = torch.empty_like(block)
    for name, plan in plans.items():
        for k in range(12):
            rows, counts = scene.trace(sets[k % 4], limit, out=block, plan=plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            rows, counts = scene.trace(sets[k % 4], limit, out=block)
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t0) / args.steps * 1e3
        stats = scene.trace_stats()
        # two in flight: every ticket of the batch needs the plan
        for ticket in range(2):
            scene.set_plan(ticket, plan, dev)
        batch = engine.TraceBatch(scene, [sets[k % 4] for k in range(args.steps)], limit, depth=2, outs=[block, block2],
                                  flags=engine.TRACE_NO_TIMING)
        batch.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.run()
        torch.cuda.synchronize()
        overlap_ms = (time.perf_counter() - t0) / args.steps * 1e3
        for ticket in range(2):
            scene.set_plan(ticket, None, dev)
        out[name] = {"blocking_ms_per_trace": round(sync_ms, 4), "two_in_flight_ms_per_trace": round(overlap_ms, 4),
                     "rows_stored": int(sum(counts)), "kernel_ms_last_trace": round(stats["kernel_ms"], 4),
                     "launches": stats["kernel_launches"]}
        print(f"{name:32s} blocking {sync_ms:.4f} ms   two in flight {overlap_ms:.4f} ms   rows stored {int(sum(counts)):8d}   "
              f"kernel {stats['kernel_ms']:.4f} ms / {stats['kernel_launches']} launches", flush=True)
    print(json.dumps({"rays": args.rays, "steps": args.steps, "telemetry": scene.telemetry(), "results": out}))


if __name__ == "__main__":
    main()
