#!/usr/bin/env python3
"""Per-rank step time at the shard sizes of the 1/2/4/8-GPU strong-scaling curve, measured on ONE GPU.

bench.py --gpus N traces the contiguous id range [r n/N, (r+1) n/N) of one 1M-ray job per rank, with no
collective in the timed region -- so what one rank does at N = 1/2/4/8 is exactly a 1M / 500k / 250k /
125k-ray bench on one GPU.  This tool runs those four ways -- "overlap" (bench.py's default: 2-3 traces
in flight on as many HIP streams, the region issued as one prt_trace_batch call), "overlap py" (the same
issued from a Python loop over prt_trace_begin / prt_trace_end), "one stream" (one trace ahead on the same stream: prt_trace_begin /
end) and "synchronous" (prt_trace) -- and prints ms_per_step, the generation kernels' own time per step
(always measured on one stream) and, for the two single-stream modes, the host share.

usage (GPU box): python tools/shard_scaling.py [--steps 200] > profiles/r3/shard_scaling.txt"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--sizes", type=int, nargs="*", default=[1_000_000, 500_000, 250_000, 125_000])
ap.add_argument("--workload", default="config2")
args = ap.parse_args()

print(f"# {args.workload}, one GPU, {args.steps} timed steps per line; 'N' = the GPU count whose per-rank shard this is")
print(f"{'rays':>9s} {'N':>2s} {'mode':>12s} {'ms/step':>9s} {'kernel ms':>10s} {'host us':>8s} {'rows/s (1 GPU)':>15s} "
      f"{'x N':>10s} {'frac':>6s} {'launch us':>9s} {'device frac':>11s}")
for n in args.sizes:
    for mode, extra in (("overlap", []), ("overlap py", ["--python-loop"]), ("one stream", ["--streams", "1"]),
                        ("synchronous", ["--no-pipeline"])):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rays", str(n), "--steps", str(args.steps), "--warmup", "20",
               "--no-cpu-baseline", "--side-steps", "0", "--workload", args.workload] + extra
        done = subprocess.run(cmd, capture_output=True, text=True)
        try:
            d = json.loads(done.stdout.strip().splitlines()[-1])
        except Exception:  # noqa: BLE001
            print(f"{n:9d} ERROR {done.stderr[-300:]}")
            continue
        r = d["roofline"]
        gpus = round(args.sizes[0] / n)
        host = "" if mode.startswith("overlap") else f"{(d['ms_per_step'] - r['kernel_ms_per_step']) * 1e3:8.1f}"
        label = mode if not mode.startswith("overlap") else f"{mode} x{d['config']['streams']}"
        print(f"{n:9d} {gpus:2d} {label:>12s} {d['ms_per_step']:9.4f} {r['kernel_ms_per_step']:10.4f} "
              f"{host:>8s} {d['value']:15.4e} {d['value'] * gpus:10.3e} "
              f"{r['frac']:6.3f} {r['avg_launch_ms'] * 1e3:9.1f} {r['device_aggregate']['frac']:11.3f}", flush=True)
