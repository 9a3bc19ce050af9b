#!/usr/bin/env python3
"""The benchmark's timed region from a host without Python or torch (examples/c_host/prt_bench_file): BASELINE
config 2 at the shard sizes of the 1/2/4/8-GPU curve, traced through prt_trace_batch on streams the C program
makes itself.  Python only writes the input files here (the scene snapshot's tables go out as they are).

usage (GPU box): python tools/c_host_bench.py [--steps 300] >> profiles/r3/shard_scaling.txt"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from test_gpu_c_host import write_input  # noqa: E402

from pyrayt_amd.g3d.objects import CountedObject  # noqa: E402
from pyrayt_amd.scene import SceneSnapshot  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--sizes", type=int, nargs="*", default=[1_000_000, 500_000, 250_000, 125_000])
args = ap.parse_args()
host_dir = os.path.join(ROOT, "examples", "c_host")
subprocess.run(["make", "-C", host_dir], check=True, capture_output=True)
print("# the timed region from a C host (no Python, no torch): examples/c_host/prt_bench_file, config 2, "
      f"{args.steps} steps through prt_trace_batch")
with tempfile.TemporaryDirectory() as tmp:
    for n in args.sizes:
        CountedObject.reset_ids()
        parts, rays = scenes.config2(scenes.product_api(), 1_000_000, seed=1234)
        path = os.path.join(tmp, f"config2_{n}.bin")
        write_input(path, SceneSnapshot(parts), rays[:, :n].copy(), 10)
        for depth in (1, 2, 3):
            done = subprocess.run([os.path.join(host_dir, "prt_bench_file"), path, str(args.steps), str(depth)],
                                  capture_output=True, text=True)
            print(done.stdout.strip() if done.returncode == 0 else f"ERROR rays {n} depth {depth}: {done.stderr[-300:]}", flush=True)
