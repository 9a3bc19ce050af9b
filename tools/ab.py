#!/usr/bin/env python3
"""A/B helper: run bench.py variants back to back (interleaved, repeated) on one box.
usage: tools/ab.py [--reps R] "label:ENV=V,ENV2=V2:--flags 2" ...   (env and args optional)"""
import json, os, subprocess, sys
reps = 2
specs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--reps":
        reps = int(args.pop(0))
    else:
        specs.append(a)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
results = {}
for rep in range(reps):
    for spec in specs:
        parts = spec.split(":")
        label = parts[0]
        env = dict(os.environ)
        if len(parts) > 1 and parts[1]:
            for kv in parts[1].split(","):
                k, v = kv.split("=")
                env[k] = v
        extra = parts[2].split() if len(parts) > 2 and parts[2] else []
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + extra,
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            results.setdefault(label, []).append((d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["value"]))
        except Exception as e:
            results.setdefault(label, []).append(("ERR", out.stderr[-300:], 0))
for label, vals in results.items():
    print(label, " | ".join(f"step {v[0]:.4f} ms kern {v[1]:.4f} ms {v[2]:.3e}/s" if v[0] != "ERR" else f"ERR {v[1]}" for v in vals))
