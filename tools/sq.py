#!/usr/bin/env python3
"""Summarise tools/sq.sh output: per kernel, mean of every counter over its dispatches."""
import collections, csv, glob, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sq"
tag = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(f"{src}/{tag}*_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][:48]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, counters in acc.items():
    if not any(k in name for k in ("k_hit", "k_generation", "k_render")):
        continue
    print(name)
    for c, v in counters.items():
        v = v[1:] if len(v) > 2 else v
        print(f"   {c:28s} {sum(v) / len(v):16.1f}  ({len(v)} dispatches)")
