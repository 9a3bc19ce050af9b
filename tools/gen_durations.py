#!/usr/bin/env python3
"""Per-generation medians from the per-dispatch duration files of tools/gen_trace.sh:
    python tools/gen_durations.py gpurun_out/gen_trace > profiles/rN/gen_durations.txt
(expects config3 / config3_nokeep / config2_rotating / config2_rotating_nokeep: see tools/refresh_profiles.sh)"""
import os
import sys

import numpy as np

where = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gen_trace"
print("# k_generation dispatches in launch order (rocprofv3 --kernel-trace, microseconds); tools/gen_trace.sh")
print("# (each process ramps up over its first traces: compare like positions, or the per-generation medians of the last four traces)")
for name, gens, label in [
        ("config3", 7, "config 3, 4M rays, default (generation 1 dense with its absorbed rays kept)"),
        ("config3_nokeep", 7, "config 3, 4M rays, PRT_TRACE_NO_SPARSE_KEEP (generation 1 compacts by look-back)"),
        ("config2_rotating", 3, "config 2, 1M rays, rotating ray sets, default (generation 1: mode 4, generation 2: on the dead list)"),
        ("config2_rotating_nokeep", 3, "config 2, 1M rays, rotating ray sets, PRT_TRACE_NO_SPARSE_KEEP (generation 1 by look-back)")]:
    path = os.path.join(where, f"{name}_durations.txt")
    if not os.path.exists(path):
        continue
    values = [float(x) for x in open(path).read().splitlines()[-1].split()]
    values = [v for v in values if v > 20]  # (drop the one-block kernels)
    table = np.array(values[:len(values) // gens * gens]).reshape(-1, gens)
    median = np.median(table[-4:], axis=0)
    print(f"{label}: per-generation median of the last four traces: " + " ".join(f"{x:.1f}" for x in median) + f"  (sum {median.sum():.1f})")
    print("   all: " + " ".join(f"{x:.1f}" for x in values))
