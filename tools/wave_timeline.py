#!/usr/bin/env python3
"""Per-wave phase breakdown of generation 0 of k_generation from a PRT_TIMING build.
stamps: 0 start | 1 origin/direction rows arrived | 2 nearest hit done | 3 barrier 1 passed |
        4 metadata rows arrived | 5 shaded | 6 look-back + barrier 2 passed | 7 stores issued"""
import sys
import numpy as np
GHZ = 0.1  # s_memrealtime: 100 MHz, one clock for the whole chip
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
t = t[(t[:, 0] > 0) & (t[:, 7] > t[:, 0])]
t0 = t[:, 0].min()
t = t[(t[:, 7] - t0) < 5e6]
us = lambda c: c / (GHZ * 1e3)
names = ["load o,d", "nearest hit", "barrier 1", "load metadata", "shade", "look-back+barrier 2", "stores"]
d = np.diff(t, axis=1)
print(f"waves {len(t)}  kernel span {us(t[:,7].max()-t0):.1f} us   mean wave lifetime {us((t[:,7]-t[:,0]).mean()):.2f} us")
for k, nm in enumerate(names):
    print(f"  {nm:22s} mean {us(d[:,k].mean()):7.2f} us ({100*d[:,k].mean()/(t[:,7]-t[:,0]).mean():4.1f}%)  p50 {us(np.median(d[:,k])):7.2f}  p95 {us(np.percentile(d[:,k],95)):7.2f}")
start = us(t[:, 0] - t0)
print("  start-time quartiles (us):", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
