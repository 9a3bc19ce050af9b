#!/usr/bin/env python3
"""Who makes the look-back wait?  From a PRT_TIMING build's stamps of generation 0 (dumped by tools/wave_stamps.py): for every tile the
time its aggregate was published (stamp 3, wave 0), when it finished shading (stamp 5) and when its
look-back + barrier ended (stamp 6); a tile can finish its look-back only after every predecessor
has published."""
import sys
import numpy as np
GHZ = 0.1  # s_memrealtime: 100 MHz, one clock for the whole chip
t = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4, 8)
ok = (t[:, 0, 0] > 0) & (t[:, 0, 7] > t[:, 0, 0])
n = int(np.nonzero(ok)[0].max()) + 1 if ok.any() else 0
t = t[:n]
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
us = lambda c: (c - t0) / (GHZ * 1e3)
start = us(t[:, 0, 0]); publish = us(t[:, 0, 3]); shaded = us(t[:, :, 5].max(axis=1)); done = us(t[:, :, 6].max(axis=1)); end = us(t[:, :, 7].max(axis=1))
pred_publish = np.maximum.accumulate(np.concatenate([[0], publish[:-1]]))
wait = done - shaded
inherent = np.maximum(0, pred_publish - shaded)
print(f"tiles {n}  kernel span {end.max():.1f} us")
print(f"hit phase (start->publish): mean {np.mean(publish-start):.2f} p50 {np.median(publish-start):.2f} p95 {np.percentile(publish-start,95):.2f} p99 {np.percentile(publish-start,99):.2f} max {np.max(publish-start):.2f}")
print(f"look-back+barrier wait: mean {wait.mean():.2f} us; of which forced by a predecessor publishing later than own shading end: mean {inherent.mean():.2f} us ({100*np.mean(inherent>0.05):.0f}% of tiles)")
print(f"residual (poll latency, barrier): mean {(wait-inherent).mean():.2f} us")
# who blocks: index distance to the blocking predecessor
blocker = np.array([np.argmax(publish[:i]) if i else 0 for i in range(n)])
dist = np.arange(n) - blocker
b = inherent > 0.05
print(f"distance to the blocking predecessor (tiles): p50 {np.median(dist[b]):.0f} p95 {np.percentile(dist[b],95):.0f}")
for lo in range(0, n, max(1, n // 8)):
    hi = min(n, lo + n // 8)
    print(f"  tiles {lo:5d}-{hi:5d}: start {start[lo:hi].mean():7.1f}  hit {np.mean(publish[lo:hi]-start[lo:hi]):6.2f}  wait {wait[lo:hi].mean():5.2f}  inherent {inherent[lo:hi].mean():5.2f}")
