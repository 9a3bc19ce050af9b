#!/usr/bin/env python3
"""When does the HIP runtime read GPU_MAX_HW_QUEUES -- when libamdhip64 is loaded (import torch) or when the runtime
is initialised (first HIP call)?  Usage: queue_probe.py <value> <early|late|after_available|after_count>: set the variable before / after `import
torch` (no HIP call in between), then run the same spin kernel on 1 and on 4 streams and print the ratio of the wall
times: ~1 when the four overlap (each stream has a hardware queue), ~4 when they share one queue."""
import os
import sys
import time

value, when = sys.argv[1], sys.argv[2]
if when == "early":
    os.environ["GPU_MAX_HW_QUEUES"] = value
import torch

if when == "late":
    assert not torch.cuda.is_initialized()
    os.environ["GPU_MAX_HW_QUEUES"] = value
if when == "after_available":  # torch.cuda.is_available() has asked the runtime for its device count
    torch.cuda.is_available()
    print("is_initialized after is_available():", torch.cuda.is_initialized())
    os.environ["GPU_MAX_HW_QUEUES"] = value
if when == "after_count":
    torch.cuda.device_count()
    os.environ["GPU_MAX_HW_QUEUES"] = value
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(4)]
cycles = 20_000_000  # ~10 ms


def run(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams[:k]:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run(4)
one = min(run(1) for _ in range(3))
four = min(run(4) for _ in range(3))
print(f"GPU_MAX_HW_QUEUES={value} set {when}: 1 stream {one * 1e3:.2f} ms, 4 streams {four * 1e3:.2f} ms, ratio {four / one:.2f}")
