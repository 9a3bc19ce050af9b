#!/usr/bin/env python3
"""A design loop as the reference's lens_design notebook runs it: move a part, trace, read the spot size off
the detector -- timed per stage, with the result never leaving the GPU (RayTracer.trace_device +
DeviceFrame.group_stats)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pyrayt_amd as pyrayt

lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
det = pyrayt.components.baffle((1, 1)).move_x(1)
tracer = pyrayt.RayTracer(src, [lens, det], rays_per_source=1_000_000)
for _ in range(3):
    tracer.trace_device()


def iteration(move, fused):
    if move:
        det.move_x(1e-4)
    if fused:  # the sums are accumulated by the generation kernels: no row is stored, none is read back
        return tracer.trace_stats(surface=det).values()["rms_radius"][0]
    frame = tracer.trace_device()
    return frame.group_stats(surface=det.get_id())["rms_radius"].iloc[0]


for fused in (False, True):
    for move in (False, True):
        for _ in range(5):
            iteration(move, fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            rms = iteration(move, fused)
        dt = (time.perf_counter() - t0) / 50
        print(f"{'fused sink (trace_stats)' if fused else 'frame + group_stats   '} | {'moving the detector' if move else 'unchanged system   '}: "
              f"{dt * 1e3:.3f} ms per iteration (rms spot {rms:.6e})")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(30):
        iteration(True, True)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
