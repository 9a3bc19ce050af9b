import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pyrayt_amd as pyrayt
lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
det = pyrayt.components.baffle((1, 1)).move_x(1)
tracer = pyrayt.RayTracer(src, [lens, det], rays_per_source=1_000_000)
def it(move):
    if move: det.move_x(1e-4)
    return tracer.trace_stats(surface=det).values()["rms_radius"][0]
for move in (False, True):
    for _ in range(20): it(move)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(200): it(move)
    print("move" if move else "same", (time.perf_counter()-t0)/200*1e3, "ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): it(move)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
