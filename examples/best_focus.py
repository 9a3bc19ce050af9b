#!/usr/bin/env python3
"""Where is a lens's best focus?  A design loop written against the reference's API (`import pyrayt_amd as pyrayt`):
move the detector, trace, read one number -- the loop of the reference's examples/lens_design.ipynb, here with the
number (the RMS spot radius on the detector) accumulated inside the generation kernels, so that an iteration
neither stores a row nor reads one back (RayTracer.trace_stats).

    python examples/best_focus.py [rays]

The last step traces once more keeping the detector's rows only (RayTracer.record_only) and computes the same radius
the way the notebook does, from the pandas frame."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyrayt_amd as pyrayt  # noqa: E402


def build(rays):
    lens = pyrayt.components.thick_lens(40, -200, 5, aperture=25.4, material=pyrayt.materials.glass["BK7"])
    source = pyrayt.components.LineOfRays(spacing=16, wavelength=0.55).move_x(-50)
    detector = pyrayt.components.baffle((25.4, 25.4)).move_x(40)
    tracer = pyrayt.RayTracer(source, [lens, detector], rays_per_source=rays)
    return tracer, detector


def spot_radius(tracer, detector, x):
    """RMS spot radius with the detector at x (the fused sink: sums only)."""
    detector.move_x(x - detector.get_position()[0])
    return float(tracer.trace_stats(surface=detector).values("last")["rms_radius"][0])


def best_focus(tracer, detector, lo, hi, coarse=25, tol=1e-4):
    """Coarse scan, then a golden-section search around its minimum; returns (x, radius, traces)."""
    xs = np.linspace(lo, hi, coarse)
    radii = [spot_radius(tracer, detector, x) for x in xs]
    k = int(np.argmin(radii))
    a, b = xs[max(k - 1, 0)], xs[min(k + 1, coarse - 1)]
    traces = coarse
    g = (np.sqrt(5.0) - 1.0) / 2.0
    c, d = b - g * (b - a), a + g * (b - a)
    fc, fd = spot_radius(tracer, detector, c), spot_radius(tracer, detector, d)
    traces += 2
    while b - a > tol:
        if fc < fd:
            b, d, fd = d, c, fc
            c = b - g * (b - a)
            fc = spot_radius(tracer, detector, c)
        else:
            a, c, fc = c, d, fd
            d = a + g * (b - a)
            fd = spot_radius(tracer, detector, d)
        traces += 1
    x = 0.5 * (a + b)
    return x, spot_radius(tracer, detector, x), traces + 1


def main(rays=200_000, verbose=True):
    tracer, detector = build(rays)
    spot_radius(tracer, detector, 40.0)  # (the first trace of a scene compiles it and sizes the workspace)
    t0 = time.perf_counter()
    x, radius, traces = best_focus(tracer, detector, 40.0, 120.0)
    seconds = time.perf_counter() - t0
    # the notebook's way, on the rows of the detector alone
    tracer.record_only(detector)
    frame = tracer.trace()
    last = frame[frame["generation"] == frame["generation"].max()]
    y, z = last["y1"].to_numpy(), last["z1"].to_numpy()
    from_rows = float(np.sqrt(np.mean((y - y.mean()) ** 2 + (z - z.mean()) ** 2)))
    if verbose:
        print(f"best focus at x = {x:.4f} mm, RMS spot radius {radius:.6e} mm "
              f"({traces} traces of {rays} rays in {seconds * 1e3:.1f} ms: {seconds / traces * 1e3:.3f} ms per iteration)")
        print(f"from the detector's rows ({len(frame)} rows, every one on surface {detector.get_id()}): {from_rows:.6e} mm")
    return dict(x=x, radius=radius, from_rows=from_rows, traces=traces, rows=len(frame),
                surfaces=set(frame["surface"].astype(int)), detector=detector.get_id())


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200_000)
