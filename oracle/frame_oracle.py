"""CPU restatement of the result-sink reductions (TEST INFRASTRUCTURE: only tests/ may import this).

What the reference's users compute from the frame ``RayTracer.trace()`` returns
(``/root/reference/examples/lens_design.ipynb`` cells 11-16: ``results.loc[results['surface'] == id]``, grouped
by ``source_id = id // rays_per_source`` -- ``pyrayt/_pyrayt.py:349-354`` -- then means / spreads of the end
points ``y1, z1`` and of the x-axis intercepts ``x0 - x_tilt * y0 / y_tilt``), split the way the HIP library
splits it (``pyrayt_amd/csrc/prt_frame.hpp``): additive per-group sums of one pass over the rows, the pivots a
second pass runs about, and the final arithmetic.  Pinned by tests/test_distributed.py against pandas on the
reference's own frames (the golden fixtures)."""
import numpy as np

COL = {name: k for k, name in enumerate(("generation", "intensity", "wavelength", "index", "id", "surface", "x0", "y0",
                                         "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt"))}


def reduce_sums(rows, surface, generation, rays_per_source, n_groups, pivots=None):
    """(n_groups, 9) sums of the selected rows of a (15, R) block: count, sum (y1 - py), sum (z1 - pz),
    sum ((y1 - py)^2 + (z1 - pz)^2), sum (f - pf), sum (f - pf)^2, sum wavelength, sum intensity, rows with a
    finite intercept f."""
    rows = np.asarray(rows, dtype=float)
    keep = np.ones(rows.shape[1], dtype=bool)
    if surface is not None:
        keep &= rows[COL["surface"]] == surface
    if generation is not None:
        keep &= rows[COL["generation"]] == generation
    group = np.floor(rows[COL["id"]] / rays_per_source).astype(np.int64) if rays_per_source else np.zeros(rows.shape[1], np.int64)
    keep &= (group >= 0) & (group < n_groups)
    out = np.zeros((n_groups, 9))
    pivots = np.zeros((n_groups, 3)) if pivots is None else np.asarray(pivots, dtype=float)
    with np.errstate(all="ignore"):
        focus = rows[COL["x0"]] - rows[COL["x_tilt"]] * rows[COL["y0"]] / rows[COL["y_tilt"]]
    for g in range(n_groups):
        sel = keep & (group == g)
        y, z = rows[COL["y1"], sel] - pivots[g, 0], rows[COL["z1"], sel] - pivots[g, 1]
        f = focus[sel] - pivots[g, 2]
        ok = np.isfinite(f)
        out[g] = (sel.sum(), y.sum(), z.sum(), (y * y + z * z).sum(), f[ok].sum(), (f[ok] ** 2).sum(),
                  rows[COL["wavelength"], sel].sum(), rows[COL["intensity"], sel].sum(), ok.sum())
    return out


def pivots_of(sums):
    """(n_groups, 3): mean y1, mean z1, mean intercept of a first pass's sums (zeros for an empty group)."""
    sums = np.asarray(sums, dtype=float)
    count, with_focus = np.maximum(sums[:, 0], 1.0), np.where(sums[:, 8] > 0, sums[:, 8], 1.0)
    return np.stack((sums[:, 1] / count, sums[:, 2] / count, sums[:, 4] / with_focus), axis=1)


def finish(sums, pivots):
    """(n_groups, 8): count, y, z, rms radius, focus, focus std, wavelength, intensity (NaN for an empty group)."""
    sums, pivots = np.asarray(sums, dtype=float), np.asarray(pivots, dtype=float)
    out = np.full((sums.shape[0], 8), np.nan)
    out[:, 0] = sums[:, 0]
    for g, s in enumerate(sums):
        if s[0] > 0:
            dy, dz = s[1] / s[0], s[2] / s[0]
            out[g, 1], out[g, 2] = pivots[g, 0] + dy, pivots[g, 1] + dz
            out[g, 3] = np.sqrt(max(s[3] / s[0] - dy * dy - dz * dz, 0.0))
            out[g, 6], out[g, 7] = s[6] / s[0], s[7] / s[0]
        if s[8] > 0:
            df = s[4] / s[8]
            out[g, 4] = pivots[g, 2] + df
            out[g, 5] = np.sqrt(max(s[5] / s[8] - df * df, 0.0))
    return out
