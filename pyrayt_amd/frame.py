"""DeviceFrame: the trace result kept columnar in HBM.

``RayTracer.trace()`` returns a pandas DataFrame like the reference
(``pyrayt/_pyrayt.py:147-186``); for a 1M-ray trace that is a 360 MB device-to-host copy which
costs two orders of magnitude more than the trace itself.  ``RayTracer.trace_device()`` returns
this view instead: the engine's (15, R) record block, one contiguous row per column, with the
handful of selections the reference's examples make on the frame (``results.loc[results[
"surface"] == id]``, per-generation slices, spot statistics: ``examples/lens_design.ipynb``)
done on the device, so that only what is looked at crosses PCIe.  The grouped reductions
(``group_stats``: per-source / per-wavelength spot and focus statistics, the notebook's cells
11-16) are one HIP kernel over the column block (``prt_frame_reduce``, ``csrc/prt_frame.hpp``);
row selections that return a new frame are torch indexing -- plumbing around the result.
"""

import numpy as np
import pandas as pd

COLUMNS = ("generation", "intensity", "wavelength", "index", "id", "surface",
           "x0", "y0", "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt")
_INDEX = {name: k for k, name in enumerate(COLUMNS)}


class DeviceFrame:
    def __init__(self, rows, rows_per_generation=None):
        """rows: (15, R) tensor (device or host), generation-major."""
        assert rows.shape[0] == len(COLUMNS)
        self.rows = rows
        self.rows_per_generation = list(rows_per_generation or [])

    # --- shape / access -----------------------------------------------------------------------
    columns = COLUMNS

    def __len__(self):
        return int(self.rows.shape[1])

    @property
    def shape(self):
        return (len(self), len(COLUMNS))

    def __getitem__(self, column):
        """One column as a 1-D tensor view (no copy)."""
        return self.rows[_INDEX[column]]

    # --- selections ------------------------------------------------------------------------------
    def generation(self, g):
        """Rows of generation g: a contiguous slice (rows are generation-major), no kernel."""
        if g < len(self.rows_per_generation):
            start = sum(self.rows_per_generation[:g])
            return DeviceFrame(self.rows[:, start:start + self.rows_per_generation[g]],
                               [0] * g + [self.rows_per_generation[g]])
        return self.where(generation=g)

    def where(self, **equals):
        """Rows whose named columns equal the given values, e.g. where(surface=6, generation=2)."""
        mask = None
        for name, value in equals.items():
            m = self[name] == float(value)
            mask = m if mask is None else (mask & m)
        if mask is None:
            return self
        return DeviceFrame(self.rows[:, mask])

    def select(self, mask):
        return DeviceFrame(self.rows[:, mask])

    # --- reductions the notebook does on the frame -------------------------------------------------
    def group_stats(self, surface=None, generation=None, rays_per_source=None, n_groups=None):
        """Per-source statistics of the rows that hit ``surface`` and / or belong to ``generation``
        (``examples/lens_design.ipynb`` cells 11-16: ``results.loc[results['surface'] == id]``
        grouped by ``source_id = id // rays_per_source``, ``_pyrayt.py:349-354``).

        Returns a DataFrame indexed by source id with columns ``count``, ``y`` / ``z`` (spot centroid
        of the end points), ``rms_radius`` (about that centroid), ``focus`` / ``focus_std`` (mean and
        spread of the x-axis intercepts ``x0 - x_tilt * y0 / y_tilt``, the notebook's paraxial-focus
        estimate), ``wavelength`` and ``intensity`` (means).  One library call (``prt_frame_stats``): the
        HIP reduction kernel runs twice, the second pass about the first pass's per-group means so that
        the second moments are well conditioned, and the final arithmetic happens on the device too.
        Without ``rays_per_source`` everything is one group."""
        if rays_per_source:
            if n_groups is None:
                n_groups = int(float(self["id"].max()) // rays_per_source) + 1 if len(self) else 1
        else:
            n_groups = 1
        stats = self._stats(surface, generation, rays_per_source, n_groups)
        frame = pd.DataFrame({
            "count": stats[:, 0].astype(np.int64), "y": stats[:, 1], "z": stats[:, 2], "rms_radius": stats[:, 3],
            "focus": stats[:, 4], "focus_std": stats[:, 5], "wavelength": stats[:, 6], "intensity": stats[:, 7],
        })
        frame.index.name = "source_id"
        return frame

    def _stats(self, surface, generation, rays_per_source, n_groups):
        """``prt_frame_stats``: both reduction passes and the final arithmetic on the device, one
        (n_groups, 8) block brought to the host."""
        import torch

        from . import engine

        rows = self.rows
        if rows.stride(1) != 1:
            rows = rows.contiguous()
        dev = rows.device
        lib = engine.library()
        out = torch.empty((n_groups, 8), dtype=torch.float64, device=dev)
        work = torch.empty(int(lib.prt_frame_stats_workspace_bytes(n_groups)), dtype=torch.uint8, device=dev)
        nan = float("nan")
        engine._check(lib.prt_frame_stats(
            dev.index or 0, rows.data_ptr(), rows.stride(0), rows.shape[1],
            nan if surface is None else float(surface), nan if generation is None else float(generation),
            float(rays_per_source or 0), n_groups, out.data_ptr(), work.data_ptr(), engine._stream_ptr(torch, dev)))
        return out.cpu().numpy()

    def spot(self, plane=("y1", "z1")):
        """(centroid, rms radius) of the end points in a transverse plane."""
        if tuple(plane) == ("y1", "z1") and getattr(self.rows, "is_cuda", False) and len(self):
            stats = self.group_stats().iloc[0]
            return (float(stats["y"]), float(stats["z"])), float(stats["rms_radius"])
        a, b = self[plane[0]], self[plane[1]]
        ca, cb = a.mean(), b.mean()
        rms = (((a - ca) ** 2 + (b - cb) ** 2).mean()) ** 0.5
        return (float(ca), float(cb)), float(rms)

    def axis_crossing(self):
        """x where each ray of this frame crosses the optical (x) axis in the xy plane:
        x1 - y1 * x_tilt / y_tilt (the paraxial-focus estimate of the lens-design notebook)."""
        return self["x1"] - self["y1"] * self["x_tilt"] / self["y_tilt"]

    # --- export -------------------------------------------------------------------------------------
    def to_numpy(self):
        """(R, 15) float64 view of a host copy (one D2H transfer)."""
        from . import engine

        return engine.to_host(self.rows).T  # a strided view crosses PCIe as it is: no device-side repack

    def to_pandas(self):
        values = self.to_numpy()
        if values.shape[0] == 0:
            return pd.DataFrame(columns=COLUMNS, dtype="float64")
        return pd.DataFrame(values, columns=COLUMNS, copy=False)
