"""DeviceFrame: the trace result kept columnar in HBM.

``RayTracer.trace()`` returns a pandas DataFrame like the reference
(``pyrayt/_pyrayt.py:147-186``); for a 1M-ray trace that is a 360 MB device-to-host copy which
costs two orders of magnitude more than the trace itself.  ``RayTracer.trace_device()`` returns
this view instead: the engine's (15, R) record block, one contiguous row per column, with the
handful of selections the reference's examples make on the frame (``results.loc[results[
"surface"] == id]``, per-generation slices, spot statistics: ``examples/lens_design.ipynb``)
done on the device, so that only what is looked at crosses PCIe.  Selection / reduction use
torch tensor ops: they are conveniences around the result, not part of the traced path.
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
    def spot(self, plane=("y1", "z1")):
        """(centroid, rms radius) of the end points in a transverse plane."""
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

        rows = self.rows.contiguous() if hasattr(self.rows, "contiguous") else self.rows
        return engine.to_host(rows).T

    def to_pandas(self):
        values = self.to_numpy()
        if values.shape[0] == 0:
            return pd.DataFrame(columns=COLUMNS, dtype="float64")
        return pd.DataFrame(values, columns=COLUMNS, copy=False)
