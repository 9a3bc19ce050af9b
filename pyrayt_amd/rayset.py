"""RaySet: the (13, n) float64 ray state matrix.

Layout is bit-identical to the reference's ``pyrayt._pyrayt.RaySet`` (``_pyrayt.py:13-144``):
rows 0-3 origin xyzw, rows 4-7 direction xyzw, rows 8-12 ``generation, intensity, wavelength,
index, id``.  Row-major with one contiguous row per field, so the very same bytes are the
device buffer the HIP kernels stream (one ray per lane, coalesced loads of every row).
"""
import numpy as np

ROWS = 13
_SPATIAL_ROWS = 8


class RaySet(np.ndarray):
    fields = ("generation", "intensity", "wavelength", "index", "id")
    """The metadata fields that can be accessed from the rayset"""

    def __new__(cls, n_rays):
        return np.zeros((ROWS, n_rays), dtype=float).view(cls)

    def __init__(self, n_rays, *args, **kwargs):
        super().__init__()
        # defaults of _pyrayt.py:38-43
        self[3] = 1.0  # homogeneous w of the origins
        self.generation = 0
        self.intensity = 100.0
        self.wavelength = 0.633
        self.index = 1
        self.id = np.arange(n_rays)

    @property
    def n_rays(self):
        return self.shape[-1]

    @property
    def rays(self):
        """(2,4,n) view: [0] origins, [1] directions, homogeneous coordinates."""
        return self[:_SPATIAL_ROWS].reshape((2, 4, -1))

    @rays.setter
    def rays(self, update):
        self[:_SPATIAL_ROWS] = np.asarray(update).reshape(_SPATIAL_ROWS, -1)

    @property
    def metadata(self):
        return self[_SPATIAL_ROWS:]

    @metadata.setter
    def metadata(self, update):
        self[_SPATIAL_ROWS:] = update


def _row_property(row, name):
    def getter(self):
        return self[row]

    def setter(self, update):
        self[row] = update

    return property(getter, setter, doc=f"view of the `{name}` row of every ray")


for _offset, _name in enumerate(RaySet.fields):
    setattr(RaySet, _name, _row_property(_SPATIAL_ROWS + _offset, _name))
