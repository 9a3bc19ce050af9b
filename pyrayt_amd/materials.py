"""Traceable materials.

Same surface as the reference's ``pyrayt/materials.py``: ``absorber``, ``mirror``, the
``glass`` table, ``BasicRefractor``, ``SellmeierRefractor`` and ``Material.trace(surface,
ray_set)``.  A material here is a *description* (kind + up to six coefficients, see
``include/prt.h`` PRT_MAT_*); the shading arithmetic -- zeroing, reflecting about the world
normal, vector Snell refraction with the Sellmeier index (``materials.py:47-50, 58-62, 70-75,
136-145`` and ``tinygfx/g3d/operations.py:86-162``) -- is done per ray by the HIP kernels.
``trace()`` forwards to the engine so existing call sites keep working.
"""
import math
from functools import lru_cache

import numpy as np

from .g3d.materials import gooch

# kind codes, must match include/prt.h PRT_MAT_*
NONE, ABSORBER, MIRROR, CONST_INDEX, SELLMEIER = range(5)


class TracableMaterial(gooch.Material):
    kind = NONE
    _base_material = gooch.BLACK
    """What the renderers draw the material with (``materials.py:11-24``): absorbers black,
    mirrors and glasses blue."""

    def shade(self, rays, normals, light_positions):
        return self._base_material.shade(rays, normals, light_positions)

    def coefficients(self):
        return ()

    def packed_coefficients(self):
        c = [float(v) for v in self.coefficients()]
        return c + [0.0] * (6 - len(c))

    def trace(self, surface, ray_set):
        """Shade ``ray_set`` (whose origins sit on ``surface``) in place and return it.
        Runs on the HIP engine; there is no host implementation."""
        from . import engine

        return engine.material_trace(self, surface, ray_set)


class _AbsorbingMaterial(TracableMaterial):
    """Ideal absorber: the direction of every interacting ray becomes <0,0,0>, which the
    tracer reads as 'terminate' (``materials.py:41-50``)."""

    kind = ABSORBER


class _ReflectingMaterial(TracableMaterial):
    """Ideal mirror, no change of index or intensity (``materials.py:53-62``)."""

    kind = MIRROR
    _base_material = gooch.BLUE


class Glass(TracableMaterial):
    """Refracting material with a wavelength dependent index (``materials.py:65-99``)."""

    _base_material = gooch.BLUE

    def index_at(self, wavelength):
        raise NotImplementedError

    @lru_cache(100)
    def abbe(self):
        """Abbe number from the F, d and C lines (host-side helper, ``materials.py:77-86``)."""
        n_f, n_d, n_c = (self.index_at(w) for w in (0.4861, 0.5893, 0.6563))
        return (n_d - 1) / (n_f - n_c)


class BasicRefractor(Glass):
    """Non-dispersive glass (``materials.py:102-118``)."""

    kind = CONST_INDEX

    def __init__(self, refractive_index):
        self._refractive_index = refractive_index

    def coefficients(self):
        return (self._refractive_index,)

    def index_at(self, wavelength):
        if isinstance(wavelength, np.ndarray):
            return np.full(wavelength.shape, self._refractive_index)
        return self._refractive_index


class SellmeierRefractor(Glass):
    """n(w)^2 = 1 + sum_i b_i w^2 / (w^2 - c_i), w in microns (``materials.py:121-145``)."""

    kind = SELLMEIER

    def __init__(self, b1=0, b2=0, b3=0, c1=0, c2=0, c3=0):
        self.b1, self.b2, self.b3 = b1, b2, b3
        self.c1, self.c2, self.c3 = c1, c2, c3

    def coefficients(self):
        return (self.b1, self.b2, self.b3, self.c1, self.c2, self.c3)

    def index_at(self, wavelength):
        """Host-side convenience evaluation (used by ``abbe`` and by users placing optics);
        the traced index is computed on the device."""
        w2 = wavelength ** 2
        total = 1 + sum(
            (b * w2) / (w2 - c)
            for b, c in ((self.b1, self.c1), (self.b2, self.c2), (self.b3, self.c3))
        )
        return np.sqrt(total) if isinstance(total, np.ndarray) else math.sqrt(total)


absorber = _AbsorbingMaterial()
"""A bulk absorbing material"""

mirror = _ReflectingMaterial()
"""A perfectly reflecting material"""

# coefficients as published by SCHOTT, same presets the reference ships (materials.py:155-171)
glass = {
    "ideal": BasicRefractor(1.5),
    "BK7": SellmeierRefractor(
        1.03961212, 0.231792344, 1.01046945, 6.00069867e-3, 2.00179144e-2, 1.03560653e02
    ),
    "SF5": SellmeierRefractor(
        1.52481889, 0.187085527, 1.42729015, 0.011254756, 0.0588995392, 129.141675
    ),
    "SF2": SellmeierRefractor(
        1.40301821, 0.231767504, 0.939056586, 0.0105795466, 0.0493226978, 112.405955
    ),
}
"""A Dictionary of common glasses."""
