"""Traceable materials.

Same surface as the reference's ``pyrayt/materials.py``: ``absorber``, ``mirror``, the
``glass`` table, ``BasicRefractor``, ``SellmeierRefractor`` and ``Material.trace(surface,
ray_set)``.  A material here is a *description* (kind + up to six coefficients, see
``include/prt.h`` PRT_MAT_*); the shading arithmetic -- zeroing, reflecting about the world
normal, vector Snell refraction with the Sellmeier index (``materials.py:47-50, 58-62, 70-75,
136-145`` and ``tinygfx/g3d/operations.py:86-162``) -- is done per ray by the HIP kernels.
``trace()`` forwards to the engine so existing call sites keep working.

The reference's two extension points (``docs/source/reference/materials.rst:17-19``) work as upstream:

* a ``Glass`` subclass that supplies its own ``index_at(wavelength)`` (``materials.py:88-99``) -- a Cauchy
  glass, a measured dispersion curve -- is traced at full speed: ``index_at`` is evaluated on the host, on
  the distinct wavelengths of the ray set, and the kernels look every ray's wavelength up in that table
  (``PRT_MAT_TABLE``);
* a ``TracableMaterial`` subclass that supplies its own ``trace(surface, ray_set)`` (``materials.py:26-37``)
  is called exactly as upstream calls it (``_pyrayt.py:401-410``): each generation the rays that hit its
  surfaces are gathered on the device, handed to ``trace()`` as a host ``RaySet`` with their origins on the
  surface, and what it returns is the post-interaction state of those rays (``PRT_MAT_HOST``).  Everything
  else in the scene stays on the device path.

``device_kind(material)`` decides which of these a material object is.
"""
import math
from functools import lru_cache

import numpy as np

from .g3d.materials import gooch

# kind codes, must match include/prt.h PRT_MAT_*
NONE, ABSORBER, MIRROR, CONST_INDEX, SELLMEIER, TABLE, HOST = range(7)


class TracableMaterial(gooch.Material):
    kind = NONE
    _base_material = gooch.BLACK
    """What the renderers draw the material with (``materials.py:11-24``): absorbers black,
    mirrors and glasses blue."""

    def __init__(self, base_material=None, *args, **kwargs):
        """``base_material``: the material the renderers draw the object with (``materials.py:12-24``);
        default: the class's own (black; blue for mirrors and glasses)."""
        super().__init__(*args, **kwargs)
        if base_material is not None:
            self._base_material = base_material

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

    def __init__(self, *args, **kwargs):
        super().__init__(gooch.BLACK, *args, **kwargs)


class _ReflectingMaterial(TracableMaterial):
    """Ideal mirror, no change of index or intensity (``materials.py:53-62``)."""

    kind = MIRROR
    _base_material = gooch.BLUE

    def __init__(self, *args, **kwargs):
        super().__init__(gooch.BLUE, *args, **kwargs)


class Glass(TracableMaterial):
    """Refracting material with a wavelength dependent index (``materials.py:65-99``)."""

    _base_material = gooch.BLUE

    def __init__(self, *args, **kwargs):
        kwargs.setdefault("base_material", gooch.BLUE)
        super().__init__(*args, **kwargs)

    def index_at(self, wavelength):
        """Refractive index at ``wavelength`` (microns; a float or an array, the result has the argument's
        shape).  Abstract upstream (``materials.py:88-99``): a subclass that defines it is a glass the tracer
        can use -- see the module docstring."""
        raise NotImplementedError

    @lru_cache(100)
    def abbe(self):
        """Abbe number from the F, d and C lines (host-side helper, ``materials.py:77-86``)."""
        n_f, n_d, n_c = (self.index_at(w) for w in (0.4861, 0.5893, 0.6563))
        return (n_d - 1) / (n_f - n_c)


class BasicRefractor(Glass):
    """Non-dispersive glass (``materials.py:102-118``)."""

    kind = CONST_INDEX

    def __init__(self, refractive_index, *args, **kwargs):
        self._refractive_index = refractive_index
        super().__init__(*args, **kwargs)

    def coefficients(self):
        return (self._refractive_index,)

    def index_at(self, wavelength):
        if isinstance(wavelength, np.ndarray):
            return np.full(wavelength.shape, self._refractive_index)
        return self._refractive_index


class SellmeierRefractor(Glass):
    """n(w)^2 = 1 + sum_i b_i w^2 / (w^2 - c_i), w in microns (``materials.py:121-145``)."""

    kind = SELLMEIER

    def __init__(self, b1=0, b2=0, b3=0, c1=0, c2=0, c3=0, *args, **kwargs):
        super().__init__(*args, **kwargs)
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


def _defined_by(material, name):
    """The class in ``type(material)``'s MRO that defines attribute ``name`` (None if nobody does)."""
    for cls in type(material).__mro__:
        if name in vars(cls):
            return cls
    return None


def device_kind(material, shading_only=False):
    """Which PRT_MAT_* kind serves ``material`` (``include/prt.h``).

    * no ``trace`` at all (the default render-only ``GoochMaterial``, ``world_objects.py:341``): ``NONE`` --
      hitting such a surface raises, as upstream's missing attribute does;
    * ``trace`` defined outside this module -- a user's material, subclass of ``TracableMaterial`` or not
      (upstream duck-types: ``_pyrayt.py:408`` only ever calls ``surface.material.trace``): ``HOST``;
    * a glass whose ``index_at`` is not one of the two built-in ones (a direct ``Glass`` subclass, or a
      subclass of ``BasicRefractor`` / ``SellmeierRefractor`` that overrides it): ``TABLE``;
    * otherwise the class's built-in kind.

    ``shading_only``: the kind that serves ``TracableMaterial.trace`` itself, i.e. ignoring a ``trace``
    override -- what a user's ``trace()`` reaches through ``super().trace(surface, ray_set)``."""
    if not callable(getattr(material, "trace", None)):
        return NONE
    if not isinstance(material, TracableMaterial):
        return NONE if shading_only else HOST
    if not shading_only and _defined_by(material, "trace") is not TracableMaterial:
        return HOST
    if isinstance(material, Glass):
        closed_form = {BasicRefractor: CONST_INDEX, SellmeierRefractor: SELLMEIER}
        owner = _defined_by(material, "index_at")
        # (a subclass that only changes numbers keeps the closed form; one that redefines index_at or the
        # coefficient packing does not)
        if owner in closed_form and _defined_by(material, "coefficients") is owner:
            return closed_form[owner]
        return TABLE
    if isinstance(material, _ReflectingMaterial):
        return MIRROR
    if isinstance(material, _AbsorbingMaterial):
        return ABSORBER
    return NONE


def table_indices(material, wavelengths):
    """``material.index_at`` on an ascending float64 array of wavelengths, as a float64 array of the same
    shape (upstream calls it with the ray set's wavelength row, ``materials.py:72-73``: an array in, an array
    or a scalar out)."""
    wavelengths = np.asarray(wavelengths, dtype=float)
    out = np.asarray(material.index_at(wavelengths.copy()), dtype=float)
    return np.ascontiguousarray(np.broadcast_to(out, wavelengths.shape))


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
