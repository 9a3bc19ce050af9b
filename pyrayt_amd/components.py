"""Optical parts and ray sources (host-side scene construction).

These build the same scene graphs as the reference's ``pyrayt/components.py`` so that the
benchmark configurations can be expressed with the API the reference's
users already know.  Nothing here is on the per-ray hot path: parts are O(1) trees of
``pyrayt_amd.g3d`` objects, sources fill an initial ``RaySet`` once per trace.

Part orientation convention (``components.py:13-28, 232-247``): every lens / mirror is
modelled with its optical axis along object-space +z and then turned by ``rotate_y(90)``
followed by ``rotate_x(90)`` so that the axis is world +x and the aperture lies in the yz
plane.
"""
from functools import lru_cache

import numpy as np

from . import g3d as cg
from . import materials as matl
from .rayset import RaySet


# --------------------------------------------------------------------------------------------
# apertures and helpers
# --------------------------------------------------------------------------------------------
def _is_pair(aperture):
    return hasattr(aperture, "__len__")


def _create_aperture(aperture, thickness):
    """Aperture stock: a float is the diameter of a circular aperture (a cylinder), a pair of
    positive numbers the side lengths of a rectangular one (a cuboid)
    (``components.py:31-53``; the elliptical branch upstream returns None and is not
    reproduced)."""
    if not _is_pair(aperture):
        return cg.Cylinder(radius=aperture / 2, min_height=-thickness / 2, max_height=thickness / 2)
    if aperture[0] > 0 and aperture[1] > 0:
        half = (aperture[0] / 2, aperture[1] / 2, thickness / 2)
        return cg.Cuboid(tuple(-h for h in half), half)
    raise TypeError(f"Could not deduce an aperture from {aperture}")


def _aperture_reach(aperture):
    """Largest distance from the optical axis still inside the aperture."""
    return np.linalg.norm(aperture) / 2 if _is_pair(aperture) else aperture / 2


def _sag(radius, reach):
    """Depth of a spherical cap of curvature ``radius`` at distance ``reach`` off axis."""
    return np.abs(radius) - np.sqrt(np.abs(radius) ** 2 - reach ** 2)


def _lens_full_thickness(r1, r2, thickness, aperture):
    """(edge-to-edge thickness of the aperture stock, its shift along the axis) for a thick
    lens whose concave faces bulge outwards past the vertex planes (``components.py:130-163``)."""
    reach = _aperture_reach(aperture)
    left = thickness / 2
    if np.isfinite(r1) and r1 < 0:
        left += _sag(r1, reach)
    right = thickness / 2
    if np.isfinite(r2) and r2 > 0:
        right += _sag(r2, reach)
    return right + left, right - left


def _to_optical_axis(part):
    return part.rotate_y(90).rotate_x(90)


_LENS_DEFAULTS = {"aperture": 1, "material": matl.glass["ideal"]}
_MIRROR_DEFAULTS = {"aperture": 1, "material": matl.mirror, "off_axis": (0, 0)}


# --------------------------------------------------------------------------------------------
# lenses
# --------------------------------------------------------------------------------------------
def thick_lens(r1, r2, thickness, **kwargs):
    """Thick lens with arbitrary spherical faces (``components.py:73-127``).  Radii follow the
    optics sign convention; ``np.inf`` gives a flat face.  Built as the aperture stock
    intersected with (convex) or minus (concave) one sphere per curved face."""
    opts = {**_LENS_DEFAULTS, **kwargs}
    material = opts["material"]
    stock_thickness, stock_shift = _lens_full_thickness(r1, r2, thickness, opts["aperture"])

    lens = _create_aperture(opts["aperture"], stock_thickness).move_z(stock_shift / 2)
    lens.material = material

    if np.isfinite(r1):
        face = cg.Sphere(r1, material=material).move_z(r1 - thickness / 2)
        lens = cg.csg.intersect(lens, face) if r1 > 0 else cg.csg.difference(lens, face)
    if np.isfinite(r2):
        face = cg.Sphere(r2, material=material).move_z(r2 + thickness / 2)
        lens = cg.csg.intersect(lens, face) if r2 < 0 else cg.csg.difference(lens, face)
    return _to_optical_axis(lens)


def biconvex_lens(r1, r2, thickness, **kwargs):
    """Lens with two convex faces: (sphere & sphere) & aperture stock
    (``components.py:166-198``).  Upstream gives the first sphere radius ``r2`` but offsets it
    by ``r1`` (``:185``); kept, it only matters when r1 != r2."""
    opts = {**_LENS_DEFAULTS, **kwargs}
    stock = _create_aperture(opts["aperture"], thickness)
    first = cg.Sphere(r2).move_z(r1 - thickness / 2)
    second = cg.Sphere(r1).move_z(-(r1 - thickness / 2))
    for part in (stock, first, second):
        part.material = opts["material"]
    return _to_optical_axis(cg.csg.intersect(cg.csg.intersect(first, second), stock))


def plano_convex_lens(r, thickness, **kwargs):
    """Flat face towards -x, spherical face towards +x (``components.py:201-229``)."""
    opts = {**_LENS_DEFAULTS, **kwargs}
    stock = _create_aperture(opts["aperture"], thickness)
    face = cg.Sphere(r).move_z(-(r - thickness / 2))
    stock.material = opts["material"]
    face.material = opts["material"]
    return _to_optical_axis(cg.csg.intersect(face, stock))


# --------------------------------------------------------------------------------------------
# mirrors
# --------------------------------------------------------------------------------------------
def plane_mirror(thickness, **kwargs):
    """Slab whose every face reflects (``components.py:250-266``)."""
    opts = {**_MIRROR_DEFAULTS, **kwargs}
    slab = _create_aperture(opts["aperture"], thickness).move(*opts["off_axis"], 0)
    slab.material = opts["material"]
    return _to_optical_axis(slab)


def spherical_mirror(radius, thickness, **kwargs):
    """Absorbing stock minus a reflecting sphere (``components.py:269-321``)."""
    opts = {**_MIRROR_DEFAULTS, **kwargs}
    off_axis = opts["off_axis"]
    reach = np.sqrt(off_axis[0] ** 2 + off_axis[1] ** 2) + _aperture_reach(opts["aperture"])
    front = abs(radius) - np.sqrt(radius ** 2 - reach ** 2)
    total = front + thickness

    stock = _create_aperture(opts["aperture"], thickness + front)
    stock.material = matl.absorber
    stock.move(*off_axis, 0)
    if radius > 0:
        bowl = cg.Sphere(radius, material=opts["material"]).move_z(radius)
        stock.move_z(total / 2 - thickness)
    elif radius < 0:
        bowl = cg.Sphere(abs(radius), material=opts["material"]).move_z(radius)
        stock.move_z(thickness - total / 2)
    else:
        raise ValueError("mirror radius must be non-zero")
    return _to_optical_axis(cg.csg.difference(stock, bowl))


def parabolic_mirror(focus, thickness, **kwargs):
    """Absorbing stock minus a reflecting paraboloid, focus at the origin
    (``components.py:350-398``)."""
    opts = {**_MIRROR_DEFAULTS, **kwargs}
    off_axis, aperture = opts["off_axis"], opts["aperture"]
    if _is_pair(aperture):
        reach = np.linalg.norm(np.abs(np.asarray(off_axis)) + np.asarray(aperture) / 2)
    else:
        reach = np.linalg.norm(np.asarray(off_axis)) + aperture
    front = 1 / (4 * focus) * reach ** 2
    total = thickness + front

    stock = _create_aperture(aperture, total).move(*off_axis, 0)
    stock.material = matl.absorber
    stock.move_z(total / 2 - thickness)
    dish = cg.Paraboloid(focus, height=1.5 * front, material=opts["material"])
    mirror = cg.csg.difference(stock, dish)
    mirror.move_z(-focus)
    return _to_optical_axis(mirror)


# --------------------------------------------------------------------------------------------
# prism, baffle, aperture stop
# --------------------------------------------------------------------------------------------
def equilateral_prism(side_length, width, material=matl.glass["BK7"]):
    """A cube with two wedges cut away at +-30 degrees (``components.py:401-436``)."""
    sin60 = np.sin(60 * np.pi / 180)
    cut = 1.1 * side_length / sin60

    def wedge(sign):
        return (
            cg.Cuboid.from_sides(cut, 1.1 * width, cut, material=material)
            .move(sign * cut / 2, 0, cut / 2)
            .rotate_y(sign * -30)
            .move(sign * side_length / 2, 0, -side_length / 2)
        )

    body = cg.Cuboid.from_sides(side_length, width, side_length, material=material)
    prism = cg.csg.difference(cg.csg.difference(body, wedge(-1)), wedge(+1))
    return prism.move_z(side_length / 2 * (1 - sin60))


def baffle(aperture):
    """Absorbing rectangle in the yz plane (``components.py:439-448``)."""
    return cg.XYPlane(aperture[0], aperture[1], material=matl.absorber).rotate_y(90)


def aperture(size, aperture_size):
    """Absorbing rectangle with a hole (``components.py:451-468``).  The hole stock keeps the
    default untracable material, as upstream."""
    stop = baffle(size).rotate_y(-90)
    hole = _create_aperture(aperture_size, thickness=0.1)
    return cg.csg.difference(stop, hole).rotate_y(90).rotate_x(-90)


# --------------------------------------------------------------------------------------------
# sources
# --------------------------------------------------------------------------------------------
class Source(cg.WorldObject):
    """Emits a RaySet: object-space pattern -> world transform -> unit directions
    (``components.py:471-508``)."""

    def __init__(self, wavelength=0.633, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._wavelength = wavelength

    @property
    def wavelength(self):
        return self._wavelength

    @wavelength.setter
    def wavelength(self, value):
        self._wavelength = value

    def _local_ray_generation(self, n_rays):
        raise NotImplementedError

    def generate_rays(self, n_rays):
        ray_set = self._local_ray_generation(n_rays)
        ray_set.rays = np.matmul(self._world, ray_set.rays)
        ray_set.rays[1] /= np.linalg.norm(ray_set.rays[1], axis=0)
        return ray_set

    def _blank(self, n_rays):
        ray_set = RaySet(n_rays)
        ray_set.wavelength = self._wavelength
        return ray_set

    def device_spec(self):
        """(kind, params, seed) if the HIP engine can emit this source's rays on the GPU
        (``prt_generate_rays``), else None: a user-defined source is generated on the host by
        its own ``_local_ray_generation`` and uploaded."""
        return None


class LineOfRays(Source):
    """Parallel rays along +x spread over ``spacing`` on the y axis (``components.py:511-530``)."""

    def __init__(self, spacing=1, wavelength=0.633, *args, **kwargs):
        super().__init__(wavelength, *args, **kwargs)
        self._spacing = spacing

    def _local_ray_generation(self, n_rays):
        ray_set = self._blank(n_rays)
        if n_rays > 1:
            ray_set.rays[0, 1] = np.linspace(-self._spacing / 2, self._spacing / 2, n_rays)
        ray_set.rays[1, 0] = 1
        return ray_set

    def device_spec(self):
        return 0, (self._spacing,), 0


class CircleOfRays(Source):
    """Parallel rays along +x on a circle in the yz plane (``components.py:533-558``)."""

    def __init__(self, diameter=1, wavelength=0.633, *args, **kwargs):
        super().__init__(wavelength, *args, **kwargs)
        self._diameter = diameter

    def _local_ray_generation(self, n_rays):
        ray_set = self._blank(n_rays)
        theta = np.linspace(0, 2 * np.pi, n_rays)
        ray_set.rays[0, 1] = self._diameter / 2 * np.sin(theta)
        ray_set.rays[0, 2] = self._diameter / 2 * np.cos(theta)
        ray_set.rays[1, 0] = 1
        return ray_set

    def device_spec(self):
        return 1, (self._diameter,), 0


class ConeOfRays(Source):
    """Rays from one point on a cone of half-angle ``cone_angle`` degrees about +x
    (``components.py:561-585``)."""

    def __init__(self, cone_angle, wavelength=0.633, *args, **kwargs):
        super().__init__(wavelength, *args, **kwargs)
        self._angle = cone_angle * np.pi / 180.0

    def _local_ray_generation(self, n_rays):
        ray_set = self._blank(n_rays)
        if n_rays > 1:
            azimuth = 2 * np.pi * np.arange(0, n_rays) / n_rays
            ray_set.rays[1, 1] = np.sin(self._angle) * np.sin(azimuth)
            ray_set.rays[1, 2] = np.sin(self._angle) * np.cos(azimuth)
        ray_set.rays[1, 0] = np.cos(self._angle)
        return ray_set

    def device_spec(self):
        return 2, (self._angle,), 0


class WedgeOfRays(Source):
    """Fan of rays in the xy plane spanning ``angle`` degrees (``components.py:588-613``)."""

    def __init__(self, angle, wavelength=0.633, *args, **kwargs):
        super().__init__(wavelength, *args, **kwargs)
        self._angle = angle * np.pi / 180.0

    def _local_ray_generation(self, n_rays):
        ray_set = self._blank(n_rays)
        fan = np.linspace(-self._angle / 2, self._angle / 2, n_rays)
        ray_set.rays[1, 0] = np.cos(fan)
        ray_set.rays[1, 1] = np.sin(fan)
        return ray_set

    def device_spec(self):
        return 3, (self._angle,), 0


class Lamp(Source):
    """Lambertian emitter over a width x length rectangle in the yz plane
    (``components.py:616-654``); random, drawn from the global numpy RNG like upstream."""

    def __init__(self, width, length, max_angle=90, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._max_angle = max_angle * np.pi / 180
        self._width = width
        self._length = length

    def _local_ray_generation(self, n_rays):
        ray_set = self._blank(n_rays)
        # inverse-CDF sampling of the polar angle (components.py:56-70)
        uv = np.random.random_sample((2, n_rays))
        theta = np.arccos(1 - uv[0] * (1 - np.cos(self._max_angle)))
        phi = uv[1] * 2 * np.pi
        ray_set.rays[0, 1] = self._width * (np.random.random_sample(n_rays) - 0.5)
        ray_set.rays[0, 2] = self._length * (np.random.random_sample(n_rays) - 0.5)
        ray_set.rays[1, 0] = np.cos(theta)
        ray_set.rays[1, 1] = np.sin(theta) * np.cos(phi)
        ray_set.rays[1, 2] = np.sin(theta) * np.sin(phi)
        ray_set.intensity = 100.0 * np.cos(theta)
        return ray_set

    def device_spec(self):
        # a fresh stream per call, like upstream's draws from the global RNG
        seed = int(np.random.randint(0, 2 ** 62))
        return 4, (self._width, self._length, self._max_angle), seed


class StaticLamp(Lamp):
    """A Lamp that returns the same rays for the same ``n_rays`` (``components.py:657-662``)."""

    @lru_cache(10)
    def generate_rays(self, n_rays):
        return super().generate_rays(n_rays)

    def device_spec(self):
        return None  # must replay the cached host rays (components.py:657-662)
