"""Primitive shape records.

Host-side these carry only what the device needs (a kind code and up to six parameters) plus
the object-space bounding corners used to derive world-space bounding boxes.  All
intersection / normal arithmetic for these shapes lives in the HIP kernels
(``pyrayt_amd/csrc``); the reference computes it in numpy at
``tinygfx/g3d/primitives.py`` (Sphere :220-296, Paraboloid :299-419, Plane :422-498,
Cube :501-602, Cylinder :621-741).
"""
import itertools

import numpy as np

from ._epoch import SceneEpoch

# kind codes, must match include/prt.h PRT_PRIM_*
SPHERE, CYLINDER, PLANE, CUBE, PARABOLOID = range(5)


def bundle_of_rays(n_rays):
    """(2,4,n) ray block, origins (0,0,0,1) and zero directions (primitives.py:18-28)."""
    block = np.zeros((2, 4, n_rays))
    block[0, 3] = 1.0
    return block


def bundle_rays(rays):
    """Stack single (2,4) rays into a (2,4,n) block (primitives.py:31-32)."""
    return np.stack(rays, axis=2)


def _named(index):
    return property(lambda self: self[index], lambda self, value: self.__setitem__(index, value))


class HomogeneousCoordinate(np.ndarray):
    """A float64 4-vector with named x / y / z / w components (primitives.py:35-82)."""

    def __new__(cls, x=0.0, y=0.0, z=0.0, w=0.0):
        return np.array((x, y, z, w), dtype=float).view(cls)

    x, y, z, w = _named(0), _named(1), _named(2), _named(3)

    def normalize(self):
        """Scale the xyz part to unit length, in place."""
        self[:-1] /= np.linalg.norm(self[:-1])
        return self


class Point(HomogeneousCoordinate):
    """Homogeneous point, w = 1 (primitives.py:85-89)."""

    def __new__(cls, x=0.0, y=0.0, z=0.0, *args, **kwargs):
        return super().__new__(cls, x, y, z, 1.0)


class Vector(HomogeneousCoordinate):
    """Homogeneous vector, w = 0 (primitives.py:92-94)."""

    def __new__(cls, x=0.0, y=0.0, z=0.0, *args, **kwargs):
        return super().__new__(cls, x, y, z, 0.0)


class Ray(np.ndarray):
    """One ray as a (2,4) block: row 0 the origin, row 1 the direction (primitives.py:97-122)."""

    def __new__(cls, origin=None, direction=None):
        ray = np.zeros((2, 4), dtype=float).view(cls)
        ray[0] = Point() if origin is None else origin
        ray[1] = Vector(1, 0, 0) if direction is None else direction
        return ray

    @property
    def origin(self):
        return self[0].view(HomogeneousCoordinate)

    @origin.setter
    def origin(self, value):
        self[0] = value

    @property
    def direction(self):
        return self[1].view(HomogeneousCoordinate)

    @direction.setter
    def direction(self, value):
        self[1] = value


def box_corners(lo, hi):
    """The eight homogeneous corner points (4,8) of the box spanned by two corners."""
    spans = np.sort(np.vstack((np.asarray(lo, float)[:3], np.asarray(hi, float)[:3])), axis=0).T
    pts = [(x, y, z, 1.0) for x, y, z in itertools.product(*spans)]
    return np.array(pts, dtype=float).T


class AxisBox:
    """Axis-aligned box given by per-axis (min, max) spans, shape (3,2).

    Plays the role of the reference's ``primitives.Cube`` when it is used as a bounding
    volume (``world_objects.py:15-23``, ``csg.py:93-116``)."""

    def __init__(self, lo, hi):
        self.axis_spans = np.sort(
            np.vstack((np.asarray(lo, float)[:3], np.asarray(hi, float)[:3])), axis=0
        ).T

    @classmethod
    def around(cls, points):
        """Smallest box containing a (4,k) or (3,k) point set (world_objects.py:15-23)."""
        return cls(np.min(points[:3], axis=1), np.max(points[:3], axis=1))

    @property
    def bounding_points(self):
        return box_corners(self.axis_spans[:, 0], self.axis_spans[:, 1])

    def flat(self):
        """xmin,xmax,ymin,ymax,zmin,zmax."""
        return [float(v) for v in self.axis_spans.reshape(-1)]


class Shape:
    """kind + params + object-space bounds of one primitive."""

    kind = -1

    def __setattr__(self, name, value):  # (a shape's parameters are scene data: assigning them counts as a change)
        object.__setattr__(self, name, value)
        SceneEpoch.value += 1

    def __init__(self, params, lo, hi):
        self.params = tuple(float(p) for p in params)
        self.bounding_points = box_corners(lo, hi)

    def packed_params(self):
        return list(self.params) + [0.0] * (6 - len(self.params))

    def intersect(self, rays):
        """(2, n) ray parameters where the (2,4) ray / (2,4,n) rays meet the shape in its own
        coordinate frame: the raw pair of the reference's ``primitive.intersect`` -- not sorted,
        +inf for a miss.  HIP engine (``prt_primitive_intersect``)."""
        from .. import engine

        block = np.atleast_3d(np.asarray(rays, dtype=float)).reshape(8, -1)
        return engine.ops_primitive("prt_primitive_intersect", self.kind, self.packed_params(), block, 2)

    def normal(self, intersections):
        """Unit object-space normal(s) (w = 0) at a (4,) point or (4, n) points assumed to lie on the
        shape.  HIP engine (``prt_primitive_normal``)."""
        from .. import engine

        points = np.asarray(intersections, dtype=float)
        if points.ndim not in (1, 2):
            raise AttributeError(
                f"Argument intersections has too many dimensions, expect 1 or 2, got {points.ndim}")
        out = engine.ops_primitive("prt_primitive_normal", self.kind, self.packed_params(),
                                   points.reshape(points.shape[0], -1), 4)
        return out[:, 0] if points.ndim == 1 else out


class SphereShape(Shape):
    kind = SPHERE

    def __init__(self, radius=1):
        super().__init__((radius,), (-radius,) * 3, (radius,) * 3)

    def get_radius(self):
        return self.params[0]


class CylinderShape(Shape):
    kind = CYLINDER

    def __init__(self, radius=1, min_height=-1, max_height=1):
        super().__init__(
            (radius, min_height, max_height),
            (-radius, -radius, min_height),
            (radius, radius, max_height),
        )

    def get_radius(self):
        return self.params[0]


class PlaneShape(Shape):
    kind = PLANE

    def __init__(self, width=2, length=2):
        # the reference pads the flat patch by +-0.01 in z for its bounds (primitives.py:431-434)
        super().__init__(
            (width, length), (-width / 2, -length / 2, -0.01), (width / 2, length / 2, 0.01)
        )


class CubeShape(Shape):
    kind = CUBE

    def __init__(self, min_corner=(-1, -1, -1), max_corner=(1, 1, 1)):
        box = AxisBox(min_corner, max_corner)
        self.axis_spans = box.axis_spans
        super().__init__(box.flat(), box.axis_spans[:, 0], box.axis_spans[:, 1])


class ParaboloidShape(Shape):
    kind = PARABOLOID

    def __init__(self, focus=1, height=1):
        if focus <= 0 or height <= 0:
            # same guard as primitives.py:306-307
            raise ValueError("Focus and height must be positive numbers")
        rim = np.sqrt(4 * focus * height)
        super().__init__((focus, height), (-rim, -rim, -0.0), (rim, rim, height))

    def get_focus(self):
        return self.params[0]
