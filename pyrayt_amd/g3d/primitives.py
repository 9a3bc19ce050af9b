"""Shape primitives under the reference's names (``tinygfx/g3d/primitives.py``).

``Sphere``, ``Paraboloid``, ``Plane``, ``Cube`` and ``Cylinder`` take upstream's constructor arguments
and offer upstream's ``intersect(rays)`` / ``normal(points)`` in the shape's own coordinate frame,
evaluated by the device routines the trace is built from (``primitive_pair`` / ``object_normal`` in
``csrc/prt_device.hpp``); ``Point``, ``Vector``, ``Ray``, ``bundle_of_rays`` and ``bundle_rays`` are the
small host-side carriers.  The 2-D helpers of upstream's module (``Shape2D``, ``Disk``, ``Rectangle``, ``overlap``;
``primitives.py:163-217, 605-618``) are on no traced path -- nothing upstream calls them either -- and are plain numpy
here, kept so that the module's names are all there.
"""
import abc

import numpy as np

from .shapes import (
    CubeShape as Cube,
    CylinderShape as Cylinder,
    HomogeneousCoordinate,
    ParaboloidShape as Paraboloid,
    PlaneShape as Plane,
    Point,
    Ray,
    Shape as SurfacePrimitive,
    SphereShape as Sphere,
    Vector,
    bundle_of_rays,
    bundle_rays,
)

__all__ = [
    "Cube", "Cylinder", "Disk", "HomogeneousCoordinate", "Paraboloid", "Plane", "Point", "Ray", "Rectangle", "Shape2D",
    "Sphere", "SurfacePrimitive", "Vector", "bundle_of_rays", "bundle_rays", "overlap",
]


class Shape2D(abc.ABC):
    """A region of the xy plane (``primitives.py:163-176``)."""

    @abc.abstractmethod
    def point_in_shape(self, points):
        """points: (2,) or (2, n) -> bool or (n,) bools, True inside or on the edge."""


class Disk(Shape2D):
    """Centred disk (``primitives.py:179-195``)."""

    def __init__(self, radius=1.0):
        self._radius = radius

    @classmethod
    def from_diameter(cls, diameter):
        return cls(diameter / 2)

    def point_in_shape(self, points):
        return np.hypot(*np.asarray(points, dtype=float)[:2]) <= self._radius


class Rectangle(Shape2D):
    """Centred, axis-aligned rectangle (``primitives.py:198-217``)."""

    def __init__(self, x_length=2, y_length=2):
        self._x_length = x_length
        self._y_length = y_length

    def point_in_shape(self, points):
        x, y = np.asarray(points, dtype=float)[:2]
        inside = (np.abs(x) <= self._x_length / 2) & (np.abs(y) <= self._y_length / 2)
        return bool(inside) if np.ndim(inside) == 0 else inside


def overlap(arr1, arr2):
    """The members of two 1-D arrays that lie in the intersection of their value ranges (``primitives.py:605-618``;
    like upstream, arrays of more dimensions give None)."""
    arr1, arr2 = np.asarray(arr1), np.asarray(arr2)
    if arr1.ndim != 1:
        return None
    lo, hi = max(arr1.min(), arr2.min()), min(arr1.max(), arr2.max())
    return arr1[(arr1 >= lo) & (arr1 <= hi)], arr2[(arr2 >= lo) & (arr2 <= hi)]
