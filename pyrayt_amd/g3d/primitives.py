"""Shape primitives under the reference's names (``tinygfx/g3d/primitives.py``).

``Sphere``, ``Paraboloid``, ``Plane``, ``Cube`` and ``Cylinder`` take upstream's constructor arguments
and offer upstream's ``intersect(rays)`` / ``normal(points)`` in the shape's own coordinate frame,
evaluated by the device routines the trace is built from (``primitive_pair`` / ``object_normal`` in
``csrc/prt_device.hpp``); ``Point``, ``Vector``, ``Ray``, ``bundle_of_rays`` and ``bundle_rays`` are the
small host-side carriers.  The 2-D helpers (``Disk``, ``Rectangle``) and ``overlap`` of upstream's module
are not on any path this package serves and are not provided.
"""
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
    "Cube", "Cylinder", "HomogeneousCoordinate", "Paraboloid", "Plane", "Point", "Ray", "Sphere",
    "SurfacePrimitive", "Vector", "bundle_of_rays", "bundle_rays",
]
