"""pyrayt_amd.g3d -- scene-graph layer (the reference's ``tinygfx.g3d`` names).

Only the objects the ray-propagation hot path consumes are provided: transforms, traceable
surfaces, CSG nodes.  Renderers, cameras and Gooch shading (``tinygfx/g3d/renderers.py``,
``materials/``) are out of scope.
"""
from . import shapes
from .shapes import Point, Vector, bundle_of_rays
from . import objects
from .objects import (
    BLACK,
    CountedObject,
    Cuboid,
    Cylinder,
    Intersectable,
    ObjectGroup,
    Paraboloid,
    Sphere,
    TracerSurface,
    WorldObject,
    XYPlane,
)
from . import csg

__all__ = [
    "BLACK", "CountedObject", "Cuboid", "Cylinder", "Intersectable", "ObjectGroup", "Paraboloid", "Point",
    "Sphere", "TracerSurface", "Vector", "WorldObject", "XYPlane", "bundle_of_rays", "csg",
    "objects", "shapes",
]
