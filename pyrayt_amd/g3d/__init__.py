"""pyrayt_amd.g3d -- scene-graph layer (the reference's ``tinygfx.g3d`` names).

The objects the ray-propagation hot path consumes -- transforms, traceable surfaces, CSG
nodes -- plus the second consumer of the intersect kernels: the
orthographic camera, Gooch materials and the two renderers of ``tinygfx/g3d/renderers.py``.
"""
from . import operations
from . import primitives
from .operations import binomial_root, element_wise_dot, reflect, refract, smallest_positive_root
from . import shapes
from .shapes import HomogeneousCoordinate, Point, Ray, Vector, bundle_of_rays, bundle_rays
from . import materials
from . import objects
from .objects import (
    BLACK,
    CountedObject,
    Cuboid,
    Cylinder,
    Intersectable,
    ObjectGroup,
    OrthographicCamera,
    Paraboloid,
    Sphere,
    TracerSurface,
    WorldObject,
    XYPlane,
    bounding_box,
)
from . import world_objects
from . import csg
from . import renderers

__all__ = [
    "BLACK", "CountedObject", "Cuboid", "Cylinder", "HomogeneousCoordinate", "Intersectable", "ObjectGroup",
    "OrthographicCamera", "Paraboloid", "Point", "Ray", "Sphere", "TracerSurface", "Vector", "WorldObject",
    "XYPlane", "binomial_root", "bounding_box", "bundle_of_rays", "bundle_rays", "csg", "element_wise_dot", "materials",
    "objects", "operations", "primitives", "reflect", "refract", "renderers", "shapes", "smallest_positive_root", "world_objects",
]
