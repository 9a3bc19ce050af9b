"""Scene-graph objects: transforms, ids, traceable surfaces.

Host-side counterpart of the reference's ``tinygfx/g3d/world_objects.py``.  These classes only
do the O(#surfaces) bookkeeping the hot path consumes as *inputs* (SURVEY.md section 8 row
a15): the 4x4 world transform chain and its inverse (``world_objects.py:100,122-129``), the
process-global id counter (``:26-40``), normal inversion (``:305,319-323``) and world-space
bounding boxes (``:15-23,348-358``).  The per-ray work -- ``intersect`` (``:360-383``) and
``get_world_normals`` (``:401-418``) -- is executed by the HIP engine; the methods here just
hand the call to it.
"""
import collections
import copy
from itertools import count

import numpy as np

from . import shapes
from ._epoch import SceneEpoch  # noqa: F401  (the scene objects' change counter; re-exported)
from .materials import gooch

_UNITS = {"deg": np.pi / 180.0, "rad": 1.0}


class CountedObject:
    """Every scene object draws its id from one process-global counter
    (``world_objects.py:26-40``); CSG nodes and sources consume ids too."""

    _ids = count(0)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._id = next(CountedObject._ids)

    def get_id(self):
        return self._id

    @staticmethod
    def reset_ids(start=0):
        """Restart the global counter (test / fixture helper; the reference restarts it only
        by starting a fresh interpreter)."""
        CountedObject._ids = count(start)


class WorldObject(CountedObject):
    """An object with a 4x4 homogeneous object->world transform.

    New transforms are applied on the *left* of the accumulated matrix and the inverse is
    re-derived with ``np.linalg.inv`` after every change, exactly as the reference does
    (``world_objects.py:122-129``) so the matrices handed to the device are the same floats.
    """

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        SceneEpoch.value += 1

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._world = np.identity(4, dtype=float)
        self._object = np.identity(4, dtype=float)
        # functions called (without arguments) after every change of the world transform
        # (world_objects.py:98-100, 122-124); CSG nodes hang their bounding-box refresh in here
        self.var_watchlist = []

    # --- matrix access -------------------------------------------------------------------
    def get_world_transform(self):
        return copy.copy(self._world)

    def get_object_transform(self):
        return copy.copy(self._object)

    def to_object_coordinates(self, coordinates):
        return np.matmul(self._object, coordinates)

    def to_world_coordinates(self, coordinates):
        return np.matmul(self._world, coordinates)

    def get_position(self):
        return np.matmul(self._world, shapes.Point(0, 0, 0))

    def get_quaternion(self):
        """The rotation part of the world transform as an (x, y, z, w) quaternion
        (``world_objects.py:156-160``)."""
        from scipy.spatial import transform

        return transform.Rotation.from_matrix(self._world[:-1, :-1]).as_quat()

    def get_orientation(self):
        axis = np.matmul(self._world, shapes.Vector(0, 0, 1))
        length = np.linalg.norm(axis)
        if length < 1e-7:
            raise ValueError(f"Measured Norm of World Vector below tolerance: {length}")
        return axis / length

    # --- the one mutation point -------------------------------------------------------------
    def _append_world_transform(self, matrix):
        self._world = np.matmul(matrix, self._world)
        if np.linalg.norm(np.matmul(self._world, shapes.Vector(0, 0, 1))) < 1e-7:
            # world_objects.py:113-117
            raise ValueError("transform collapses the object's z axis")
        self._object = np.linalg.inv(self._world)
        for notify in self.var_watchlist:
            notify()

    def transform(self, matrix):
        self._append_world_transform(np.asarray(matrix, dtype=float))
        return self

    # --- translation ------------------------------------------------------------------------
    def move(self, x=0, y=0, z=0):
        shift = np.identity(4)
        shift[:3, 3] = (x, y, z)
        return self.transform(shift)

    def move_x(self, d):
        return self.move(x=d)

    def move_y(self, d):
        return self.move(y=d)

    def move_z(self, d):
        return self.move(z=d)

    # --- scaling ----------------------------------------------------------------------------
    def scale(self, x=1, y=1, z=1):
        if min(x, y, z) < 0:
            raise ValueError("Negative values for scale operations are prohibited")
        return self.transform(np.diag((x, y, z, 1)).astype(float))

    def scale_x(self, s):
        return self.scale(x=s)

    def scale_y(self, s):
        return self.scale(y=s)

    def scale_z(self, s):
        return self.scale(z=s)

    def scale_all(self, s):
        return self.scale(s, s, s)

    # --- rotation ---------------------------------------------------------------------------
    def _rotation(self, angle, units, i, j):
        """Rotation in the (i,j) coordinate plane; sign convention of world_objects.py:238-269."""
        if units not in _UNITS:
            raise ValueError(f"{units} is not a valid option for angle units")
        radians = angle * np.pi / 180.0 if units == "deg" else angle
        c, s = np.cos(radians), np.sin(radians)
        rot = np.identity(4)
        rot[i, i] = c
        rot[j, j] = c
        rot[i, j] = -s
        rot[j, i] = s
        return self.transform(rot)

    def rotate_x(self, angle, units="deg"):
        return self._rotation(angle, units, 1, 2)

    def rotate_y(self, angle, units="deg"):
        return self._rotation(angle, units, 2, 0)

    def rotate_z(self, angle, units="deg"):
        return self._rotation(angle, units, 0, 1)


class ObjectGroup(WorldObject, collections.UserList):
    """A list of world objects that move together: a transform applied to the group is applied
    to every member, groups nest (``world_objects.py:283-295``).  Host-side scene editing only."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)  # WorldObject state, then UserList's ``data``

    def _append_world_transform(self, matrix):
        super()._append_world_transform(matrix)
        for member in self.data:
            member.transform(matrix)


class pin:
    """Context manager: whatever is done to the pinned objects' transforms inside the ``with``
    block is undone on exit (``pyrayt/_pyrayt.py:539-575``).  Lets a design loop move parts,
    trace, and come back to the starting layout."""

    def __init__(self, *objects_to_pin):
        self._objects = objects_to_pin

    def __enter__(self):
        self._start = [obj.get_world_transform() for obj in self._objects]
        return self._objects

    def __exit__(self, exception_type, exception_value, traceback):
        for obj, start in zip(self._objects, self._start):
            change = np.matmul(obj.get_world_transform(), np.linalg.inv(start))
            obj.transform(np.linalg.inv(change))


class Intersectable(WorldObject):
    """Anything a RayTracer can hold as a component (``world_objects.py:298-335``)."""

    _normal_scale = 1

    def attach_to(self, parent_object):
        """Become a part of `parent_object` (``world_objects.py:315-317``): from now on a change of
        this object's transform also runs the functions the parent is watching *at this moment* --
        which is how a CSG node's cached bounding box follows its children.  Like upstream, what the
        parent starts watching later (being attached to a node of its own) is not passed down to
        parts attached earlier: the outer node of a right-nested tree that is moved after construction
        keeps the box it computed before its inner node's children had moved (see ``csg.CSGSurface``)."""
        self._parent = parent_object
        self.var_watchlist += parent_object.var_watchlist

    def invert_normals(self):
        self._normal_scale = -1

    def reset_normals(self):
        self._normal_scale = 1

    @property
    def surface_ids(self):
        return ((self.get_id(), self),)

    @property
    def bounding_box(self):
        raise NotImplementedError

    @property
    def bounding_volume(self):
        return self.bounding_box

    def intersect(self, rays):
        """``component.intersect(rays) -> (hits (m,n), surface ids (m,n))`` executed by the
        HIP engine (reference: ``world_objects.py:360-383`` / ``csg.py:118-160``)."""
        from .. import engine

        return engine.component_intersect(self, rays)


def bounding_box(point_set):
    """Axis-aligned box that contains a (3+, k) point set (``world_objects.py:15-23``)."""
    return shapes.AxisBox.around(np.asarray(point_set, dtype=float))


BLACK = gooch.BLACK
"""Default surface material (``world_objects.py:341``): render-only.  It has no ``trace``, so a
ray hitting such a surface in a RayTracer raises AttributeError, as upstream."""


class TracerSurface(Intersectable):
    """A primitive shape placed in the world, with a material (``world_objects.py:338-422``)."""

    shape_type = None
    surface = None  # upstream's name for the same class attribute (``world_objects.py:339``): either one may be set

    def __init__(self, surface_args, material=BLACK, *args, **kwargs):
        super().__init__(*args, **kwargs)
        cls = type(self)
        self._shape = (cls.shape_type or cls.surface)(*surface_args)
        self.material = material

    @property
    def primitive(self):
        return self._shape

    @property
    def bounding_box(self):
        return shapes.AxisBox.around(np.matmul(self._world, self._shape.bounding_points))

    def get_world_normals(self, positions):
        """World-space unit normals at (4,n) points; runs on the HIP engine
        (reference: ``world_objects.py:401-418``)."""
        from .. import engine

        return engine.surface_normals(self, positions)

    def shade(self, rays, distances, **kwargs):
        """(4,n) RGBA where the (2,4,n) ``rays`` meet this surface at parameters ``distances``,
        lit from ``light_positions`` (reference: ``world_objects.py:385-399``); HIP engine."""
        from .. import engine

        return engine.surface_shade(self, rays, distances, **kwargs)


class Sphere(TracerSurface):
    shape_type = surface = shapes.SphereShape

    def __init__(self, radius=1, material=BLACK, *args, **kwargs):
        super().__init__((radius,), material, *args, **kwargs)


class Cylinder(TracerSurface):
    shape_type = surface = shapes.CylinderShape

    def __init__(self, radius=1, min_height=-1, max_height=1, material=BLACK, *args, **kwargs):
        super().__init__((radius, min_height, max_height), material, *args, **kwargs)


class Paraboloid(TracerSurface):
    shape_type = surface = shapes.ParaboloidShape

    def __init__(self, focus=1, height=1, material=BLACK, *args, **kwargs):
        super().__init__((focus, height), material, *args, **kwargs)


class XYPlane(TracerSurface):
    shape_type = surface = shapes.PlaneShape

    def __init__(self, width=2, length=2, material=BLACK, *args, **kwargs):
        super().__init__((width, length), material, *args, **kwargs)


class Cuboid(TracerSurface):
    shape_type = surface = shapes.CubeShape

    def __init__(self, l_corner=(-1, -1, -1), r_corner=(1, 1, 1), material=BLACK, *args, **kwargs):
        super().__init__((l_corner, r_corner), material, *args, **kwargs)

    @classmethod
    def from_sides(cls, x=1, y=1, z=1, **kwargs):
        half = np.array((x, y, z), dtype=float) * 0.5
        return cls(-half, half, **kwargs)

    @classmethod
    def from_length(cls, length, **kwargs):
        return cls.from_sides(length, length, length, **kwargs)


class OrthographicCamera(WorldObject):
    """A v x h grid of parallel rays: the camera sits in its y-z plane and looks along its +x
    axis (``world_objects.py:499-537``).  Pixels run row-major from (+h/2, +v/2) to
    (-h/2, -v/2).  The rays are produced on the device (``prt_camera_rays``); a render never
    materialises them at all (``prt_render``)."""

    def __init__(self, h_pixel_count, h_width, aspect_ratio, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._h_pixels = h_pixel_count
        self._h_width = h_width
        self._v_width = aspect_ratio * h_width
        self._v_pixels = int(aspect_ratio * self._h_pixels)

    def get_resolution(self):
        return (self._h_pixels, self._v_pixels)

    def get_span(self):
        return (self._h_width, self._v_width)

    def generate_rays(self):
        """(2,4,n) world-space rays, one per pixel, unit directions (host array, as upstream)."""
        from .. import engine

        return engine.camera_rays(self).cpu().numpy().reshape(2, 4, -1)
