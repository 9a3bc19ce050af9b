"""Scene snapshot: flatten a list of components into the C structs of ``include/prt.h``.

The snapshot is the only thing the device ever sees of the scene graph (SURVEY.md section 8
row a15): per primitive the cached world->object matrix
(``tinygfx/g3d/world_objects.py:122-124``), shape parameters, normal sign (``:305,319-323``),
surface id (``:26-40``) and material; per CSG node its operation, children and world-space
cull box (``tinygfx/g3d/csg.py:93-116``); the component order of ``RayTracer._components``
(``pyrayt/_pyrayt.py:229-239``).  It is taken at trace time, so moving parts between traces
is picked up like upstream.
"""
import numpy as np

from . import materials as matl
from .g3d import csg as _csg
from .g3d.objects import BLACK, TracerSurface

PRIM_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("material", "<i4"),
        ("normal_scale", "<i4"),
        ("reserved", "<i4"),
        ("surface_id", "<i8"),
        ("params", "<f8", (6,)),
        ("minv", "<f8", (16,)),
    ],
    align=True,
)
NODE_DTYPE = np.dtype(
    [("op", "<i4"), ("left", "<i4"), ("right", "<i4"), ("prim", "<i4"), ("aabb", "<f8", (6,))],
    align=True,
)
MATERIAL_DTYPE = np.dtype(
    [("kind", "<i4"), ("reserved", "<i4"), ("coef", "<f8", (6,))], align=True
)
assert PRIM_DTYPE.itemsize == 200 and NODE_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 56

NODE_LEAF = 0  # inner nodes use Operation.value (1..3)


class SceneSnapshot:
    """Flat arrays describing ``components`` + the surface look-up table."""

    def __init__(self, components, material_override=None):
        """material_override: shade every surface with this material's own arithmetic instead of its
        ``material`` attribute (``Material.trace(surface, ray_set)`` called directly, possibly through
        ``super().trace`` from a user's ``trace()``: a ``trace`` override is not looked at then)."""
        if not hasattr(components, "__iter__"):
            components = (components,)
        self.components = tuple(components)
        prims, nodes, roots, mats = [], [], [], []
        self._material_slots = {}
        self.surfaces = []  # leaf surfaces in look-up-table order (_pyrayt.py:257-260)
        self.table_materials = []  # (material slot, material) of every user-defined glass (PRT_MAT_TABLE)
        self.host_surfaces = []    # (primitive index, surface) whose material.trace() is user code (PRT_MAT_HOST)

        def material_slot(material):
            key = id(material)
            if key not in self._material_slots:
                kind = matl.NONE if material is BLACK else matl.device_kind(
                    material, shading_only=material_override is not None)
                coef = [0.0] * 6
                if kind in (matl.ABSORBER, matl.MIRROR, matl.CONST_INDEX, matl.SELLMEIER):
                    coef = material.packed_coefficients()
                elif kind == matl.TABLE:
                    self.table_materials.append((len(mats), material))
                    # a NaN wavelength has no place in an ascending table: its index travels in coef[3]
                    # (include/prt.h) -- what upstream's index_at(NaN) answers, NaN if the glass will not say
                    try:
                        coef[3] = float(matl.table_indices(material, np.array([np.nan]))[0])
                    except Exception:  # noqa: BLE001
                        coef[3] = float("nan")
                self._material_slots[key] = len(mats)
                mats.append((kind, 0, coef))
            return self._material_slots[key]

        def add(obj):
            """Post-order insertion; returns the node index of ``obj``."""
            if isinstance(obj, _csg.CSGSurface):
                left, right = obj.children
                li = add(left)
                ri = add(right)
                nodes.append((obj.operation.value, li, ri, -1, obj.bounding_box.flat()))
                return len(nodes) - 1
            if isinstance(obj, TracerSurface):
                shape = obj.primitive
                prims.append(
                    (
                        shape.kind,
                        material_slot(obj.material if material_override is None else material_override),
                        int(obj._normal_scale),
                        0,
                        obj.get_id(),
                        shape.packed_params(),
                        obj.get_object_transform().reshape(-1),
                    )
                )
                self.surfaces.append(obj)
                if mats[prims[-1][1]][0] == matl.HOST:
                    self.host_surfaces.append((len(prims) - 1, obj))
                nodes.append((NODE_LEAF, -1, -1, len(prims) - 1, [0.0] * 6))
                return len(nodes) - 1
            raise TypeError(f"{obj!r} is neither a TracerSurface nor a CSGSurface")

        for component in self.components:
            roots.append(add(component))

        self.prims = np.array(prims, dtype=PRIM_DTYPE)
        self.nodes = np.array(nodes, dtype=NODE_DTYPE)
        self.roots = np.array(roots, dtype=np.int32)
        self.materials = np.array(mats, dtype=MATERIAL_DTYPE)
        if len(self.materials) == 0:
            self.materials = np.zeros(1, dtype=MATERIAL_DTYPE)

    # look-ups ---------------------------------------------------------------------------------
    @property
    def surface_lut(self):
        """((id, surface), ...) exactly like RayTracer._surface_lut (_pyrayt.py:257-260)."""
        return tuple((s.get_id(), s) for s in self.surfaces)

    def gooch_table(self):
        """(P,8) float64: shade_warm | shade_cool (``materials/gooch.py:36-37``) of every leaf
        surface's render material, in primitive order.  Tracer materials render with their
        ``_base_material`` (``pyrayt/materials.py:16-24``)."""
        table = np.zeros((max(1, len(self.surfaces)), 8))
        for k, surface in enumerate(self.surfaces):
            paint = getattr(surface.material, "_base_material", surface.material)
            if not hasattr(paint, "shade_pair"):
                raise AttributeError(f"{surface.material!r} cannot be rendered: it has no Gooch shading")
            table[k, :4], table[k, 4:] = paint.shade_pair()
        return table

    def prim_index(self, surface):
        for i, s in enumerate(self.surfaces):
            if s is surface:
                return i
        raise KeyError("surface is not part of this scene")

    def component_rows(self, root):
        """Number of hit rows component ``root`` returns: two per leaf surface."""

        def leaves(n):
            node = self.nodes[n]
            if node["op"] == NODE_LEAF:
                return 1
            return leaves(node["left"]) + leaves(node["right"])

        return 2 * leaves(int(self.roots[root]))

    def as_dict(self):
        """Plain-array copy of the snapshot."""
        return {
            "prims": self.prims.copy(),
            "nodes": self.nodes.copy(),
            "roots": self.roots.copy(),
            "materials": self.materials.copy(),
        }
