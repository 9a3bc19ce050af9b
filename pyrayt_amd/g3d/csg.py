"""Constructive solid geometry nodes (host bookkeeping only).

Counterpart of the reference's ``tinygfx/g3d/csg.py``: a binary node joining two
``Intersectable`` children with UNION / INTERSECT / DIFFERENCE.  The per-ray interval algebra
(``csg.py:13-61`` ``array_csg`` and ``:118-160`` ``CSGSurface.intersect``) runs in the HIP
kernels; what stays here is what the kernels take as input: the tree shape, the normal
inversion of a DIFFERENCE's right child (``csg.py:87-89``) and the world-space bounding box
used as a per-ray cull predicate (``csg.py:93-116,126-128``).
"""
from enum import Enum

import numpy as np

from .objects import Intersectable
from .shapes import AxisBox


class Operation(Enum):
    UNION = 1
    INTERSECT = 2
    DIFFERENCE = 3


def _span_algebra(left, right, operation):
    """Bounding span of ``left (op) right`` on one axis, where each argument is a (min, max)
    pair.  The reference gets this by running its hit-list algebra on the two spans and keeping
    the first two survivors (``csg.py:98-109``), so the same rule is applied here, entry by
    entry: stable merge, +1 on even source positions / -1 on odd ones, running sum, keep the
    entries where the sum enters/leaves zero (UNION) or where it is or was 2 (INTERSECT), with
    the wrap-around of ``np.roll`` (``csg.py:41-58``).  Disjoint UNION operands therefore yield
    only the first operand's span, as upstream."""
    values = [left[0], left[1], right[0], right[1]]
    order = sorted(range(4), key=lambda i: (values[i], i))  # stable
    running, counts = 0, []
    for i in order:
        running += -1 if (i & 1) else 1
        counts.append(running)
    survivors = []
    for j, i in enumerate(order):
        before = counts[j - 1]  # j == 0 wraps to the last entry, like np.roll
        if operation is Operation.UNION:
            keep = (counts[j] != 0) != (before != 0)
        else:
            keep = counts[j] == 2 or before == 2
        survivors.append(values[i] if keep else np.inf)
    survivors.sort()
    return survivors[0], survivors[1]


class CSGSurface(Intersectable):
    def __init__(self, l_child, r_child, operation, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(operation, Operation):
            raise ValueError(f"operation {operation} is invalid")
        self._operation = operation
        # the cull box is cached and refreshed through the watch lists, in upstream's order
        # (csg.py:76-91): watch first, attach the children, flip normals, compute
        self.var_watchlist.append(self._update_bounding_box)
        self._l_child = l_child
        self._l_child.attach_to(self)
        self._r_child = r_child
        self._r_child.attach_to(self)
        if operation is Operation.DIFFERENCE:
            # the subtracted solid shows its inside: flip its normals (csg.py:87-89)
            r_child.invert_normals()
        self._update_bounding_box()

    @property
    def operation(self):
        return self._operation

    @property
    def children(self):
        return self._l_child, self._r_child

    def _update_bounding_box(self):
        """World-space cull box from the children's *current* boxes (``csg.py:93-116``): the left
        child's for a DIFFERENCE, the span algebra of both otherwise.  It runs whenever this node or
        one of the parts attached to it changes its transform -- and only then, which reproduces
        upstream's stale box of a right-nested tree moved after construction
        (``world_objects.py:315-317``; tests/golden/scene_stale_box.npz)."""
        if self._operation is Operation.DIFFERENCE:
            self._aobb = self._l_child.bounding_box
            return
        l_spans = self._l_child.bounding_box.axis_spans
        r_spans = self._r_child.bounding_box.axis_spans
        lo, hi = zip(*(_span_algebra(l_spans[a], r_spans[a], self._operation) for a in range(3)))
        self._aobb = AxisBox(lo, hi)

    @property
    def bounding_box(self):
        return self._aobb

    def invert_normals(self):
        self._l_child.invert_normals()
        self._r_child.invert_normals()

    def reset_normals(self):
        self._l_child.reset_normals()
        self._r_child.reset_normals()

    @property
    def surface_ids(self):
        return self._l_child.surface_ids + self._r_child.surface_ids

    def _append_world_transform(self, matrix):
        # moving the node moves the whole subtree (csg.py:175-179)
        super()._append_world_transform(matrix)
        self._l_child.transform(matrix)
        self._r_child.transform(matrix)


def array_csg(array1, array2, operation, sort_output=True):
    """The CSG interval algebra on two arrays of sorted hits (one column per ray, or 1-D for a single
    ray): entries of the merged list that are not crossings of the combined solid's boundary become
    +inf (``tinygfx/g3d/csg.py:13-61``).  Runs on the HIP engine (``prt_array_csg``)."""
    from .. import engine

    if not isinstance(operation, Operation):
        raise ValueError(f"operation {operation} is invalid")
    left = np.asarray(array1, dtype=float)
    right = np.asarray(array2, dtype=float)
    single = left.ndim == 1
    out = engine.ops_array_csg(left.reshape(left.shape[0], -1), right.reshape(right.shape[0], -1),
                               operation.value, sort_output)
    return out[:, 0] if single else out


def union(s0, s1):
    return CSGSurface(s0, s1, Operation.UNION)


def intersect(s0, s1):
    return CSGSurface(s0, s1, Operation.INTERSECT)


def difference(s0, s1):
    return CSGSurface(s0, s1, Operation.DIFFERENCE)
