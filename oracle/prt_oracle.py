"""CPU oracle: a numpy restatement of PyRayT's batched ray-propagation path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker* for the HIP engine: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package ``pyrayt_amd`` (which has no CPU fallback and
fails loudly without its HIP library).

It restates, as flat functions over a scene snapshot (the plain-array form of
``include/prt.h``), what the reference computes with its object graph:

    propagate      pyrayt/_pyrayt.py:370-392        nearest positive hit over components
    interact       pyrayt/_pyrayt.py:394-452        advance, shade, drop dead rays, re-launch
    record rows    pyrayt/_pyrayt.py:168-186        the 15 result columns
    surface hits   tinygfx/g3d/world_objects.py:360-383
    CSG hits       tinygfx/g3d/csg.py:13-61, 118-160
    primitives     tinygfx/g3d/primitives.py:241-271 (sphere) 320-399 (paraboloid)
                   436-492 (plane) 516-581 (cube) 650-712 (cylinder) and the normals at
                   273-296, 401-419, 494-498, 583-602, 714-741
    quadratic      tinygfx/g3d/operations.py:28-63
    reflect/refract tinygfx/g3d/operations.py:86-162
    materials      pyrayt/materials.py:47-50, 58-62, 70-75, 112-118, 136-145

Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here against golden
vectors produced by the genuine reference (``tests/golden/generate_golden.py``), including the
reference's own known-answer tests.  The one deliberate choice: the merge of CSG hit lists is
*stable* (``kind="stable"``), which is what the numpy version the reference locks (1.20.2)
does for these short columns; numpy 2's default argsort is not.

Scene format (dict of arrays; P primitives, N nodes, C components, M materials):
    prim_type (P) int32, prim_material (P) int32, prim_normal_scale (P) int32,
    prim_surface_id (P) int64, prim_params (P,6), prim_minv (P,16) row-major
    node_op (N) int32 [0 leaf, 1 union, 2 intersect, 3 difference], node_left, node_right,
    node_prim (N) int32, node_aabb (N,6) [xmin,xmax,ymin,ymax,zmin,zmax]
    roots (C) int32, mat_kind (M) int32 [0 none 1 absorber 2 mirror 3 const 4 sellmeier
    5 a user's Glass subclass: index_at is host code  6 a user's TracableMaterial: trace() is host code],
    mat_coef (M,6); for kinds 5 / 6 also "user_materials" {material slot: the material object} and, for
    kind 6, "user_surfaces" {primitive index: the surface object handed to its trace()}
"""
import numpy as np

SPHERE, CYLINDER, PLANE, CUBE, PARABOLOID = range(5)
LEAF, UNION, INTERSECT, DIFFERENCE = range(4)
MAT_NONE, MAT_ABSORBER, MAT_MIRROR, MAT_CONST, MAT_SELLMEIER, MAT_TABLE, MAT_HOST = range(7)

INF = np.inf


class UntracableSurfaceHit(AttributeError):
    """A ray hit a surface whose material cannot trace (reference: AttributeError)."""


# ---------------------------------------------------------------------------------------------
# small numeric helpers
# ---------------------------------------------------------------------------------------------
def near_zero(x):
    """np.isclose(x, 0): |x| <= 1e-8."""
    return np.isclose(x, 0)


def coldot(a, b):
    """Column-wise dot product of two (k,n) arrays (operations.py:66-83)."""
    return np.einsum("ij,ij->j", a, b)


def quadratic_pair(a, b, c):
    """Roots of a x^2 + b x + c with the reference's degenerate handling
    (operations.py:28-63): |a|<=1e-8 -> the linear root twice; additionally |b|<=1e-8 ->
    (+inf,+inf), first entry -inf when c <= 0; negative discriminant -> (+inf,+inf)."""
    disc = b ** 2 - 4 * a * c
    lin = near_zero(a)
    s = np.sqrt(np.maximum(0, disc))
    pair = np.vstack((-b + s, -b - s)) / (2 * a + lin)
    pair = np.where(disc >= 0, pair, INF)
    pair = np.where(lin, np.tile(-c / (b + (b == 0)), (2, 1)), pair)
    only_c = np.logical_and(lin, near_zero(b))
    pair = np.where(only_c, INF, pair)
    pair[0] = np.where(np.logical_and(only_c, c <= 0), -INF, pair[0])
    return pair


def slab_pair(origin_z, dir_z, z_lo, z_hi):
    """Parameters at which a ray crosses the planes z=z_lo and z=z_hi; for a ray parallel to
    them (+inf,+inf), first entry -inf when it runs between them (primitives.py:683-703)."""
    par = near_zero(dir_z)
    between = np.logical_and(origin_z >= z_lo, origin_z <= z_hi)
    den = dir_z + par
    pair = np.vstack(((z_lo - origin_z) / den, (z_hi - origin_z) / den))
    pair = np.where(par, INF, pair)
    pair[0] = np.where(np.logical_and(par, between), -INF, pair[0])
    return pair


def overlap_of_sorted_pairs(p, q):
    """[max(lo), min(hi)] of two (2,n) pairs once each is sorted; all-inf unless lo <= hi
    (the hstack/sort/reshape trick of primitives.py:705-711)."""
    p = np.sort(p, axis=0)
    q = np.sort(q, axis=0)
    both = np.vstack((np.maximum(p[0], q[0]), np.minimum(p[1], q[1])))
    return np.where(both[0] <= both[1], both, INF)


# ---------------------------------------------------------------------------------------------
# primitives in object space: o, d are (3,n)
# ---------------------------------------------------------------------------------------------
def hit_sphere(params, o, d):
    """primitives.py:241-271 -- no guard on a == 0 (zero directions give NaN)."""
    r = params[0]
    a = coldot(d, d)
    b = 2 * coldot(d, o)
    c = coldot(o, o) - r ** 2
    disc = b ** 2 - 4 * a * c
    s = np.sqrt(np.maximum(0, disc))
    with np.errstate(invalid="ignore", divide="ignore"):
        pair = np.array((-b + s, -b - s)) / (2 * a)
    return np.where(disc >= 0, pair, INF)


def hit_cylinder(params, o, d):
    """primitives.py:650-712."""
    r, h_lo, h_hi = params[0], params[1], params[2]
    a = coldot(d[:2], d[:2])
    b = 2 * coldot(d[:2], o[:2])
    c = coldot(o[:2], o[:2]) - r ** 2
    side = np.sort(quadratic_pair(a, b, c), axis=0)
    return overlap_of_sorted_pairs(side, slab_pair(o[2], d[2], h_lo, h_hi))


def hit_paraboloid(params, o, d):
    """primitives.py:320-399: x^2 + y^2 = 4 f z clipped to 0 <= z <= height."""
    f, height = params[0], params[1]
    a = coldot(d[:2], d[:2])
    b = 2 * coldot(o[:2], d[:2]) - 4 * f * d[2]
    c = coldot(o[:2], o[:2]) - 4 * f * o[2]
    disc = b ** 2 - 4 * a * c
    lin = near_zero(a)
    s = np.sqrt(np.maximum(0, disc))
    pair = np.vstack((-b + s, -b - s)) / (2 * a + lin)
    pair = np.where(disc >= 0, pair, INF)
    single = np.empty_like(pair)
    single[0] = -c / (b + near_zero(b))
    single[1] = np.where(d[2] >= 0, INF, -INF)
    pair = np.where(lin, single, pair)
    return overlap_of_sorted_pairs(pair, slab_pair(o[2], d[2], 0, height))


def _axis_slab(o_ax, d_ax, lo, hi, inside):
    """One axis of the cube / plane-patch slab test: parameters of the two bounding planes,
    with the parallel case mapped to (-inf if inside else +inf, +inf)."""
    z = near_zero(d_ax)
    first = np.where(z, np.where(inside, -INF, INF), -(o_ax - lo) / (d_ax + z))
    second = np.where(z, INF, -(o_ax - hi) / (d_ax + z))
    return np.minimum(first, second), np.maximum(first, second)


def hit_plane(params, o, d):
    """primitives.py:436-492: z=0 patch |x|<=W/2, |y|<=L/2; returns the hit twice."""
    los, his = [], []
    for axis in (0, 1):
        half = params[axis] / 2
        # the reference tests the parallel case with |o| <= half and computes the two crossings
        # as -(o - half)/d and -(o + half)/d
        lo, hi = _axis_slab(o[axis], d[axis], half, -half, np.abs(o[axis]) <= half)
        los.append(lo)
        his.append(hi)
    enter = np.maximum(los[0], los[1])
    leave = np.minimum(his[0], his[1])
    skew = near_zero(d[2])
    t = np.where(skew, INF, -o[2] / (d[2] + skew))
    t = np.where(np.logical_and(t >= enter, t <= leave), t, INF)
    return np.tile(t, (2, 1))


def hit_cube(params, o, d):
    """primitives.py:516-581: strict lo < hi."""
    los, his = [], []
    for axis in range(3):
        lo_v, hi_v = params[2 * axis], params[2 * axis + 1]
        inside = np.logical_and(o[axis] <= hi_v, o[axis] >= lo_v)
        lo, hi = _axis_slab(o[axis], d[axis], lo_v, hi_v, inside)
        los.append(lo)
        his.append(hi)
    enter = np.max(np.vstack(los), axis=0)
    leave = np.min(np.vstack(his), axis=0)
    pair = np.vstack((enter, leave))
    return np.where(enter < leave, pair, INF)


_HIT = {SPHERE: hit_sphere, CYLINDER: hit_cylinder, PLANE: hit_plane, CUBE: hit_cube,
        PARABOLOID: hit_paraboloid}


def object_normal(kind, params, p):
    """Object-space normal (4,n) at object-space points p (4,n)."""
    n = np.zeros_like(p)
    if kind == SPHERE:  # primitives.py:273-296
        n[:3] = p[:3]
    elif kind == CYLINDER:  # :714-741
        n[:2] = p[:2]
        down = np.isclose(p[2], params[1])
        up = np.isclose(p[2], params[2])
        n = np.where(down, np.array([[0.0], [0.0], [-1.0], [0.0]]), n)
        n = np.where(up, np.array([[0.0], [0.0], [1.0], [0.0]]), n)
    elif kind == PLANE:  # :494-498 (already unit, never normalised)
        n[2] = 1.0
        return n
    elif kind == CUBE:  # :583-602
        spans = np.asarray(params[:6], dtype=float).reshape(3, 2)
        n[:3] = np.where(np.isclose(p[:3], spans[:, 0:1]), -1.0, 0.0)
        n[:3] = np.where(np.isclose(p[:3], spans[:, 1:2]), 1.0, n[:3])
    elif kind == PARABOLOID:  # :401-419
        n[:2] = p[:2]
        n[2] = -2 * params[0]
        n = np.where(np.isclose(p[2], params[1]), np.array([[0.0], [0.0], [1.0], [0.0]]), n)
    else:
        raise ValueError(kind)
    with np.errstate(invalid="ignore", divide="ignore"):
        n /= np.linalg.norm(n, axis=0)
    return n


# ---------------------------------------------------------------------------------------------
# surfaces and CSG in world space: rays is (2,4,n)
# ---------------------------------------------------------------------------------------------
def surface_hits(scene, p, rays):
    """TracerSurface.intersect (world_objects.py:360-383): world->object by the 4x4 inverse
    (directions are not renormalised so t stays a world-space parameter), primitive test,
    ascending sort."""
    minv = scene["prim_minv"][p].reshape(4, 4)
    local = np.matmul(minv, rays)
    pair = _HIT[int(scene["prim_type"][p])](scene["prim_params"][p], local[0, :3], local[1, :3])
    return np.sort(pair, axis=0)


def world_normals(scene, p, points):
    """TracerSurface.get_world_normals (world_objects.py:401-418)."""
    minv = scene["prim_minv"][p].reshape(4, 4)
    local = np.matmul(minv, points)
    n_obj = object_normal(int(scene["prim_type"][p]), scene["prim_params"][p], local)
    n_w = np.matmul(minv.T, n_obj)
    n_w[3] = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        n_w /= np.linalg.norm(n_w, axis=0)
    return n_w * int(scene["prim_normal_scale"][p])


def merge_lists(left, right, op):
    """array_csg without the final sort (csg.py:13-61): returns (values with rejected entries
    set to +inf, permutation that sorted the stacked inputs)."""
    stacked = np.vstack((left, right))
    cols = np.arange(stacked.shape[1])
    order = np.argsort(stacked, axis=0, kind="stable")
    merged = stacked[order, cols]
    step = np.where(order & 1, -1, 1)
    if op == DIFFERENCE:
        step = np.where(np.logical_xor(order & 1, order >= left.shape[0]), -1, 1)
    depth = np.cumsum(step, axis=0) + (1 if op == DIFFERENCE else 0)
    if op == UNION:
        keep = np.logical_xor(depth != 0, np.roll(depth, 1, axis=0) != 0)
    elif op in (INTERSECT, DIFFERENCE):
        two = depth == 2
        keep = np.logical_or(two, np.roll(two, 1, axis=0))
    else:
        raise ValueError(f"operation {op} is invalid")
    return np.where(keep, merged, INF), order


def box_touches(aabb, rays):
    """CSG cull predicate (csg.py:126-128): a ray 'touches' the node's world-space box iff the
    cube test returns any finite parameter."""
    with np.errstate(invalid="ignore", divide="ignore"):
        pair = hit_cube(aabb, rays[0, :3], rays[1, :3])
    return np.any(np.isfinite(pair), axis=0)


def node_hits(scene, node, rays):
    """(hits (m,n), ids (m,n)) of a component subtree; ids are surface ids where the hit is
    not +inf and -1 there (the reference leaves arbitrary carried ids at +inf entries; they
    are never consumed, _pyrayt.py:380-386).  A -inf entry -- a slab or linear branch that holds
    the whole line -- keeps its id: the tracer discards it with every hit <= 0, but the renderers'
    rule can select it (renderers.py:79-83), and the reference then reports that surface."""
    op = int(scene["node_op"][node])
    n = rays.shape[-1]
    if op == LEAF:
        p = int(scene["node_prim"][node])
        with np.errstate(invalid="ignore", divide="ignore"):
            hits = surface_hits(scene, p, rays)
        ids = np.where(hits < INF, scene["prim_surface_id"][p], -1).astype(np.int64)
        return hits, ids
    touched = box_touches(scene["node_aabb"][node], rays)
    sub = rays[:, :, touched]
    l_hits, l_ids = node_hits(scene, int(scene["node_left"][node]), sub)
    r_hits, r_ids = node_hits(scene, int(scene["node_right"][node]), sub)
    cols = np.arange(sub.shape[-1])
    values, order = merge_lists(l_hits, r_hits, op)
    ids = np.vstack((l_ids, r_ids))[order, cols]
    final = np.argsort(values, axis=0, kind="stable")
    values = values[final, cols]
    ids = np.where(values < INF, ids[final, cols], -1)
    m = values.shape[0]
    hits_all = np.full((m, n), INF)
    ids_all = np.full((m, n), -1, dtype=np.int64)
    hits_all[:, touched] = values
    ids_all[:, touched] = ids
    return hits_all, ids_all


def component_hits(scene, root, rays):
    return node_hits(scene, int(scene["roots"][root]), np.atleast_3d(rays))


def propagate(scene, rays13):
    """RayTracer._st_propagate (_pyrayt.py:370-392)."""
    rays = np.ascontiguousarray(rays13[:8]).reshape(2, 4, -1)
    n = rays.shape[-1]
    best_t = np.full(n, INF)
    best_s = np.full(n, -1, dtype=np.int64)
    cols = np.arange(n)
    for root in range(len(scene["roots"])):
        hits, ids = component_hits(scene, root, rays)
        hits = np.where(hits > 0, hits, INF)
        row = np.argmin(hits, axis=0)
        t, s = hits[row, cols], ids[row, cols]
        better = t < best_t
        best_t = np.where(better, t, best_t)
        best_s = np.where(better, s, best_s)
    return best_t, best_s


# ---------------------------------------------------------------------------------------------
# shading
# ---------------------------------------------------------------------------------------------
def reflect(v, n):
    """operations.py:104-107."""
    return v - 2 * n * coldot(v, n)


def refract(v, n, n1, n2, n_world=1):
    """operations.py:110-162 (v is normalised first; returns new directions and indices)."""
    v = v / np.linalg.norm(v, axis=0)
    cos_p = coldot(v, n)
    cos_n = coldot(v, -n)
    leaving = cos_p > 0
    n2 = np.where(leaving, n_world, n2)
    n = np.where(leaving, -n, n)
    r = n1 / n2
    cos1 = np.where(leaving, cos_p, cos_n)
    radicand = 1 - (r ** 2) * (1 - cos1 ** 2)
    cos2 = np.sqrt(np.maximum(0, radicand))
    out = np.where(radicand > 0, r * v + (r * cos1 - cos2) * n, v + 2 * cos1 * n)
    out /= np.linalg.norm(out, axis=0)
    return out, np.where(radicand > 0, n2, n1)


def material_index(kind, coef, wavelength):
    """materials.py:112-118 (constant) and :136-145 (Sellmeier, wavelength in um)."""
    if kind == MAT_CONST:
        return np.full(np.shape(wavelength), coef[0])
    w2 = wavelength ** 2
    return np.sqrt(1 + (coef[0] * w2) / (w2 - coef[3]) + (coef[1] * w2) / (w2 - coef[4])
                   + (coef[2] * w2) / (w2 - coef[5]))


def material_trace(scene, p, sub):
    """surface.material.trace(surface, ray_subset) (materials.py:47-50, 58-62, 70-75);
    ``sub`` is a (13,k) copy whose origins already sit on the surface; returns it updated."""
    m = int(scene["prim_material"][p])
    kind = int(scene["mat_kind"][m])
    if kind == MAT_NONE:
        raise UntracableSurfaceHit(f"surface {int(scene['prim_surface_id'][p])} has no tracable material")
    if kind == MAT_ABSORBER:
        sub[4:8] = 0
        return sub
    normals = world_normals(scene, p, sub[0:4])
    if kind == MAT_MIRROR:
        sub[4:8] = reflect(sub[4:8], normals)
        return sub
    if kind == MAT_HOST:
        # a user's trace() (materials.py:26-37): called as _pyrayt.py:408 calls it, on a RaySet view of the copy
        material = scene["user_materials"][m]
        surface = scene.get("user_surfaces", {}).get(p)
        ray_set_type = scene.get("ray_set_type")
        handed = sub.view(ray_set_type) if ray_set_type is not None else sub
        sub[...] = np.asarray(material.trace(surface, handed), dtype=float)
        return sub
    if kind == MAT_TABLE:
        # a user's Glass subclass: Glass.trace (materials.py:70-75) with ITS index_at on the wavelength row
        n_glass = np.broadcast_to(np.asarray(scene["user_materials"][m].index_at(np.array(sub[10])), dtype=float), sub[10].shape)
    else:
        n_glass = material_index(kind, scene["mat_coef"][m], sub[10])
    with np.errstate(invalid="ignore", divide="ignore"):
        sub[4:8], sub[11] = refract(sub[4:8], normals, sub[11], n_glass)
    return sub


def interact(scene, rays13, t, surf, generation, generation_limit, ray_offset=1e-6):
    """RayTracer._st_interact (_pyrayt.py:394-452) + _RayTraceDataframe.insert (:168-186).
    Returns (rows (k,15) or None when every ray is dead, next ray set (13,k) or None)."""
    cur = np.asarray(rays13, dtype=float)
    nxt = cur.copy()
    nxt[4:8, surf == -1] = 0
    for p in range(len(scene["prim_type"])):
        mask = surf == scene["prim_surface_id"][p]
        if np.any(mask):
            nxt[0:4, mask] += nxt[4:8, mask] * t[mask]
            nxt[:, mask] = material_trace(scene, p, nxt[:, mask])
    absorbed = np.isclose(np.linalg.norm(cur[4:8], axis=0), 0)
    dead = np.logical_or(absorbed, surf == -1)  # the intensity test is a no-op upstream (Q2)
    if np.all(dead):
        return None, None
    live = np.logical_not(dead)
    nxt = nxt[:, live]
    before = cur[:, live]
    tilt = before[4:7] / np.linalg.norm(before[4:7], axis=0)
    rows = np.vstack((before[8:13], surf[live], before[0:3], nxt[0:3], tilt)).T
    nxt[8] = generation + 1
    if generation + 1 != generation_limit:
        nxt[0:4] += ray_offset * nxt[4:8]
    return rows, nxt


def trace(scene, rays13, generation_limit=10, ray_offset=1e-6, log=None):
    """RayTracer.trace() from an initial ray set (_pyrayt.py:329-339): (rows (R,15),
    rows per generation).  ``log``, if a dict, receives per-generation t/surf/next arrays."""
    rays = np.array(rays13, dtype=float)
    blocks, counts = [], []
    generation = 0
    while True:
        t, surf = propagate(scene, rays)
        if log is not None:
            log[f"t_{generation}"], log[f"surf_{generation}"] = t, surf
        rows, nxt = interact(scene, rays, t, surf, generation, generation_limit, ray_offset)
        if rows is None:
            break
        blocks.append(rows)
        counts.append(rows.shape[0])
        if log is not None:
            log[f"next_{generation}"] = nxt.copy()
        rays = nxt
        generation += 1
        if generation == generation_limit:
            break
    frame = np.vstack(blocks) if blocks else np.zeros((0, 15))
    return frame, counts
