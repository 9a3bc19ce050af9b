#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the *genuine* reference.

Run only in the build container (the reference lives at /root/reference, read-only, and never
travels to the GPU box):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

What is written is data only -- inputs and the outputs the reference produced for them:
plain float64 / int64 arrays in .npz files.  No reference source, bytecode or pickled
reference object goes into the repository.

Version skew recorded with every fixture: this container has numpy 2.x / pandas 2.x whereas
the reference locks numpy 1.20.2 / pandas 1.2.4.  Two consequences handled here:
  * ``DataFrame.append`` is gone in pandas 2 -> an in-process shim maps it onto ``concat``;
  * numpy 2's default argsort is not stable on AVX-512 even for tiny columns, numpy 1.20's
    was (insertion sort for n <= 16).  Every end-to-end fixture is therefore produced twice,
    with the default argsort and with argsort forced stable inside ``tinygfx.g3d.csg``, and is
    only accepted if both runs agree bit for bit.
"""
import itertools
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))  # tests/ for scenes.py

import numpy as np
import pandas as pd

if not hasattr(pd.DataFrame, "append"):
    pd.DataFrame.append = lambda self, other, ignore_index=False: pd.concat(
        [self, other], ignore_index=ignore_index
    )

import pyrayt  # noqa: E402  (the reference)
import tinygfx.g3d as cg  # noqa: E402
import tinygfx.g3d.csg as ref_csg  # noqa: E402
import tinygfx.g3d.operations as ref_ops  # noqa: E402

import scenes  # noqa: E402

API = scenes.reference_api()


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
class _StableNumpy:
    """numpy proxy whose argsort is always stable; swapped into tinygfx.g3d.csg."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, axis=-1, **kw):
        return np.argsort(a, axis=axis, kind="stable")


def reset_ids():
    cg.world_objects.CountedObject._ids = itertools.count(0)


PRIM_KIND = {"Sphere": 0, "Cylinder": 1, "Plane": 2, "Cube": 3, "Paraboloid": 4}


def material_record(material):
    import pyrayt.materials as m

    if isinstance(material, m._AbsorbingMaterial):
        return 1, [0.0] * 6
    if isinstance(material, m._ReflectingMaterial):
        return 2, [0.0] * 6
    if isinstance(material, m.BasicRefractor):
        return 3, [float(material._refractive_index)] + [0.0] * 5
    if isinstance(material, m.SellmeierRefractor):
        return 4, [material.b1, material.b2, material.b3, material.c1, material.c2, material.c3]
    return 0, [0.0] * 6


def prim_params(prim):
    name = type(prim).__name__
    if name == "Sphere":
        p = [prim._radius]
    elif name == "Cylinder":
        p = [prim._radius, prim._h_min, prim._h_max]
    elif name == "Plane":
        p = [prim._width, prim._length]
    elif name == "Cube":
        p = list(prim.axis_spans.reshape(-1))
    elif name == "Paraboloid":
        p = [prim._focus, prim._height]
    else:
        raise TypeError(name)
    return PRIM_KIND[name], [float(v) for v in p] + [0.0] * (6 - len(p))


def snapshot(components):
    """Walk reference objects into the flat layout of include/prt.h (plain arrays)."""
    prims, nodes, roots, mats, mat_ids = [], [], [], [], {}

    def mat_slot(material):
        if id(material) not in mat_ids:
            mat_ids[id(material)] = len(mats)
            mats.append(material_record(material))
        return mat_ids[id(material)]

    def add(obj):
        if isinstance(obj, ref_csg.CSGSurface):
            li, ri = add(obj._l_child), add(obj._r_child)
            nodes.append((obj._operation.value, li, ri, -1, list(obj._aobb.axis_spans.reshape(-1))))
        else:
            kind, params = prim_params(obj._surface_primitive)
            prims.append(
                (kind, mat_slot(obj.material), int(obj._normal_scale), obj.get_id(), params,
                 list(obj._get_object_transform().reshape(-1)))
            )
            nodes.append((0, -1, -1, len(prims) - 1, [0.0] * 6))
        return len(nodes) - 1

    for comp in components:
        roots.append(add(comp))
    return {
        "prim_type": np.array([p[0] for p in prims], dtype=np.int32),
        "prim_material": np.array([p[1] for p in prims], dtype=np.int32),
        "prim_normal_scale": np.array([p[2] for p in prims], dtype=np.int32),
        "prim_surface_id": np.array([p[3] for p in prims], dtype=np.int64),
        "prim_params": np.array([p[4] for p in prims], dtype=float).reshape(-1, 6),
        "prim_minv": np.array([p[5] for p in prims], dtype=float).reshape(-1, 16),
        "node_op": np.array([n[0] for n in nodes], dtype=np.int32),
        "node_left": np.array([n[1] for n in nodes], dtype=np.int32),
        "node_right": np.array([n[2] for n in nodes], dtype=np.int32),
        "node_prim": np.array([n[3] for n in nodes], dtype=np.int32),
        "node_aabb": np.array([n[4] for n in nodes], dtype=float).reshape(-1, 6),
        "roots": np.array(roots, dtype=np.int32),
        "mat_kind": np.array([m[0] for m in mats], dtype=np.int32),
        "mat_coef": np.array([m[1] for m in mats], dtype=float).reshape(-1, 6),
    }


class _PresetSource:
    """Duck-typed source that emits a prepared ray block."""

    def __init__(self, rays):
        self._rays = rays

    def generate_rays(self, n):
        return self._rays.copy().view(pyrayt.RaySet)


class RecordingTracer(pyrayt.RayTracer):
    """Reference tracer that also keeps the per-generation intermediates."""

    def reset(self):
        super().reset()
        self.log_t, self.log_surf, self.log_next = [], [], []

    def _st_propagate(self):
        super()._st_propagate()
        self.log_t.append(self._hit_distances.copy())
        self.log_surf.append(self._hit_surfaces.copy())

    def _st_interact(self):
        super()._st_interact()
        self.log_next.append(np.array(self._ray_set).copy())


def run_reference(components, rays, generation_limit):
    tracer = RecordingTracer(_PresetSource(rays), components, rays_per_source=rays.shape[1],
                             generation_limit=generation_limit)
    frame = tracer.trace()
    out = {"frame": frame.to_numpy(dtype=float), "n_generations": np.int64(len(tracer.log_t))}
    for g, (t, s) in enumerate(zip(tracer.log_t, tracer.log_surf)):
        out[f"t_{g}"] = t
        out[f"surf_{g}"] = s.astype(np.int64)
    for g, nxt in enumerate(tracer.log_next):
        out[f"next_{g}"] = nxt
    return out


def scene_fixture(name, generation_limit, *args, keep_next=True, allow_sensitive=False, **kwargs):
    """Build + trace twice (default / stable argsort), require equality, save."""
    results = []
    for stable in (False, True):
        ref_csg.np = _StableNumpy() if stable else np
        try:
            reset_ids()
            components, rays = scenes.SCENES[name](API, *args, **kwargs)
            snap = snapshot(components)
            lut = []
            for comp in components:
                lut += [sid for sid, _ in comp.surface_ids]
            res = run_reference(components, rays, generation_limit)
        finally:
            ref_csg.np = np
        results.append(res)
    b, a = results  # a = stable run (golden), b = default argsort
    same = a.keys() == b.keys() and all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
    if not same:
        assert allow_sensitive, (name, "default and stable argsort runs differ")
        print(f"  note: {name} depends on argsort stability; stable result kept")
    a["argsort_sensitive"] = np.bool_(not same)
    if not keep_next:
        a = {k: v for k, v in a.items() if not k.startswith("next_")}
    payload = {**snap, **a, "rays0": rays, "lut_ids": np.array(lut, dtype=np.int64),
               "generation_limit": np.int64(generation_limit)}
    path = os.path.join(HERE, f"scene_{name}.npz")
    np.savez_compressed(path, **payload)
    rows = a["frame"].shape[0]
    print(f"{name:20s} rays={rays.shape[1]:7d} rows={rows:8d} gens={int(a['n_generations'])} "
          f"-> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")
    return payload


# ---------------------------------------------------------------------------------------------
# per-function vectors
# ---------------------------------------------------------------------------------------------
def surface_under_test(kind, variant):
    """One surface of each primitive kind under identity / translated / rotated / anisotropic
    scale transforms."""
    make = {
        "sphere": lambda: cg.Sphere(1.3),
        "cylinder": lambda: cg.Cylinder(0.9, -0.7, 1.1),
        "plane": lambda: cg.XYPlane(2.5, 1.5),
        "cube": lambda: cg.Cuboid((-1.0, -0.5, -0.8), (0.7, 1.2, 0.9)),
        "paraboloid": lambda: cg.Paraboloid(0.8, 1.7),
    }[kind]
    s = make()
    if variant == "moved":
        s.move(0.4, -0.3, 0.6)
    elif variant == "rotated":
        s.rotate_x(33).rotate_y(-71).rotate_z(12).move(0.1, 0.2, -0.3)
    elif variant == "scaled":
        s.scale(1.5, 0.6, 2.0).rotate_y(40).move(-0.2, 0.1, 0.3)
    return s


def primitive_vectors():
    out = {}
    n = 1024
    for ki, kind in enumerate(("sphere", "cylinder", "plane", "cube", "paraboloid")):
        for vi, variant in enumerate(("identity", "moved", "rotated", "scaled")):
            reset_ids()
            surf = surface_under_test(kind, variant)
            rays = scenes.random_rays(n, seed=100 + 10 * ki + vi, box=3.0)
            if variant == "identity":
                # origins exactly on faces / caps, directions exactly axis parallel
                rays[0:3, 40:52] = np.array(
                    [[0.7, 0, 0], [-1.0, 0, 0], [0, 1.2, 0], [0, -0.5, 0], [0, 0, 0.9], [0, 0, -0.8],
                     [0, 0, 1.1], [0, 0, -0.7], [0, 0, 1.7], [0, 0, 0], [0.9, 0, 0], [0, 1.3, 0]]
                ).T
            block = rays[:8].reshape(2, 4, n)
            hits, ids = surf.intersect(block)
            # normals at the first positive finite hit (origin itself where there is none)
            masked = np.where(hits > 0, hits, np.inf)
            first = np.min(masked, axis=0)
            has = np.isfinite(first)
            pts = block[0] + block[1] * np.where(has, first, 0.0)
            with np.errstate(all="ignore"):
                normals = surf.get_world_normals(pts)
            key = f"{kind}_{variant}"
            snap = snapshot([surf])
            for k, v in snap.items():
                out[f"{key}__{k}"] = v
            out[f"{key}__rays"] = rays
            out[f"{key}__hits"] = hits
            out[f"{key}__points"] = pts
            out[f"{key}__normals"] = normals
            out[f"{key}__has_hit"] = has
    path = os.path.join(HERE, "primitives.npz")
    np.savez_compressed(path, **out)
    print(f"primitives           -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def csg_vectors():
    """Component-level intersect(): pairs and chains of primitives under the three operations,
    default and forced-stable argsort required equal."""
    out = {}
    n = 1024
    recipes = {
        "union_spheres": lambda: ref_csg.union(cg.Sphere(1), cg.Sphere(1).move_y(-1)),
        "intersect_spheres": lambda: ref_csg.intersect(cg.Sphere(1), cg.Sphere(1).move_y(-1)),
        "difference_spheres": lambda: ref_csg.difference(cg.Sphere(1), cg.Sphere(1).move_y(-1)),
        "plane_minus_cylinder": lambda: ref_csg.difference(cg.XYPlane(3, 3), cg.Cylinder(0.5, -0.05, 0.05)),
        "cube_chain": lambda: ref_csg.difference(
            ref_csg.difference(cg.Cuboid.from_length(2.0), cg.Cuboid.from_length(1.5).rotate_y(30).move(-1, 0, 1)),
            cg.Cuboid.from_length(1.5).rotate_y(-30).move(1, 0, 1)),
        "right_nested": lambda: ref_csg.union(
            cg.Sphere(0.9).move_x(-0.8),
            ref_csg.intersect(cg.Cylinder(0.7, -1, 1).rotate_x(90), cg.Sphere(1.0).move_x(0.4))),
        "balanced": lambda: ref_csg.difference(
            ref_csg.union(cg.Sphere(1.0), cg.Cuboid.from_length(1.4).move_x(0.9)),
            ref_csg.intersect(cg.Cylinder(0.5, -2, 2), cg.Paraboloid(0.5, 1.5).move_z(-0.5))),
    }
    for ri, (name, make) in enumerate(recipes.items()):
        results = []
        for stable in (False, True):
            ref_csg.np = _StableNumpy() if stable else np
            try:
                reset_ids()
                comp = make()
                if name in ("cube_chain", "balanced"):
                    comp.rotate_z(25).move(0.2, -0.1, 0.3)
                rays = scenes.random_rays(n, seed=200 + ri, box=3.0)
                hits, ids = comp.intersect(rays[:8].reshape(2, 4, n))
                snap = snapshot([comp])
            finally:
                ref_csg.np = np
            results.append((hits, ids))
        # The stable run is the golden one: it is what the locked numpy 1.20 does (introsort
        # falls back to insertion sort below 16 elements).  A recipe whose default-argsort run
        # differs (a Plane's double hit (t,t) is a guaranteed tie) is flagged, not rejected.
        hits, ids = results[1]
        finite = np.isfinite(hits)
        same = np.array_equal(results[0][0], hits, equal_nan=True) and np.array_equal(
            results[0][1][np.isfinite(results[0][0])], ids[finite])
        out[f"{name}__argsort_sensitive"] = np.bool_(not same)
        if not same:
            print(f"  note: {name} depends on argsort stability; stable result kept")
        for k, v in snap.items():
            out[f"{name}__{k}"] = v
        out[f"{name}__rays"] = rays
        out[f"{name}__hits"] = hits
        out[f"{name}__ids"] = np.where(finite, ids, -1).astype(np.int64)
    # the reference's own array_csg known answers (test_csg.py:212-231) plus random lists
    a1 = np.array((1, 4, 5, 10), dtype=float)
    a2 = np.array((0, 2, 3, 5, 6, 7, 8, 9, 11, 12), dtype=float)
    out["array_csg__a1"], out["array_csg__a2"] = a1, a2
    for op in ref_csg.Operation:
        out[f"array_csg__{op.name}"] = ref_csg.array_csg(a1, a2, op)
    rng = np.random.default_rng(5)
    for ml, mr in ((2, 2), (4, 2), (2, 4), (4, 4), (6, 2)):
        left = np.sort(rng.uniform(-3, 3, (ml, 512)), axis=0)
        right = np.sort(rng.uniform(-3, 3, (mr, 512)), axis=0)
        left[:, :32] = np.inf  # missed children
        right[:, 16:48] = np.inf
        out[f"array_csg_rand_{ml}_{mr}__left"] = left
        out[f"array_csg_rand_{ml}_{mr}__right"] = right
        for op in ref_csg.Operation:
            ref_csg.np = _StableNumpy()
            try:
                out[f"array_csg_rand_{ml}_{mr}__{op.name}"] = ref_csg.array_csg(left, right, op)
            finally:
                ref_csg.np = np
    path = os.path.join(HERE, "csg.npz")
    np.savez_compressed(path, **out)
    print(f"csg                  -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def shading_vectors():
    """refract / reflect / index_at and Material.trace on seeded inputs incl. TIR and exits."""
    import pyrayt.materials as m

    out = {}
    rng = np.random.default_rng(9)
    k = 2048
    v = np.zeros((4, k))
    v[:3] = rng.normal(size=(3, k))
    v[:3] *= rng.uniform(0.5, 2.0, k) / np.linalg.norm(v[:3], axis=0)  # not unit length on purpose
    nrm = np.zeros((4, k))
    nrm[:3] = rng.normal(size=(3, k))
    nrm[:3] /= np.linalg.norm(nrm[:3], axis=0)
    n1 = rng.choice([1.0, 1.5, 1.6, 1.3], k)
    n2 = rng.choice([1.5, 1.6, 1.7, 1.0], k)
    out["vectors"], out["normals"], out["n1"], out["n2"] = v, nrm, n1, n2
    refr, n_out = ref_ops.refract(v.copy(), nrm.copy(), n1.copy(), n2.copy())
    out["refracted"], out["n_refracted"] = refr, n_out
    out["reflected"] = ref_ops.reflect(v.copy(), nrm.copy())
    wl = np.linspace(0.35, 1.1, 64)
    out["wavelengths"] = wl
    for name, glass in m.glass.items():
        out[f"index_{name}"] = np.asarray(glass.index_at(wl), dtype=float)

    # Material.trace on every primitive kind (surface given by its snapshot)
    n = 512
    for ki, kind in enumerate(("sphere", "cylinder", "plane", "cube", "paraboloid")):
        for mname, material in (("absorber", m.absorber), ("mirror", m.mirror),
                                ("ideal", m.glass["ideal"]), ("SF5", m.glass["SF5"])):
            reset_ids()
            surf = surface_under_test(kind, "scaled")
            surf.material = material
            rays = scenes.random_rays(n, seed=300 + ki, box=3.0, wavelength=0.5)
            rays[10] = np.linspace(0.4, 0.8, n)
            rays[11] = np.where(np.arange(n) % 3 == 0, 1.5, 1.0)
            block = rays[:8].reshape(2, 4, n)
            hits, _ = surf.intersect(block)
            masked = np.where(hits > 0, hits, np.inf)
            first = np.min(masked, axis=0)
            has = np.isfinite(first)
            sel = np.flatnonzero(has)
            sub = rays[:, sel].copy()
            sub[0:4] += sub[4:8] * first[sel]
            rs = sub.copy().view(pyrayt.RaySet)
            with np.errstate(all="ignore"):
                traced = np.array(material.trace(surf, rs))
            key = f"trace_{kind}_{mname}"
            for kk, vv in snapshot([surf]).items():
                out[f"{key}__{kk}"] = vv
            out[f"{key}__in"] = sub
            out[f"{key}__out"] = traced
    path = os.path.join(HERE, "shading.npz")
    np.savez_compressed(path, **out)
    print(f"shading              -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def source_vectors():
    out = {}
    c = pyrayt.components
    recipes = {
        "line": lambda: c.LineOfRays(spacing=0.1, wavelength=0.5).move_x(-0.5).rotate_y(-3),
        "circle": lambda: c.CircleOfRays(diameter=2.0).move(0.1, 0.2, 0.3),
        "cone": lambda: c.ConeOfRays(6).move_x(-1.9).rotate_z(10),
        "wedge": lambda: c.WedgeOfRays(30, wavelength=0.7).rotate_x(45),
    }
    for name, make in recipes.items():
        for n in (1, 7, 100):
            out[f"{name}_{n}"] = np.array(make().generate_rays(n))
    path = os.path.join(HERE, "sources.npz")
    np.savez_compressed(path, **out)
    print(f"sources              -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def config2_summary(n=1_000_000, seed=1234, sample=4096):
    """Summary (not the 360 MB frame) of the north-star run: rows per generation x surface,
    ids of the rays that skip the second lens surface, column checksums.
    seed 1234 is BASELINE config 2 itself (config2_1m_summary.npz); bench.py rotates its timed steps through the
    ray sets of seeds 1234 ... 1237 and checks the rows of EVERY one of them against the reference's summary of it
    (config2_1m_summary_seed<seed>.npz, a smaller point-wise sample each)."""
    reset_ids()
    components, rays = scenes.config2(API, n, seed=seed)
    tracer = pyrayt.RayTracer(_PresetSource(rays), components, rays_per_source=n, generation_limit=10)
    frame = tracer.trace().to_numpy(dtype=float)
    gens = frame[:, 0].astype(np.int64)
    surf = frame[:, 5].astype(np.int64)
    pairs, counts = np.unique(np.stack((gens, surf)), axis=1, return_counts=True)
    detector = max(sid for comp in components for sid, _ in comp.surface_ids)
    q5 = frame[(gens == 1) & (surf == detector), 4].astype(np.int64)
    out = {
        "n": np.int64(n), "rows": np.int64(frame.shape[0]),
        "gen_surface_pairs": pairs, "gen_surface_counts": counts, "q5_ids": q5,
        "column_sums": frame.sum(axis=0), "column_abs_sums": np.abs(frame).sum(axis=0),
        "surface_checksum": np.int64(int((surf * (gens + 1)).sum())),
        # a deterministic 4096-row sample for point-wise comparison
        "sample_rows": frame[:: max(1, frame.shape[0] // sample)][:sample],
        "sample_index": np.arange(0, frame.shape[0], max(1, frame.shape[0] // sample))[:sample],
    }
    path = os.path.join(HERE, "config2_1m_summary.npz" if seed == 1234 else f"config2_1m_summary_seed{seed}.npz")
    np.savez_compressed(path, **out)
    print(f"config2 1M summary   seed={seed} rows={frame.shape[0]} q5={q5.tolist()} -> {os.path.basename(path)}")


def raw_primitive_vectors():
    """primitive.intersect / primitive.normal in object space, as upstream's classes return them:
    the unsorted pair (NaN for zero directions included) and the unit normals at the hit points."""
    import tinygfx.g3d.primitives as prims

    recipes = {
        "sphere": (prims.Sphere, (1.3,)), "sphere_unit": (prims.Sphere, ()),
        "cylinder": (prims.Cylinder, (0.8, -0.5, 1.2)), "plane": (prims.Plane, (3.0, 2.0)),
        "cube": (prims.Cube, ((-1.0, -0.5, -0.25), (0.5, 1.5, 2.0))), "paraboloid": (prims.Paraboloid, (0.7, 1.5)),
    }
    out = {}
    for k, (name, (cls, args)) in enumerate(recipes.items()):
        shape = cls(*args)
        rays = scenes.random_rays(3000, seed=900 + k, box=2.5)[:8].reshape(2, 4, -1)
        with np.errstate(all="ignore"):
            hits = shape.intersect(rays.copy())
            nearest = np.where(np.isfinite(hits), hits, np.inf).min(axis=0)
            on = np.isfinite(nearest)
            points = (rays[0] + nearest * rays[1])[:, on]
            normals = shape.normal(points.copy())
        out[f"{name}__rays"], out[f"{name}__hits"] = rays, hits
        out[f"{name}__points"], out[f"{name}__normals"] = points, normals
        print(f"  raw {name:12s} {int(on.sum()):5d} of {rays.shape[-1]} rays hit, "
              f"{int(np.isnan(hits).any(axis=0).sum())} NaN columns")
    path = os.path.join(HERE, "primitives_raw.npz")
    np.savez_compressed(path, **out)
    print(f"raw primitives       -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def operations_vectors():
    """tinygfx/g3d/operations.py as functions: the quadratic helpers incl. their degenerate
    branches, the column dot product, reflect / refract in every broadcasting form they accept."""
    out = {}
    rng = np.random.default_rng(41)
    k = 1024
    a = rng.normal(size=k)
    b = rng.normal(size=k) * 3
    c = rng.normal(size=k) * 2
    a[:64] = 0.0                       # linear
    a[64:96] = rng.uniform(-1e-8, 1e-8, 32)
    b[:16] = 0.0                       # constant only
    b[16:32] = rng.uniform(-1e-8, 1e-8, 16)
    c[:8] = np.abs(c[:8])              # ... with c > 0 and c <= 0
    c[8:16] = -np.abs(c[8:16])
    c[100:110] = 0.0
    out["quad_a"], out["quad_b"], out["quad_c"] = a, b, c
    with np.errstate(all="ignore"):
        out["binomial_root"] = ref_ops.binomial_root(a.copy(), b.copy(), c.copy())
        out["smallest_positive_root"] = ref_ops.smallest_positive_root(a.copy(), b.copy(), c.copy())
    # known answers of test_operations.py:150-164
    out["binomial_known"] = np.hstack((ref_ops.binomial_root(np.array([0.0]), np.array([1.0]), np.array([-2.0])),
                                       ref_ops.binomial_root(np.array([0.0]), np.array([0.0]), np.array([-1.0]))))
    m1, m2 = rng.normal(size=(4, k)), rng.normal(size=(4, k))
    out["dot_m1"], out["dot_m2"] = m1, m2
    out["dot_axis0"] = ref_ops.element_wise_dot(m1, m2, axis=0)
    out["dot_axis1"] = ref_ops.element_wise_dot(m1, m2, axis=1)
    out["dot_1d"] = np.float64(ref_ops.element_wise_dot(m1[:, 0], m2[:, 0]))
    v = np.zeros((4, k))
    v[:3] = rng.normal(size=(3, k))
    nrm = np.zeros((4, k))
    nrm[:3] = rng.normal(size=(3, k))
    nrm[:3] /= np.linalg.norm(nrm[:3], axis=0)
    out["vectors"], out["normals"] = v, nrm
    out["reflect_full"] = ref_ops.reflect(v.copy(), nrm.copy())
    out["reflect_one_normal"] = ref_ops.reflect(v.copy(), nrm[:, 3].copy())
    out["reflect_1d"] = ref_ops.reflect(v[:, 5].copy(), nrm[:, 5].copy())
    n1 = rng.choice([1.0, 1.5, 1.6, 1.3], k)
    n2 = rng.choice([1.5, 1.6, 1.7, 1.0], k)
    out["n1"], out["n2"] = n1, n2
    vin = v.copy()
    refr, n_out = ref_ops.refract(vin, nrm.copy(), n1.copy(), n2.copy())
    out["refract_vectors_after"] = vin          # upstream normalises its argument in place
    out["refracted"], out["n_refracted"] = refr, n_out
    refr, n_out = ref_ops.refract(v.copy(), nrm.copy(), 1.0, 1.5)
    out["refracted_scalar_index"], out["n_refracted_scalar_index"] = refr, n_out
    refr, n_out = ref_ops.refract(v.copy(), nrm.copy(), n1.copy(), n2.copy(), n_global=1.33)
    out["refracted_world_133"], out["n_refracted_world_133"] = refr, n_out
    path = os.path.join(HERE, "operations.npz")
    np.savez_compressed(path, **out)
    print(f"operations           -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def gooch_record(material):
    """(shade_warm, shade_cool) of gooch.py:36-37 for a surface material (tracer materials
    render with their ``_base_material``, pyrayt/materials.py:16-24)."""
    g = getattr(material, "_base_material", material)
    warm = (1 - g.alpha) * g.warm_color + g.alpha * g.base_color
    cool = (1 - g.beta) * g.cool_color + g.beta * g.base_color
    return np.concatenate((np.asarray(warm, dtype=float), np.asarray(cool, dtype=float)))


class _CanvasAxis:
    """Stands in for a matplotlib axis: keeps what draw() hands to imshow."""

    def imshow(self, image, extent=None, **kwargs):
        self.image, self.extent = np.array(image, dtype=float), np.array(extent, dtype=float)

    def set_axisbelow(self, flag):
        pass


def render_vectors():
    """Renderer fixtures: camera grid, nearest hit per pixel, Gooch-shaded and edge canvases
    (tinygfx/g3d/renderers.py), plus what draw() passes to imshow and pyrayt/utils.py values."""
    import tinygfx.g3d.renderers as ref_render
    import pyrayt.utils as ref_utils

    out = {}
    for name, recipe in scenes.RENDER_SCENES.items():
        runs = []
        for stable in (False, True):
            ref_csg.np = _StableNumpy() if stable else np
            try:
                reset_ids()
                surfaces, camera, light = recipe(API)
                shaded = ref_render.ShadedRenderer(camera, surfaces, light)
                canvas = shaded.render()
                edges = ref_render.EdgeRender(camera, surfaces).render()
                lut = [surf for comp in surfaces for _, surf in comp.surface_ids]
                res = {
                    "cam_world": camera.get_world_transform(),
                    "cam_pixels": np.array(camera.get_resolution(), dtype=np.int64),
                    "cam_span": np.array(camera.get_span(), dtype=float),
                    "light": np.asarray(light, dtype=float),
                    "rays": np.array(shaded._rays),
                    "t": shaded._hit_distances.copy(),
                    "surf": shaded._hit_surfaces.astype(np.int64),
                    "shaded": canvas, "edges": edges,
                    "gooch": np.array([gooch_record(s.material) for s in lut]).reshape(-1, 8),
                }
                res.update(snapshot(surfaces))
            finally:
                ref_csg.np = np
            runs.append(res)
        b, a = runs
        assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a), (name, "argsort sensitive")
        for k, v in a.items():
            out[f"{name}__{k}"] = v
        hit = a["surf"] >= 0
        print(f"  render {name:10s} {a['cam_pixels'][0]}x{a['cam_pixels'][1]} px, {int(hit.sum())} hit, "
              f"{int((a['t'][hit] < 0).sum())} behind the camera, {int(a['edges'][..., 3].sum())} edge px")
    # draw(): what reaches imshow (renderers.py:251-349)
    for view in ("xy", "xz"):
        for shaded in (True, False):
            reset_ids()
            axis = _CanvasAxis()
            ref_render.draw(scenes.optical_bench(API), view=view, axis=axis, shaded=shaded, resolution=64)
            key = f"draw_{view}_{'shaded' if shaded else 'edges'}"
            out[key + "__image"], out[key + "__extent"] = axis.image, axis.extent
    reset_ids()
    axis = _CanvasAxis()
    ref_render.draw(scenes.optical_bench(API), view="xy", axis=axis, shaded=True, resolution=48,
                    bounds=((-3, -2, -1), (4, 2, 1)))
    out["draw_bounds__image"], out["draw_bounds__extent"] = axis.image, axis.extent
    # pyrayt/utils.py
    waves = np.linspace(0.3, 0.8, 201)
    out["utils__wavelengths"] = waves
    out["utils__rgb"] = ref_utils.wavelength_to_rgb(waves)
    out["utils__rgb_gamma"] = ref_utils.wavelength_to_rgb(waves, gamma=1.7)
    out["utils__lensmakers"] = np.array([ref_utils.lensmakers_equation(2, -2, 1.5, 0.25),
                                         ref_utils.lensmakers_equation(40, -200, 1.62, 5)])
    path = os.path.join(HERE, "render.npz")
    np.savez_compressed(path, **out)
    print(f"render               -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def render_ray_vectors(seeds=range(12)):
    """The renderers' nearest-hit rule (tinygfx/g3d/renderers.py:70-90, the reference's own _st_propagate) for
    arbitrary rays over crowds of random parts (tests/scenes.py render_ray_case): what it selects when no entry is
    positive -- a parameter behind the origin, or the -inf entry of a slab / linear branch -- and whose id it reports."""
    import tinygfx.g3d.renderers as ref_render

    out = {}
    for seed in seeds:
        runs = []
        for stable in (False, True):
            ref_csg.np = _StableNumpy() if stable else np
            try:
                reset_ids()
                parts, rays = scenes.render_ray_case(API, seed)
                renderer = ref_render.EdgeRender(None, parts)
                renderer._rays = rays.reshape(2, 4, -1).copy()
                with np.errstate(all="ignore"):
                    renderer._st_propagate()
                res = {"rays": rays, "t": renderer._hit_distances.copy(), "surf": renderer._hit_surfaces.astype(np.int64)}
                res.update(snapshot(parts))
            finally:
                ref_csg.np = np
            runs.append(res)
        b, a = runs
        # kept: the rays on which the default and the stable argsort agree (ties between coincident surfaces are
        # upstream's to break), of those every ray of the degenerate families, every ray whose pick is not a
        # positive parameter, and a seeded sample of the rest
        same = np.array([np.array_equal(a["t"][k], b["t"][k], equal_nan=True) and a["surf"][k] == b["surf"][k]
                         for k in range(len(a["t"]))])
        odd = np.zeros(len(same), dtype=bool)
        odd[:460] = True
        odd |= ~(a["t"] > 0)
        odd |= np.random.default_rng(seed).random(len(same)) < 0.1
        keep = np.nonzero(same & odd)[0]
        a["rays"], a["t"], a["surf"] = np.ascontiguousarray(a["rays"][:, keep]), a["t"][keep], a["surf"][keep]
        assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a if k not in ("rays", "t", "surf")), seed
        for k, v in a.items():
            out[f"case{seed}__{k}"] = v
        print(f"  render rays {seed:2d}: {len(a['roots'])} parts, {len(keep)} rays kept ({int((~same).sum())} argsort-sensitive dropped), "
              f"{int((a['surf'] >= 0).sum())} hit, {int(np.isneginf(a['t']).sum())} at -inf, "
              f"{int(((a['t'] < 0) & np.isfinite(a['t'])).sum())} behind")
    path = os.path.join(HERE, "render_rays.npz")
    np.savez_compressed(path, **out)
    print(f"render rays          -> {os.path.basename(path)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    which = set(sys.argv[1:]) or {"scenes", "primitives", "csg", "shading", "sources", "summary", "render", "operations", "raw",
                                  "adversarial", "stale", "render_rays", "custom", "summary_seeds"}
    print(f"numpy {np.__version__} pandas {pd.__version__} (reference locks numpy 1.20.2 / pandas 1.2.4)")
    if "scenes" in which:
        scene_fixture("config1", 100, 1000)
        scene_fixture("config2", 10, 2048)
        scene_fixture("config3", 10, 2048)
        scene_fixture("config4", 10, 256)
        scene_fixture("config5", 10, 2048)
        scene_fixture("two_mirrors", 10, 10)
        scene_fixture("tutorial", 10, 10)
        scene_fixture("mirrors_and_stops", 6, 4096)
        scene_fixture("stopped_lens", 10, 2048, allow_sensitive=True)
    if "adversarial" in which:  # rays on the thresholds of the engine's shortcuts (tests/scenes.py adv_*)
        for name in ("adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a", "adv_short_b",
                     "adv_short_c", "adv_bench_a", "adv_bench_b", "adv_bench_c"):
            scene_fixture(name, 6, allow_sensitive=True)
    if "custom" in which:  # user-defined materials (tests/scenes.py user_materials): Glass.index_at / TracableMaterial.trace
        scene_fixture("custom_cauchy", 10, 2048)
        scene_fixture("custom_retro", 6, 10)
        scene_fixture("custom_mixed", 10, 2048)
    if "stale" in which:  # upstream's cached cull box of a right-nested tree moved after construction
        scene_fixture("stale_box", 6, 3000, allow_sensitive=True)
    if "primitives" in which:
        primitive_vectors()
    if "csg" in which:
        csg_vectors()
    if "shading" in which:
        shading_vectors()
    if "sources" in which:
        source_vectors()
    if "summary" in which:
        config2_summary()
    if "summary_seeds" in which:  # the other ray sets of bench.py's rotation
        for seed in (1235, 1236, 1237):
            config2_summary(seed=seed, sample=512)
    if "render" in which:
        render_vectors()
    if "render_rays" in which:
        render_ray_vectors()
    if "operations" in which:
        operations_vectors()
    if "raw" in which:
        raw_primitive_vectors()


if __name__ == "__main__":
    main()
