"""Shared test helpers: fixture loading, snapshot conversion, frame comparison."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCENE_KEYS = (
    "prim_type", "prim_material", "prim_normal_scale", "prim_surface_id", "prim_params",
    "prim_minv", "node_op", "node_left", "node_right", "node_prim", "node_aabb", "roots",
    "mat_kind", "mat_coef",
)

# parity bar of BASELINE.json's north_star: surface index bit-exact, everything else 1e-6 abs
ATOL = 1e-6


def load(name):
    with np.load(os.path.join(GOLDEN, name)) as data:
        return {k: data[k] for k in data.files}


def scene_of(fixture, prefix=""):
    """Oracle-format scene dict out of a fixture (optionally key-prefixed ``name__``)."""
    return {k: fixture[prefix + k] for k in SCENE_KEYS}


def flat_scene(snapshot):
    """pyrayt_amd.scene.SceneSnapshot -> the oracle's plain-array scene dict."""
    p, n, m = snapshot.prims, snapshot.nodes, snapshot.materials
    return {
        "prim_type": p["type"].astype(np.int32), "prim_material": p["material"].astype(np.int32),
        "prim_normal_scale": p["normal_scale"].astype(np.int32),
        "prim_surface_id": p["surface_id"].astype(np.int64),
        "prim_params": p["params"].reshape(-1, 6).astype(float),
        "prim_minv": p["minv"].reshape(-1, 16).astype(float),
        "node_op": n["op"].astype(np.int32), "node_left": n["left"].astype(np.int32),
        "node_right": n["right"].astype(np.int32), "node_prim": n["prim"].astype(np.int32),
        "node_aabb": n["aabb"].reshape(-1, 6).astype(float),
        "roots": snapshot.roots.astype(np.int32),
        "mat_kind": m["kind"].astype(np.int32), "mat_coef": m["coef"].reshape(-1, 6).astype(float),
    }


def flat_scene_with_user_materials(snapshot, ray_set_type=None):
    """flat_scene plus what the oracle needs to run a user's index_at / trace() itself: the material objects by
    slot, the surfaces whose material.trace() is user code by primitive index -- for those a stand-in whose
    get_world_normals is the oracle's own (the user's trace() must not need the GPU to be checked)."""
    from oracle import prt_oracle

    flat = flat_scene(snapshot)
    flat["user_materials"] = {slot: material for slot, material in snapshot.table_materials}
    flat["user_surfaces"] = {}
    flat["ray_set_type"] = ray_set_type

    class OracleSurface:
        def __init__(self, prim, surface):
            self._prim, self.material, self._surface = prim, surface.material, surface

        def get_id(self):
            return self._surface.get_id()

        def get_world_normals(self, positions):
            return prt_oracle.world_normals(flat, self._prim, np.asarray(positions, dtype=float).reshape(4, -1))

    for prim, surface in snapshot.host_surfaces:
        flat["user_materials"][int(snapshot.prims["material"][prim])] = surface.material
        flat["user_surfaces"][prim] = OracleSurface(prim, surface)
    return flat


class FixtureSnapshot:
    """Adapter: a fixture's plain-array scene -> the structured arrays DeviceScene uploads."""

    def __init__(self, scene):
        from pyrayt_amd.scene import MATERIAL_DTYPE, NODE_DTYPE, PRIM_DTYPE

        p = np.zeros(len(scene["prim_type"]), dtype=PRIM_DTYPE)
        p["type"], p["material"] = scene["prim_type"], scene["prim_material"]
        p["normal_scale"], p["surface_id"] = scene["prim_normal_scale"], scene["prim_surface_id"]
        p["params"], p["minv"] = scene["prim_params"], scene["prim_minv"]
        n = np.zeros(len(scene["node_op"]), dtype=NODE_DTYPE)
        n["op"], n["left"], n["right"] = scene["node_op"], scene["node_left"], scene["node_right"]
        n["prim"], n["aabb"] = scene["node_prim"], scene["node_aabb"]
        m = np.zeros(max(1, len(scene["mat_kind"])), dtype=MATERIAL_DTYPE)
        m["kind"][: len(scene["mat_kind"])] = scene["mat_kind"]
        m["coef"][: len(scene["mat_kind"])] = scene["mat_coef"]
        self.prims, self.nodes, self.materials = p, n, m
        self.roots = scene["roots"].astype(np.int32)


def snapshot_of(fixture, prefix=""):
    """A fixture's scene as the structured tables the library takes."""
    return FixtureSnapshot(scene_of(fixture, prefix))


def assert_frames_match(got, want, atol=ATOL, what="frame"):
    """Result frames (R,15): same shape, `surface` (col 5), `generation` and `id` exact, the
    float columns within atol (NaN == NaN)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    if got.size == 0:
        return
    for col in (0, 4, 5):
        assert np.array_equal(got[:, col], want[:, col]), f"{what}: exact column {col} differs"
    assert np.allclose(got, want, rtol=0, atol=atol, equal_nan=True), (
        f"{what}: max abs diff {np.nanmax(np.abs(got - want))}")
