"""ctypes wrapper of oracle/prt_oracle.c (TEST INFRASTRUCTURE ONLY, see that file's header).

Same scene-dict format and the same call shapes as oracle/prt_oracle.py, so tests can run
either checker on the same inputs.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libprt_oracle.so")
_lib = None

_FIELDS = (
    ("prim_type", np.int32), ("prim_material", np.int32), ("prim_normal_scale", np.int32),
    ("prim_surface_id", np.int64), ("prim_params", np.float64), ("prim_minv", np.float64),
    ("node_op", np.int32), ("node_left", np.int32), ("node_right", np.int32),
    ("node_prim", np.int32), ("node_aabb", np.float64), ("roots", np.int32),
    ("mat_kind", np.int32), ("mat_coef", np.float64),
)


def library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        lib = ctypes.CDLL(LIB_PATH)
        lib.prt_oracle_scene.restype = ctypes.c_void_p
        lib.prt_oracle_scene.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 14
        lib.prt_oracle_scene_free.argtypes = [ctypes.c_void_p]
        lib.prt_oracle_trace.restype = ctypes.c_int64
        lib.prt_oracle_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        lib.prt_oracle_propagate.restype = None
        lib.prt_oracle_propagate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


class CScene:
    def __init__(self, scene):
        lib = library()
        self._keep = [np.ascontiguousarray(scene[name], dtype=dt) for name, dt in _FIELDS]
        counts = (len(scene["prim_type"]), len(scene["node_op"]), len(scene["roots"]), len(scene["mat_kind"]))
        self.handle = ctypes.c_void_p(
            lib.prt_oracle_scene(*counts, *[a.ctypes.data for a in self._keep]))

    def close(self):
        if self.handle:
            library().prt_oracle_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def propagate(scene, rays13):
    cs = CScene(scene)
    rays = np.ascontiguousarray(rays13, dtype=np.float64)
    n = rays.shape[1]
    t = np.empty(n)
    surf = np.empty(n, dtype=np.int64)
    library().prt_oracle_propagate(cs.handle, rays.ctypes.data, n, n, t.ctypes.data, surf.ctypes.data)
    cs.close()
    return t, surf


def trace(scene, rays13, generation_limit=10, ray_offset=1e-6):
    """(rows (R,15), rows per generation) like prt_oracle.trace."""
    cs = CScene(scene)
    rays = np.ascontiguousarray(rays13, dtype=np.float64)
    n = rays.shape[1]
    cap = max(1, n * generation_limit)
    rows = np.empty((cap, 15))
    counts = np.zeros(max(1, generation_limit), dtype=np.int64)
    total = library().prt_oracle_trace(cs.handle, rays.ctypes.data, n, n, int(generation_limit),
                                       float(ray_offset), rows.ctypes.data, cap, counts.ctypes.data)
    cs.close()
    if total == -5:
        raise AttributeError("a ray hit a surface whose material cannot be traced")
    if total < 0:
        raise RuntimeError(f"prt_oracle_trace failed: {total}")
    per_generation = [int(c) for c in counts if c > 0]
    return rows[:total].copy(), per_generation
