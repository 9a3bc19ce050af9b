"""ctypes binding of ``libprt_hip.so`` (C-ABI: ``include/prt.h``) + torch device buffers.

PyTorch-ROCm is plumbing here: it owns device memory and streams; every computation on the
path happens inside the HIP library.  If the library is missing, or no GPU is visible, the
functions in this module raise -- there is deliberately no CPU fallback.
"""
import ctypes
import os
import warnings
import weakref

import numpy as np

from . import scene as _scene

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRT_LIB", os.path.join(_HERE, "csrc", "libprt_hip.so"))


from . import _runtime
from ._runtime import HW_QUEUES  # (set when the package is imported, before the HIP runtime initialises)

RAY_ROWS = 13
RECORD_COLS = 15
DEFAULT_RAY_OFFSET = 1e-6

TRACE_KEEP_ABSORBED = 1
TRACE_UNFUSED = 2
TRACE_NO_HINTS = 4
TRACE_FULL_ROWS = 8
TRACE_PUBLISH_KERNEL = 16
TRACE_TEST_STALL = 32
TRACE_SYNC = 64
TRACE_COUNT_PATHS = 128
TRACE_NO_TIMING = 256
TRACE_NO_SPARSE_KEEP = 1024
TRACE_BUSY = 2048
TRACE_TICKETS = 4

# prt_scene_options (include/prt.h): how a scene is compiled / which nearest-hit kernel serves
# prt_propagate.  Zeros are the defaults; the rest are A/B and test knobs.  DEFAULT_OPTIONS and
# DEFAULT_TRACE_FLAGS are what a DeviceScene uses when its caller says nothing (the test matrix sets
# them, tests/conftest.py); the library itself reads nothing from the environment.
OPTIONS_DTYPE = np.dtype(
    [("struct_size", "<i4"), ("no_chain", "<i4"), ("no_cull", "<i4"), ("cull_min", "<i4"), ("no_groups", "<i4"),
     ("no_implied", "<i4"), ("hit_lanes", "<i4"), ("hit_staged", "<i4"), ("list_order_groups", "<i4"),
     ("one_direction", "<i4"), ("no_intervals", "<i4"), ("no_clearance", "<i4"), ("reserved", "<i4", (4,))])
assert OPTIONS_DTYPE.itemsize == 64
OPTION_NAMES = tuple(name for name in OPTIONS_DTYPE.names if name not in ("struct_size", "reserved"))
DEFAULT_OPTIONS = {}
DEFAULT_TRACE_FLAGS = 0


def options_record(options=None):
    """The prt_scene_options struct for DEFAULT_OPTIONS overlaid with `options` (a dict or None)."""
    merged = dict(DEFAULT_OPTIONS)
    merged.update(options or {})
    rec = np.zeros(1, dtype=OPTIONS_DTYPE)
    rec["struct_size"] = OPTIONS_DTYPE.itemsize
    for key, value in merged.items():
        if key not in OPTION_NAMES:
            raise ValueError(f"unknown scene option {key!r} (known: {', '.join(OPTION_NAMES)})")
        rec[key] = int(value)
    return rec

# prt_record_plan (include/prt.h): what a trace records when the caller does not want every row
PLAN_DTYPE = np.dtype(
    [("struct_size", "<i4"), ("n_surfaces", "<i4"), ("store_rows", "<i4"), ("n_groups", "<i4"), ("surfaces", "<i8", (8,)),
     ("rays_per_source", "<f8"), ("sums_out", "<u8"), ("pivots", "<u8"), ("ms_quantity", "<i4"), ("ms_transform", "<i4"),
     ("ms_about", "<f8"), ("generation_limit", "<i4"), ("columns", "<i4")])
assert PLAN_DTYPE.itemsize == 128
SINK_STATS = 12
AXIS_INTERCEPT = 15  # PRT_FRAME_AXIS_INTERCEPT

ERR_ROWS_CAP = -4
ERR_UNTRACABLE = -5
ERR_WAVELENGTH = -6
PRT_VERSION = 220  # include/prt.h: the ABI this binding was written for
TABLE_KEEP_FACTOR, TABLE_KEEP_MIN = 4, 64  # index tables keep earlier wavelengths up to this multiple of a ray set's own
UNIQUE_CAP = 4096  # distinct wavelengths looked for on the device before the host sorts the whole row


class WavelengthNotInTable(LookupError):
    """A ray's wavelength is not in the index table of the user-defined glass it hit (PRT_ERR_WAVELENGTH)."""

_lib = None


class EngineUnavailable(RuntimeError):
    pass


def _declare(lib):
    c_int, c_i64, c_p, c_d = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_double
    sig = {
        "prt_version": (c_int, []),
        "prt_last_error": (ctypes.c_char_p, []),
        "prt_device_count": (c_int, []),
        "prt_scene_create": (c_int, [c_p, c_int, c_p, c_int, c_p, c_int, c_p, c_int, c_p, ctypes.POINTER(c_p)]),
        "prt_scene_destroy": (None, [c_p]),
        "prt_scene_update": (c_int, [c_p, c_p, c_int, c_p, c_int, c_p, c_int, c_p, c_int, c_p]),
        "prt_scene_component_rows": (c_int, [c_p, c_int]),
        "prt_scene_info": (c_int, [c_p, c_p]),
        "prt_intersect": (c_int, [c_p, c_int, c_int, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p]),
        "prt_propagate": (c_int, [c_p, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p]),
        "prt_world_normals": (c_int, [c_p, c_int, c_int, c_p, c_i64, c_i64, c_p, c_p]),
        "prt_material_trace": (c_int, [c_p, c_int, c_int, c_p, c_i64, c_i64, c_p]),
        "prt_interact_workspace_bytes": (c_i64, [c_i64]),
        "prt_interact": (c_int, [c_p, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_int, c_int,
                                 c_d, c_p, c_i64, c_p, c_p, c_i64, c_p, c_p]),
        "prt_scene_set_index_tables": (c_int, [c_p, c_p, c_int, c_p, c_p, c_i64]),
        "prt_gather_hits": (c_int, [c_int, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p]),
        "prt_scatter_shaded": (c_int, [c_int, c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p]),
        "prt_unique_workspace_bytes": (c_i64, [c_i64]),
        "prt_unique_values": (c_int, [c_int, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p]),
        "prt_trace_workspace_bytes": (c_i64, [c_i64]),
        "prt_trace": (c_i64, [c_p, c_int, c_p, c_i64, c_i64, c_int, c_d, c_p, c_i64, c_p, c_p,
                              c_int, c_p]),
        "prt_trace_begin": (c_int, [c_p, c_int, c_int, c_p, c_i64, c_i64, c_int, c_d, c_p, c_i64, c_p, c_int, c_p]),
        "prt_trace_end": (c_i64, [c_p, c_int, c_int, c_p]),
        "prt_trace_batch": (c_i64, [c_p, c_int, c_p, c_i64, c_int, c_d, c_int, c_p, c_p, c_int]),
        "prt_trace_batch_busy": (c_int, [c_p, c_int, c_p]),
        "prt_trace_set_plan": (c_int, [c_p, c_int, c_int, c_p]),
        "prt_trace_stats": (c_int, [c_p, c_p]),
        "prt_trace_telemetry": (c_int, [c_p, c_p]),
        "prt_generate_rays": (c_int, [c_int, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_p]),
        "prt_camera_rays": (c_int, [c_int, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
        "prt_render_hits": (c_int, [c_p, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p]),
        "prt_gooch_shade": (c_int, [c_p, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
        "prt_gooch_mix": (c_int, [c_int, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p]),
        "prt_render": (c_int, [c_p, c_int, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
        "prt_edge_workspace_bytes": (c_i64, [c_i64, c_i64]),
        "prt_edge_canvas": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_p, c_p]),
        "prt_reflect": (c_int, [c_int, c_p, c_p, c_int, c_i64, c_i64, c_p, c_i64, c_p]),
        "prt_refract": (c_int, [c_int, c_p, c_p, c_p, c_p, c_d, c_int, c_i64, c_i64, c_p, c_i64, c_p, c_p]),
        "prt_binomial_root": (c_int, [c_int, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p]),
        "prt_smallest_positive_root": (c_int, [c_int, c_p, c_p, c_p, c_i64, c_p, c_p]),
        "prt_dot": (c_int, [c_int, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
        "prt_primitive_intersect": (c_int, [c_int, c_int, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
        "prt_primitive_normal": (c_int, [c_int, c_int, c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p]),
        "prt_array_csg": (c_int, [c_int, c_p, c_int, c_p, c_int, c_i64, c_i64, c_int, c_int, c_p, c_i64, c_p]),
        "prt_comm_unique_id": (c_int, [c_p]),
        "prt_comm_create": (c_int, [c_int, c_int, c_int, c_p, ctypes.POINTER(c_p)]),
        "prt_comm_destroy": (None, [c_p]),
        "prt_comm_info": (c_int, [c_p, c_p]),
        "prt_allgather_counts": (c_int, [c_p, c_p, c_int, c_p, c_p]),
        "prt_allgather_workspace_bytes": (c_i64, [c_int, c_int, c_i64]),
        "prt_allgather_rows": (c_int, [c_p, c_p, c_i64, c_p, c_int, c_p, c_i64, c_p, c_p]),
        "prt_place_workspace_bytes": (c_i64, [c_int, c_int]),
        "prt_place_rows": (c_int, [c_int, c_p, c_i64, c_i64, c_int, c_p, c_int, c_p, c_i64, c_p, c_p]),
        "prt_frame_reduce": (c_int, [c_int, c_p, c_i64, c_i64, c_d, c_d, c_d, c_int, c_p, c_p, c_p]),
        "prt_frame_stats_workspace_bytes": (c_i64, [c_int]),
        "prt_frame_stats": (c_int, [c_int, c_p, c_i64, c_i64, c_d, c_d, c_d, c_int, c_p, c_p, c_p]),
        "prt_frame_stats_sharded": (c_int, [c_p, c_p, c_i64, c_i64, c_d, c_d, c_d, c_int, c_p, c_p, c_p]),
        "prt_frame_pivots": (c_int, [c_int, c_p, c_int, c_p, c_p]),
        "prt_frame_finish": (c_int, [c_int, c_p, c_p, c_int, c_p, c_p]),
        "prt_frame_mean_square": (c_int, [c_int, c_p, c_i64, c_i64, c_d, c_d, c_d, c_int, c_int, c_int, c_d, c_p, c_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not match prt.h
        fn.restype = res
        fn.argtypes = args
    return lib


EXPORTED_SYMBOLS = (
    "prt_version", "prt_last_error", "prt_device_count", "prt_scene_create", "prt_scene_destroy", "prt_scene_update",
    "prt_scene_component_rows", "prt_scene_info", "prt_intersect", "prt_propagate", "prt_world_normals",
    "prt_material_trace", "prt_interact_workspace_bytes", "prt_interact", "prt_scene_set_index_tables",
    "prt_gather_hits", "prt_scatter_shaded", "prt_unique_workspace_bytes", "prt_unique_values",
    "prt_trace_workspace_bytes", "prt_trace", "prt_trace_begin", "prt_trace_end", "prt_trace_batch", "prt_trace_batch_busy", "prt_trace_set_plan", "prt_trace_stats", "prt_trace_telemetry", "prt_generate_rays",
    "prt_camera_rays", "prt_render_hits", "prt_gooch_shade", "prt_gooch_mix", "prt_render",
    "prt_edge_workspace_bytes", "prt_edge_canvas", "prt_reflect", "prt_refract", "prt_binomial_root",
    "prt_smallest_positive_root", "prt_dot", "prt_array_csg", "prt_primitive_intersect",
    "prt_primitive_normal", "prt_comm_unique_id", "prt_comm_create", "prt_comm_destroy", "prt_comm_info",
    "prt_allgather_counts", "prt_allgather_workspace_bytes", "prt_allgather_rows",
    "prt_place_workspace_bytes", "prt_place_rows", "prt_frame_reduce", "prt_frame_stats_workspace_bytes",
    "prt_frame_stats", "prt_frame_stats_sharded", "prt_frame_pivots", "prt_frame_finish", "prt_frame_mean_square",
)


def library():
    """Load (once) and return the HIP library; raise EngineUnavailable if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C pyrayt_amd/csrc`). pyrayt_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        # (the ABI changed incompatibly between versions -- argument lists grew, output blocks widened: a stale
        # libprt_hip.so must fail here, loudly and before its missing symbols are asked for, not corrupt memory later)
        try:
            lib.prt_version.restype = ctypes.c_int
            lib.prt_version.argtypes = []
            found = lib.prt_version()
        except AttributeError:
            found = None
        if found != PRT_VERSION:
            raise EngineUnavailable(f"{LIB_PATH} is ABI version {found}, this binding needs {PRT_VERSION}: "
                                    "rebuild it (make -C pyrayt_amd/csrc)")
        _lib = _declare(lib)
    return _lib


def _check(code):
    if code < 0:
        msg = library().prt_last_error().decode("utf-8", "replace")
        if code == ERR_UNTRACABLE:
            # the reference fails with AttributeError: the default GoochMaterial has no trace()
            raise AttributeError(msg or "a ray hit a surface whose material cannot be traced")
        if code == ERR_WAVELENGTH:
            raise WavelengthNotInTable(msg or "a ray's wavelength is not in the index table of the glass it hit")
        if code == -1:
            raise ValueError(msg)
        raise RuntimeError(f"libprt_hip error {code}: {msg}")
    return code


_torch_ok = None


def _torch():
    global _torch_ok
    if _torch_ok is not None:
        return _torch_ok
    import torch

    if not torch.cuda.is_available():
        raise EngineUnavailable(
            "no HIP device visible: pyrayt_amd traces on an AMD GPU only (no CPU fallback)"
        )
    _torch_ok = torch
    return torch


def to_host(tensor):
    """Device tensor -> host numpy array through page-locked memory (one DMA at PCIe rate: 57 GB/s
    measured against 6-11 GB/s for a pageable ``.cpu()``).  The array owns a block of torch's pinned
    allocator, which takes it back when the array dies, so successive frames reuse one block."""
    if not hasattr(tensor, "is_cuda"):
        return np.asarray(tensor)
    if not tensor.is_cuda:
        return tensor.numpy()
    torch = _torch()
    host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    host.copy_(tensor, non_blocking=True)
    torch.cuda.current_stream(tensor.device).synchronize()
    return host.numpy()


def _stream_ptr(torch, device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_TICKET_STREAMS = {}  # device -> the process's ticket streams on it (DeviceScene.ticket_streams)


class RecordPlan:
    """What a trace records when the caller does not want every row (``prt_record_plan``, include/prt.h).

    The reference appends one row per live ray and generation (``pyrayt/_pyrayt.py:168-186``) and its users throw
    most of them away on the next line -- ``results.loc[results['surface'] == imager.get_id()]``
    (``examples/lens_design.ipynb`` cells 11, 19, 38), the rows of the last generation (cells 12, 15, 20) -- to look at
    a spot, a focus, a merit function.  A plan tells the generation kernel while the row is still in registers:

    surfaces   ids (``surface.get_id()``) whose rows pass; None / empty: the rows of every surface
    rows       store the rows that pass (the frame is then ``frame.loc[frame.surface.isin(surfaces)]`` of the
               unfiltered trace, row for row); False: store nothing, only sum
    stats      sum the rows that pass, per generation and per group (``id // rays_per_source``), into ``plan.sums``
               -- a device (generation_limit, n_groups, 12) tensor: what ``DeviceFrame.group_stats`` and
               ``mean_square`` would read back out of the stored frame (``pyrayt_amd.frame.SinkStats`` turns it into
               the same tables)
    pivots     device (n_groups, 3) tensor subtracted from (y1, z1, axis intercept) before summing (a design loop
               passes the previous iteration's means), or None
    mean_square  (quantity, about, transform): quantity a frame column name or "axis_intercept", transform None | "sin"
    """

    def __init__(self, surfaces=None, rows=True, stats=False, rays_per_source=None, n_groups=None, pivots=None,
                 mean_square=None, generation_limit=10, columns=None):
        from .frame import COLUMNS

        # columns: names of the frame columns a stored row writes (None: all fifteen) -- a spot diagram looks at
        # ("y1", "z1"); the other rows of the record block are left alone and never cross PCIe
        self.columns = None if columns is None else tuple(sorted({COLUMNS.index(name) for name in columns}))
        if self.columns is not None and not self.columns:
            raise ValueError("a record plan that stores rows needs at least one column")
        self.surfaces = tuple(int(s) for s in (surfaces or ()))
        if len(self.surfaces) > 8:
            raise ValueError("a record plan lists at most 8 surfaces")
        self.rows = bool(rows)
        self.stats = bool(stats)
        self.rays_per_source = float(rays_per_source) if rays_per_source else 0.0
        self.n_groups = int(n_groups) if n_groups else 1
        if not self.rays_per_source and self.n_groups != 1:
            raise ValueError("one group without rays_per_source")
        self.pivots = pivots
        self.generation_limit = int(generation_limit)
        self.ms = None
        if mean_square is not None:
            quantity, about, transform = (tuple(mean_square) + (None, None))[:3]
            column = AXIS_INTERCEPT if quantity == "axis_intercept" else COLUMNS.index(quantity)
            self.ms = (column, float(about or 0.0), {None: 0, "sin": 1}[transform])
        self.sums = None  # device (generation_limit, n_groups, SINK_STATS), made when the plan is put on a ticket

    def key(self):
        return (self.surfaces, self.rows, self.stats, self.rays_per_source, self.n_groups, self.ms, self.generation_limit,
                None if self.pivots is None else self.pivots.data_ptr(), self.columns)

    def record(self, torch, device):
        rec = np.zeros(1, dtype=PLAN_DTYPE)
        rec["struct_size"] = PLAN_DTYPE.itemsize
        rec["n_surfaces"] = len(self.surfaces)
        rec["surfaces"][0, :len(self.surfaces)] = self.surfaces
        rec["store_rows"] = int(self.rows)
        rec["generation_limit"] = self.generation_limit
        rec["ms_quantity"] = -1
        rec["columns"] = 0 if self.columns is None else sum(1 << k for k in self.columns)
        if self.stats:
            if self.sums is None or self.sums.device != device:
                self.sums = torch.zeros((self.generation_limit, self.n_groups, SINK_STATS), dtype=torch.float64, device=device)
            rec["n_groups"] = self.n_groups
            rec["rays_per_source"] = self.rays_per_source
            rec["sums_out"] = self.sums.data_ptr()
            if self.pivots is not None:
                assert self.pivots.is_cuda and self.pivots.dtype == torch.float64 and self.pivots.is_contiguous()
                assert tuple(self.pivots.shape) == (self.n_groups, 3)
                rec["pivots"] = self.pivots.data_ptr()
            if self.ms is not None:
                rec["ms_quantity"], rec["ms_about"], rec["ms_transform"] = self.ms[0], self.ms[1], self.ms[2]
        return rec


class DeviceScene:
    """Owns a ``prt_scene*`` built from a SceneSnapshot."""

    def __init__(self, snapshot, options=None, trace_flags=None):
        """options: dict of prt_scene_options fields (see OPTION_NAMES) on top of DEFAULT_OPTIONS;
        trace_flags: PRT_TRACE_* bits or'ed into every trace of this scene (default DEFAULT_TRACE_FLAGS)."""
        lib = library()
        self.snapshot = snapshot
        self._handle = ctypes.c_void_p()
        self._pending = [None] * TRACE_TICKETS
        self._tables = None  # (wavelengths, [indices per table material]) as last handed to the library
        self.trace_flags = DEFAULT_TRACE_FLAGS if trace_flags is None else int(trace_flags)
        prims = np.ascontiguousarray(snapshot.prims)
        nodes = np.ascontiguousarray(snapshot.nodes)
        roots = np.ascontiguousarray(snapshot.roots)
        mats = np.ascontiguousarray(snapshot.materials)
        opts = options_record(options)
        _check(
            lib.prt_scene_create(
                prims.ctypes.data, len(prims), nodes.ctypes.data, len(nodes), roots.ctypes.data,
                len(roots), mats.ctypes.data, len(mats), opts.ctypes.data, ctypes.byref(self._handle),
            )
        )

    def update(self, snapshot):
        """Put another snapshot of the same components into this scene (``prt_scene_update``): True if
        it fitted -- device tables overwritten in place, hints and telemetry kept -- False if the
        snapshot has another shape and a new DeviceScene is needed."""
        prims = np.ascontiguousarray(snapshot.prims)
        nodes = np.ascontiguousarray(snapshot.nodes)
        roots = np.ascontiguousarray(snapshot.roots)
        mats = np.ascontiguousarray(snapshot.materials)
        rc = library().prt_scene_update(self.handle, prims.ctypes.data, len(prims), nodes.ctypes.data, len(nodes),
                                        roots.ctypes.data, len(roots), mats.ctypes.data, len(mats), None)
        if rc == 1:
            return False
        _check(rc)
        if getattr(snapshot, "table_materials", ()) or self._table_materials:
            # (the snapshot may hold another glass in the same slot, or the same glass with other coefficients: the
            # tables are evaluated again -- on the wavelengths they held -- before the next trace)
            self._stale_tables, self._tables = self._tables, None
        self.snapshot = snapshot
        return True

    # --- user-defined glasses (PRT_MAT_TABLE) ------------------------------------------------------------
    def ensure_tables(self, wavelengths):
        """Make sure the index tables of the scene's user-defined glasses (``Glass`` subclasses with their own
        ``index_at``, pyrayt/materials.py:88-99) cover ``wavelengths``: ``index_at`` of every such material is
        evaluated -- on the host, it is the user's code -- on the union of ``wavelengths`` and what the tables
        already hold, and handed to the library (``prt_scene_set_index_tables``) if anything differs from what
        it has (new wavelengths, or a material whose coefficients the caller changed since).  Cheap when nothing
        changed: a few ``index_at`` calls on a handful of wavelengths, no device work."""
        mats = self._table_materials
        if not mats:
            return
        from . import materials as matl

        wanted = np.asarray(wavelengths, dtype=float).ravel()
        wanted = np.unique(wanted[~np.isnan(wanted)])  # (a NaN wavelength is served by prt_material.coef[3])
        if self._tables is not None:
            held = self._tables[0]
            # The tables keep what they held -- a loop over a few ray sets then settles on their union -- but not
            # without bound: a loop over ever new spectra would grow them, and with them every call's index_at
            # evaluation, upload and the kernels' binary search, by one ray set per trace.  Beyond a few times what
            # this ray set needs the tables start again from it (a later miss is PRT_ERR_WAVELENGTH: rescan, repeat).
            if len(held) <= max(TABLE_KEEP_FACTOR * len(wanted), TABLE_KEEP_MIN):
                wanted = np.concatenate([held, wanted])
        lam = np.unique(wanted)  # ascending, by value
        indices = [matl.table_indices(material, lam) for _, material in mats]
        if self._tables is not None and np.array_equal(lam, self._tables[0]) and all(
                np.array_equal(a, b, equal_nan=True) for a, b in zip(indices, self._tables[1])):
            return
        n_mats = len(self.snapshot.materials)
        ranges = np.zeros((n_mats, 2), dtype=np.int64)
        for k, (slot, _) in enumerate(mats):
            ranges[slot] = (k * len(lam), len(lam))
        all_lam = np.ascontiguousarray(np.tile(lam, len(mats)))
        all_idx = np.ascontiguousarray(np.concatenate(indices)) if indices else np.zeros(0)
        _check(library().prt_scene_set_index_tables(self.handle, ranges.ctypes.data, n_mats, all_lam.ctypes.data,
                                                    all_idx.ctypes.data, len(all_lam)))
        self._tables = (lam, indices)

    def _refresh_tables(self, rays):
        """Before a trace of a scene with user-defined glasses: a first trace scans the ray set's wavelengths; every
        later one evaluates ``index_at`` again on the wavelengths the tables hold (a handful of values: cheap) and
        uploads them only if a glass answers differently now -- the caller may have changed its coefficients, or
        put another glass into the slot (``update``), since the last trace."""
        if not self._table_materials:
            return
        if self._tables is not None:
            self.ensure_tables(self._tables[0])
            return
        stale = getattr(self, "_stale_tables", None)  # (after update(): what the tables held is still what the rays carry)
        self._stale_tables = None
        self.ensure_tables(stale[0] if stale is not None else self.distinct_wavelengths(rays))

    def distinct_wavelengths(self, rays):
        """The distinct wavelengths of a device (13, n) ray set, as a host array: found on the device
        (``prt_unique_values``, nothing but the few values crosses PCIe); a ray set with more than UNIQUE_CAP of
        them -- a continuous spectrum -- has its wavelength row sorted on the host instead."""
        torch = _torch()
        lib = library()
        n, dev = rays.shape[1], rays.device
        if n == 0:
            return np.zeros(0)
        out = torch.empty(UNIQUE_CAP, dtype=torch.float64, device=dev)
        work = torch.empty(int(lib.prt_unique_workspace_bytes(UNIQUE_CAP)), dtype=torch.uint8, device=dev)
        count = ctypes.c_int64(0)
        row = rays[10]
        _check(lib.prt_unique_values(dev.index or 0, row.data_ptr(), n, out.data_ptr(), UNIQUE_CAP,
                                     ctypes.byref(count), work.data_ptr(), _stream_ptr(torch, dev)))
        if count.value <= UNIQUE_CAP:
            return to_host(out[:count.value]).copy()
        return np.unique(to_host(row.contiguous()))

    @property
    def _table_materials(self):
        """(material slot, material) of the scene's user-defined glasses (a snapshot of plain arrays has none)."""
        return getattr(self.snapshot, "table_materials", ())

    @property
    def _host_surfaces(self):
        """(primitive index, surface) of the surfaces whose material.trace() is user code."""
        return getattr(self.snapshot, "host_surfaces", ())

    @classmethod
    def from_components(cls, components, options=None):
        return cls(_scene.SceneSnapshot(components), options=options)

    @property
    def handle(self):
        if not self._handle:
            raise RuntimeError("scene already destroyed")
        return self._handle

    def component_rows(self, root):
        return _check(library().prt_scene_component_rows(self.handle, root))

    def info(self):
        """What the scene compiled to (host-only): step / slot / cull-step counts."""
        out = (ctypes.c_int64 * 10)()
        _check(library().prt_scene_info(self.handle, out))
        keys = ("primitives", "components", "trace_steps", "trace_slots", "cull_steps", "render_steps",
                "render_slots", "chain_steps", "spatial_groups", "both_directions")
        return dict(zip(keys, (int(v) for v in out)))

    def close(self):
        self._view_of = self._view = None
        self._plans = None
        self._pending = [None] * TRACE_TICKETS
        self._begin_cache = [None] * TRACE_TICKETS
        self._end_views = [None] * TRACE_TICKETS
        self._ticket_work = None
        if self._handle:
            library().prt_scene_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- record plans -------------------------------------------------------------------------
    def set_plan(self, ticket, plan, device=None):
        """Put a ``RecordPlan`` (or None: record everything, as the reference does) on one of the scene's tickets
        (``prt_trace_set_plan``; ``trace()`` runs on ticket 0).  It stays in force until replaced."""
        plans = getattr(self, "_plans", None)
        if plans is None:
            plans = self._plans = [None] * TRACE_TICKETS
        held = plans[ticket]
        if plan is None:
            if held is not None:
                _check(library().prt_trace_set_plan(self.handle, held[2], int(ticket), None))
                plans[ticket] = None
            return
        torch = _torch()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        index = dev.index or 0
        if held is not None and held[0] is plan and held[1] == plan.key() and held[2] == index:
            return
        rec = plan.record(torch, dev)
        _check(library().prt_trace_set_plan(self.handle, index, int(ticket), rec.ctypes.data))
        plans[ticket] = (plan, plan.key(), index)

    # --- the hot loop -------------------------------------------------------------------------
    def trace(self, rays, generation_limit, ray_offset=DEFAULT_RAY_OFFSET, flags=0, rows_cap=None, out=None, plan=False):
        """Run all generations on the device.

        rays: CUDA float64 tensor (13, n), row-major.  Returns (rows, counts): rows is a CUDA
        (15, total) tensor view, generation-major; counts the rows recorded per generation.
        out: optional (15, cap) CUDA float64 block to record into (a design loop that is done with
        the previous frame passes it back instead of having a new block allocated per call).
        plan: a ``RecordPlan`` (what to record: rows of some surfaces only, and / or sums instead of rows), None to
        record everything again, or False (default): whatever plan ticket 0 holds stays."""
        torch = _torch()
        lib = library()
        assert rays.is_cuda and rays.dtype == torch.float64 and rays.dim() == 2
        assert rays.shape[0] == RAY_ROWS and (rays.shape[1] == 0 or rays.stride(1) == 1)
        n = rays.shape[1]
        dev = rays.device
        limit = int(generation_limit)
        if plan is not False:
            self.set_plan(0, plan, dev)
        held = getattr(self, "_plans", None)
        held = held[0][0] if held is not None and held[0] is not None else None
        if n == 0 or limit <= 0:
            if held is not None and held.sums is not None:
                held.sums.zero_()
            return torch.empty((RECORD_COLS, 0), dtype=torch.float64, device=dev), []
        if held is not None and not held.rows and out is None and rows_cap is None:
            rows_cap = 0  # (a plan that stores no rows needs no record block)
        if out is not None:
            assert out.is_cuda and out.dtype == torch.float64 and out.dim() == 2 and out.device == dev
            assert out.shape[0] == RECORD_COLS and out.is_contiguous()
            cap = out.shape[1]
        else:
            cap = int(rows_cap) if rows_cap is not None else self._rows_cap(torch, n, limit, dev)
        work = self._workspace(torch, n, dev)
        counts = (ctypes.c_int64 * limit)()
        if self._host_surfaces:
            raise TypeError("this scene has a material with a user-defined trace(): use trace_stepwise() "
                            "(RayTracer.trace() does)")
        self._refresh_tables(rays)
        rescanned = False
        while True:
            rows = out if out is not None else torch.empty((RECORD_COLS, cap), dtype=torch.float64, device=dev)
            total = lib.prt_trace(
                self.handle, dev.index or 0, rays.data_ptr(), n, rays.stride(0), limit,
                float(ray_offset), rows.data_ptr(), cap, counts, work.data_ptr(), int(flags) | self.trace_flags,
                _stream_ptr(torch, dev),
            )
            if total == ERR_ROWS_CAP and cap < n * limit and out is None:
                cap = min(n * limit, max(cap * 4, n))  # only when HBM was too tight for the full block
                continue
            if total == ERR_WAVELENGTH and not rescanned:
                # rays with wavelengths the tables were not built for (another ray set than last time): the
                # kernels never guess an index -- evaluate index_at on what these rays carry and trace again
                rescanned = True
                self.ensure_tables(self.distinct_wavelengths(rays))
                continue
            _check(total)
            break
        if held is None:
            self._cap_hint = (n, limit, cap)
        per_generation = counts[:]
        while per_generation and per_generation[-1] == 0:
            per_generation.pop()
        if out is None:
            return rows[:, :total], per_generation
        # (slicing a tensor costs about 2 us: a loop that re-traces into its own block with the same
        # outcome gets the view it got last time)
        if getattr(self, "_view_of", None) is not out or self._view_total != total:
            self._view_of, self._view_total, self._view = out, total, out[:, :total]
        return self._view, per_generation

    # --- the hot loop in two halves: enqueue now, collect the counts later -----------------------------
    def trace_begin(self, ticket, rays, generation_limit, out, ray_offset=DEFAULT_RAY_OFFSET, flags=0, stream=None):
        """Enqueue a trace on one of the TRACE_TICKETS tickets and return at once (``prt_trace_begin``).  `out`: the (15, cap)
        CUDA float64 record block of this ticket -- tickets in flight together record into different
        blocks (and get their own workspace here).  Collect with ``trace_end(ticket)``.
        stream: the torch stream to enqueue on (default: the current one); tickets on different streams
        overlap on the device."""
        if not 0 <= int(ticket) < TRACE_TICKETS:
            raise ValueError(f"ticket out of range (0..{TRACE_TICKETS - 1})")
        # (lifetime rule for callers that pass a stream: trace_end returns when the COUNTS are on the host, the
        # rows are only ordered on that stream -- before `rays`, `out` or anything else the trace touches goes
        # back to the allocator of another stream, make that stream wait: other.wait_stream(stream))
        # Tensors this ticket has traced before (a loop re-tracing into its own blocks, or rotating through a few
        # ray sets): the call's arguments are as they were -- checking and converting them again costs as much as
        # the call itself (about 15 us, which is most of a 125k-ray step).  A handful of entries per ticket, keyed
        # by the tensor objects and validated by their addresses (an id can be reused, an address in use cannot).
        if self._table_materials:  # (user-defined glasses: a glass changed since the last trace is evaluated again)
            self._refresh_tables(rays)
        caches = self._begin_cache if hasattr(self, "_begin_cache") else None
        if caches is None or not isinstance(caches[ticket], dict):
            if caches is None:
                caches = self._begin_cache = [None] * TRACE_TICKETS
            caches[ticket] = {}
        entries = caches[ticket]
        key = (id(rays), id(out), generation_limit, ray_offset, flags, id(stream))
        cached = entries.get(key)
        if cached is not None and cached[0]() is rays and cached[1]() is out and cached[2] is stream \
                and cached[4] == (rays.data_ptr(), out.data_ptr()) \
                and cached[5] is self._ticket_work[ticket]:  # (a larger trace since then replaced the workspace)
            args = cached[3]
        else:
            torch = _torch()
            if self._host_surfaces:
                raise TypeError("this scene has a material with a user-defined trace(): use trace_stepwise()")
            assert rays.is_cuda and rays.dtype == torch.float64 and rays.dim() == 2
            assert rays.shape[0] == RAY_ROWS and rays.stride(1) == 1
            assert out.is_cuda and out.dtype == torch.float64 and out.dim() == 2 and out.device == rays.device
            assert out.shape[0] == RECORD_COLS and out.is_contiguous()
            n, dev, limit = rays.shape[1], rays.device, int(generation_limit)
            work = self._ticket_workspace(torch, ticket, n, dev)
            args = (self.handle, dev.index or 0, int(ticket), rays.data_ptr(), n, rays.stride(0), limit,
                    float(ray_offset), out.data_ptr(), out.shape[1], work.data_ptr(), int(flags) | self.trace_flags,
                    None if stream is None else ctypes.c_void_p(stream.cuda_stream))
            if len(entries) >= 16:  # (a loop over ever new tensors)
                entries.clear()
            # (weak references: the cache must not keep a caller's ray sets and record blocks alive)
            entries[key] = (weakref.ref(rays), weakref.ref(out), stream, args, (rays.data_ptr(), out.data_ptr()), work)
        if args[12] is None:  # the current stream, whatever it is now
            torch = _torch()
            args = args[:12] + (_stream_ptr(torch, rays.device),)
        _check(library().prt_trace_begin(*args))
        limit = args[6]
        counts = self._pending[ticket]
        if counts is None or len(counts[0]) != limit:
            counts = ((ctypes.c_int64 * max(limit, 1))(), None, None)
        self._pending[ticket] = (counts[0], rays, out)  # (keeps rays and the block alive while in flight)

    def trace_end(self, ticket):
        """Wait for the counts of the trace begun on `ticket` (``prt_trace_end``): (rows view, counts)."""
        counts, rays, out = self._pending[ticket]
        assert out is not None, "no trace in flight on this ticket"
        self._pending[ticket] = (counts, None, None)
        total = _check(library().prt_trace_end(self.handle, out.device.index or 0, int(ticket), counts))
        per_generation = counts[:]
        while per_generation and per_generation[-1] == 0:
            per_generation.pop()
        # (slicing a tensor costs about 2 us: the view of the same block with the same row count is kept)
        views = self._end_views if hasattr(self, "_end_views") else None
        if views is None:
            views = self._end_views = [None] * TRACE_TICKETS
        view = views[ticket]
        if view is None or view[0] is not out or view[1] != total:
            view = views[ticket] = (out, total, out[:, :total])
        return view[2], per_generation

    def trace_many(self, ray_sets, generation_limit, depth=2, ray_offset=DEFAULT_RAY_OFFSET, flags=0):
        """Trace a sequence of ray sets with `depth` traces in flight, each ticket on its own HIP stream:
        the host enqueues ahead and the kernels of different traces overlap on the device (a generation's
        workgroups leave the chip partly idle while they start up and drain; 1M-ray traces: 0.18 -> 0.15 ms
        each).  Yields (rows, counts) per ray set, in order.  `rows` is a view of one of `depth` + 1 record
        blocks used in turn: it stays valid while the NEXT result is taken (a caller may hold frame k and
        compare it with frame k + 1) and is overwritten when the one after that is asked for -- copy what has
        to live longer.  Frames equal ``trace()``'s."""
        torch = _torch()
        depth = max(1, min(int(depth), TRACE_TICKETS))
        limit = int(generation_limit)
        pending, blocks, streams = [], [None] * (depth + 1), []
        try:
            yield from self._trace_many(torch, ray_sets, limit, depth, ray_offset, flags, pending, blocks, streams)
        finally:
            for lane in pending:  # a caller that stops early: collect what is in flight, the tickets are free again
                try:
                    rows, _ = self.trace_end(lane)
                    # (prt_trace_end returns once the counts are on the host, possibly while the last kernel still
                    # stores rows: the caller's stream -- on which the blocks and ray sets will be freed and reused
                    # -- must wait for the ticket's stream first)
                    torch.cuda.current_stream(rows.device).wait_stream(streams[lane])
                except Exception:  # noqa: BLE001
                    pass

    def _trace_many(self, torch, ray_sets, limit, depth, ray_offset, flags, pending, blocks, streams):
        for k, rays in enumerate(ray_sets):
            lane = k % depth
            if len(pending) == depth:  # the ticket about to be reused: collect its trace first
                yield self._collect(torch, pending.pop(0), streams)
            # depth + 1 blocks in turn: the block of the result just handed out is not the one recorded into next
            slot = k % len(blocks)
            need = (RECORD_COLS, max(rays.shape[1], 1) * limit)
            if blocks[slot] is None or blocks[slot].shape[1] < need[1] or blocks[slot].device != rays.device:
                blocks[slot] = torch.empty(need, dtype=torch.float64, device=rays.device)
            if not streams:
                streams.extend(self.ticket_streams(rays.device, depth))
            streams[lane].wait_stream(torch.cuda.current_stream(rays.device))  # whatever produced this ray set
            self.trace_begin(lane, rays, limit, blocks[slot], ray_offset=ray_offset, flags=flags, stream=streams[lane])
            pending.append(lane)
        while pending:
            yield self._collect(torch, pending.pop(0), streams)

    def trace_batch(self, ray_sets, generation_limit, depth=2, outs=None, ray_offset=DEFAULT_RAY_OFFSET, flags=0):
        """``trace_many`` for ray sets that all exist up front, as one library call (``prt_trace_batch``): the
        loop over the tickets runs in the library (no interpreter between two launches; a tight Python loop over
        trace_begin / trace_end is as fast, a generator with per-item work in between is not).  Returns [(rows, counts), ...]; see ``TraceBatch`` to run one
        prepared batch many times."""
        batch = TraceBatch(self, ray_sets, generation_limit, depth=depth, outs=outs, ray_offset=ray_offset, flags=flags)
        batch.run()
        return batch.results()

    def _ticket_workspace(self, torch, ticket, n, dev):
        """The workspace of one ticket (kept, grown when a larger ray set comes)."""
        works = getattr(self, "_ticket_work", None)
        if works is None:
            works = self._ticket_work = [None] * TRACE_TICKETS
        work = works[ticket]
        need = int(library().prt_trace_workspace_bytes(n))
        if work is None or work.device != dev or work.numel() < need:
            work = works[ticket] = torch.empty(need, dtype=torch.uint8, device=dev)
        return work

    def ticket_streams(self, device, depth=TRACE_TICKETS):
        """The HIP streams this scene's tickets run on when traces overlap (one per ticket, made once: the
        runtime maps streams onto a handful of hardware queues -- four by default -- and two streams that
        land on one queue run their kernels one after the other, so a program should not keep making new ones)."""
        torch = _torch()
        # one pool per device for the whole process, not one set per scene: the runtime hands hardware queues to
        # streams as they are created, and in a process that has made hundreds of streams (one set per scene, say)
        # four particular ones may well sit on the same queue -- found by the queue probe of _runtime in the test
        # suite's process.  Two scenes that trace on the same ticket number at the same time share a stream and
        # take turns; tickets of one scene never do.
        made = _TICKET_STREAMS.setdefault(str(device), [])
        while len(made) < depth:
            made.append(torch.cuda.Stream(device))
        streams = made[:depth]
        # more streams than the runtime's default of four hardware queues covers (one is the null stream's): fine when
        # the queue setting was in place before the runtime initialised.  A user's own setting is taken at its word;
        # one made late (_runtime: torch was imported first, and possibly used) is checked once, by running something
        if depth + 1 > 4 and not getattr(DeviceScene, "_warned_queues", False):
            if HW_QUEUES == "user":
                short = depth + 1 > int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
            else:
                short = HW_QUEUES == "set-late" and not _runtime.queues_overlap(torch, streams, device)
            if short:
                DeviceScene._warned_queues = True
                warnings.warn(f"{depth} traces in flight need {depth + 1} hardware queues (one is the null stream's) and the "
                              "HIP runtime of this process has fewer: streams that share a queue serialise.  Set "
                              "GPU_MAX_HW_QUEUES=8 before the first HIP call of the process (importing pyrayt_amd does, "
                              "unless the runtime is already initialised).", RuntimeWarning, stacklevel=3)
        return streams

    def _collect(self, torch, lane, streams):
        rows, counts = self.trace_end(lane)
        torch.cuda.current_stream(rows.device).wait_stream(streams[lane])  # consumers on the caller's stream
        return rows, counts

    def _rows_cap(self, torch, n, limit, dev):
        """Columns of the record block.  n * limit always suffices (one row per ray and generation)
        and costs nothing until written, so it is used whenever it fits comfortably in free HBM;
        otherwise start at four generations' worth and let the ROWS_CAP retry grow it (a retry
        re-runs the trace).  The size that worked is remembered for the next identical call."""
        hint = getattr(self, "_cap_hint", None)
        if hint is not None and hint[:2] == (n, limit):
            return hint[2]
        full = n * limit
        free, _ = torch.cuda.mem_get_info(dev)
        if full * RECORD_COLS * 8 <= free // 3:
            return full
        return n * min(limit, 4)

    def _workspace(self, torch, n, dev):
        """Scratch for prt_trace (ping-pong ray sets, control words); kept between calls."""
        cached = getattr(self, "_work", None)
        if cached is not None and cached.device == dev and getattr(self, "_work_n", -1) >= n:
            return cached
        need = int(library().prt_trace_workspace_bytes(n))
        self._work_n = n
        if cached is None or cached.device != dev or cached.numel() < need:
            self._work = cached = torch.empty(need, dtype=torch.uint8, device=dev)
        return cached

    def trace_stats(self):
        out = (ctypes.c_double * 8)()
        _check(library().prt_trace_stats(self.handle, out))
        return {"generations": int(out[0]), "ray_generations": int(out[1]),
                "kernel_ms": float(out[2]), "kernel_launches": int(out[3]),
                "rows": int(out[4]), "rays_carried": int(out[5]),
                "lookback_fallbacks": int(out[6]), "variant": int(out[7])}

    def telemetry(self):
        """Counters since the scene was created (``prt_trace_telemetry``)."""
        out = (ctypes.c_int64 * 12)()
        _check(library().prt_trace_telemetry(self.handle, out))
        keys = ("lookback_fallbacks", "speculation_misses", "dense_launches", "full_rows_fallbacks",
                "counted_traces", "rays_not_well_formed", "implied_box_nodes", "exact_box_tests",
                "plan_launches", "plan_misses", "sparse_keep_launches", "plan_dense_launches")
        return dict(zip(keys, (int(v) for v in out)))

    # --- per-state entry points -----------------------------------------------------------------
    def propagate(self, rays):
        torch = _torch()
        n = rays.shape[1]
        t = torch.empty(n, dtype=torch.float64, device=rays.device)
        surf = torch.empty(n, dtype=torch.int64, device=rays.device)
        if n:
            _check(library().prt_propagate(self.handle, rays.device.index or 0, rays.data_ptr(), n,
                                           rays.stride(0), t.data_ptr(), surf.data_ptr(),
                                           _stream_ptr(torch, rays.device)))
        return t, surf

    def interact(self, rays, t, surf, generation, generation_limit, ray_offset=DEFAULT_RAY_OFFSET, shaded=None,
                 rows_out=None):
        """Returns (rows (15,k), next rays (13,k)); k == 0 when every ray is dead.
        shaded: CUDA (13, n) block holding, for the rays that hit a surface with a user-defined ``trace()``, what
        that ``trace()`` returned (``scatter_shaded``); rows_out: a (15, >= n) view to record into."""
        torch = _torch()
        lib = library()
        n = rays.shape[1]
        dev = rays.device
        nxt = torch.empty((RAY_ROWS, n), dtype=torch.float64, device=dev)
        rows = rows_out if rows_out is not None else torch.empty((RECORD_COLS, n), dtype=torch.float64, device=dev)
        assert rows.shape[0] == RECORD_COLS and rows.shape[1] >= n and rows.stride(1) in (0, 1)
        n_live = torch.zeros(1, dtype=torch.int64, device=dev)
        if n:
            work = torch.empty(int(lib.prt_interact_workspace_bytes(n)), dtype=torch.uint8, device=dev)
            _check(lib.prt_interact(self.handle, dev.index or 0, rays.data_ptr(), n, rays.stride(0),
                                    t.data_ptr(), surf.data_ptr(), nxt.data_ptr(), nxt.stride(0),
                                    int(generation), int(generation_limit), float(ray_offset),
                                    rows.data_ptr(), rows.stride(0), n_live.data_ptr(),
                                    shaded.data_ptr() if shaded is not None else None,
                                    shaded.stride(0) if shaded is not None else 0,
                                    work.data_ptr(), _stream_ptr(torch, dev)))
        k = int(n_live.item())
        if k < 0:
            _check(k)
        return rows[:, :k], nxt[:, :k]

    # --- materials with a user-defined trace() (PRT_MAT_HOST) -------------------------------------------------
    def gather_hits(self, rays, t, surf, surface_id):
        """The rays whose nearest hit is ``surface_id``, in ray order, origins advanced to the hit point: what
        upstream hands to ``surface.material.trace`` (pyrayt/_pyrayt.py:401-410).  Returns (subset (13,k) CUDA,
        index (k) CUDA int64: their columns in ``rays``)."""
        torch = _torch()
        lib = library()
        n, dev = rays.shape[1], rays.device
        subset = torch.empty((RAY_ROWS, n), dtype=torch.float64, device=dev)
        index = torch.empty(n, dtype=torch.int64, device=dev)
        count = ctypes.c_int64(0)
        if n:
            work = torch.empty(int(lib.prt_interact_workspace_bytes(n)), dtype=torch.uint8, device=dev)
            _check(lib.prt_gather_hits(dev.index or 0, rays.data_ptr(), n, rays.stride(0), t.data_ptr(), surf.data_ptr(),
                                       int(surface_id), subset.data_ptr(), subset.stride(0), index.data_ptr(),
                                       ctypes.byref(count), work.data_ptr(), _stream_ptr(torch, dev)))
        return subset[:, :count.value], index[:count.value]

    def scatter_shaded(self, subset, index, shaded):
        """Columns of ``subset`` (13,k) to columns ``index`` of ``shaded`` (13,n), the block ``interact`` reads."""
        torch = _torch()
        k = subset.shape[1]
        if k:
            assert subset.stride(1) == 1 and shaded.stride(1) == 1
            _check(library().prt_scatter_shaded(shaded.device.index or 0, subset.data_ptr(), k, subset.stride(0),
                                                index.data_ptr(), shaded.data_ptr(), shaded.stride(0),
                                                _stream_ptr(torch, shaded.device)))
        return shaded

    def trace_stepwise(self, rays, generation_limit, ray_offset=DEFAULT_RAY_OFFSET):
        """The generation loop with the host in it, for scenes in which a surface's ``material.trace()`` is user
        code (pyrayt/materials.py:26-37, docs/source/reference/materials.rst:17-19).  Per generation:
        nearest hits on the device (``prt_propagate``); for every surface with such a material, in look-up-table
        order like upstream's loop (pyrayt/_pyrayt.py:401-410), the rays that hit it are gathered on the device,
        brought to the host as a ``RaySet`` with their origins on the surface, handed to
        ``surface.material.trace(surface, ray_subset)`` and what it returns goes back; then INTERACT on the device
        (``prt_interact``), which shades every other surface itself and takes those rays as given.
        Same result as ``trace()``: (rows (15, total) CUDA, rows per generation)."""
        torch = _torch()
        from .rayset import RaySet

        assert rays.is_cuda and rays.dtype == torch.float64 and rays.dim() == 2 and rays.shape[0] == RAY_ROWS
        n, dev, limit = rays.shape[1], rays.device, int(generation_limit)
        if n == 0 or limit <= 0:
            return torch.empty((RECORD_COLS, 0), dtype=torch.float64, device=dev), []
        self._refresh_tables(rays)
        rows = torch.empty((RECORD_COLS, n * limit), dtype=torch.float64, device=dev)
        current = rays if rays.stride(1) == 1 else rays.contiguous()
        counts, base = [], 0
        for generation in range(limit):
            n_cur = current.shape[1]
            if n_cur == 0:
                break
            t, surf = self.propagate(current)
            shaded = None
            for _, surface in self._host_surfaces:
                subset, index = self.gather_hits(current, t, surf, surface.get_id())
                if subset.shape[1] == 0:  # (upstream: `if np.any(surface_mask)`)
                    continue
                handed = to_host(subset).copy().view(RaySet)
                answer = surface.material.trace(surface, handed)
                if answer is None:
                    raise TypeError(f"{type(surface.material).__name__}.trace() returned None: it has to return the ray set "
                                    "it was handed (pyrayt/materials.py:26-37)")
                answer = np.asarray(answer, dtype=float)
                if answer.shape != handed.shape:
                    # (upstream assigns the answer to next_ray_set[..., surface_mask]: numpy broadcasting rules)
                    answer = np.broadcast_to(answer, handed.shape)
                if shaded is None:  # (only the columns of host-shaded rays are ever read)
                    shaded = torch.empty((RAY_ROWS, n_cur), dtype=torch.float64, device=dev)
                self.scatter_shaded(torch.from_numpy(np.ascontiguousarray(answer)).to(dev), index, shaded)
                if self._table_materials:  # a trace() that changes wavelengths: the glasses must know them
                    self.ensure_tables(np.unique(answer[10]))
            got, nxt = self.interact(current, t, surf, generation, limit, ray_offset, shaded=shaded,
                                     rows_out=rows[:, base:])
            k = got.shape[1]
            if k == 0:  # every ray is dead: nothing is recorded for this generation (_pyrayt.py:424-425)
                break
            counts.append(k)
            base += k
            current = nxt
        return rows[:, :base], counts

    def intersect(self, root, rays8):
        """rays8: CUDA (8+, n) tensor.  Returns (hits (m,n) float64, ids (m,n) int64)."""
        torch = _torch()
        n = rays8.shape[1]
        m = self.component_rows(root)
        hits = torch.empty((m, n), dtype=torch.float64, device=rays8.device)
        ids = torch.empty((m, n), dtype=torch.int64, device=rays8.device)
        if n:
            _check(library().prt_intersect(self.handle, rays8.device.index or 0, int(root),
                                           rays8.data_ptr(), n, rays8.stride(0), hits.data_ptr(),
                                           ids.data_ptr(), n, _stream_ptr(torch, rays8.device)))
        return hits, ids

    def world_normals(self, prim, points):
        torch = _torch()
        k = points.shape[1]
        out = torch.empty((4, k), dtype=torch.float64, device=points.device)
        if k:
            _check(library().prt_world_normals(self.handle, points.device.index or 0, int(prim),
                                               points.data_ptr(), k, points.stride(0),
                                               out.data_ptr(), _stream_ptr(torch, points.device)))
        return out

    def material_trace(self, prim, rays):
        torch = _torch()
        k = rays.shape[1]
        if k:
            _check(library().prt_material_trace(self.handle, rays.device.index or 0, int(prim),
                                                rays.data_ptr(), k, rays.stride(0),
                                                _stream_ptr(torch, rays.device)))
        return rays


    # --- renderers (include/prt.h: prt_render_hits / prt_gooch_shade / prt_render) ---------------
    def gooch_table(self, device):
        """The (surfaces, 8) table of blended warm / cool shades on `device` (built from the snapshot once per
        device: for a system of many parts, making and uploading it cost more than rendering the picture)."""
        torch = _torch()
        kept = getattr(self, "_gooch", None)
        if kept is None or kept[0] is not self.snapshot or kept[1] != device:
            kept = self._gooch = (self.snapshot, device, torch.from_numpy(self.snapshot.gooch_table()).to(device))
        return kept[2]

    def render_hits(self, rays8):
        """Nearest hit per ray under the renderers' rule: (t (n) float64, surface (n) int64)."""
        torch = _torch()
        n = rays8.shape[1]
        t = torch.empty(n, dtype=torch.float64, device=rays8.device)
        surf = torch.empty(n, dtype=torch.int64, device=rays8.device)
        if n:
            _check(library().prt_render_hits(self.handle, rays8.device.index or 0, rays8.data_ptr(), n,
                                             rays8.stride(0), t.data_ptr(), surf.data_ptr(),
                                             _stream_ptr(torch, rays8.device)))
        return t, surf

    def gooch_shade(self, rays8, t, surf, light, gooch=None):
        """(n,4) RGBA of the pixels whose rays hit ``surf`` at ``t``."""
        torch = _torch()
        n = rays8.shape[1]
        gooch = self.gooch_table(rays8.device) if gooch is None else gooch
        out = torch.empty((n, 4), dtype=torch.float64, device=rays8.device)
        spot = _light(light)
        if n:
            _check(library().prt_gooch_shade(self.handle, rays8.device.index or 0, rays8.data_ptr(), n,
                                             rays8.stride(0), t.data_ptr(), surf.data_ptr(),
                                             gooch.data_ptr(), spot.ctypes.data, out.data_ptr(),
                                             _stream_ptr(torch, rays8.device)))
        return out

    def render(self, camera, device, light=None, keep_hits=False):
        """One fused pass over the camera grid.  Returns (rgba (v,h,4) or None, t, surf); ``rgba``
        is produced when ``light`` is given, ``t`` / ``surf`` (flat, n) when ``keep_hits``."""
        torch = _torch()
        rec = camera_record(camera)
        h, v = int(rec["h_pixels"][0]), int(rec["v_pixels"][0])
        n = h * v
        rgba = gooch = t = surf = spot = None
        if light is not None:
            spot = _light(light)
            gooch = self.gooch_table(device)
            rgba = torch.empty((v, h, 4), dtype=torch.float64, device=device)
        if keep_hits or light is None:
            t = torch.empty(n, dtype=torch.float64, device=device)
            surf = torch.empty(n, dtype=torch.int64, device=device)
        ptr = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
        if n:
            _check(library().prt_render(self.handle, device.index or 0, rec.ctypes.data, 0, n, ptr(gooch),
                                        spot.ctypes.data if spot is not None else None, ptr(rgba),
                                        ptr(t), ptr(surf), _stream_ptr(torch, device)))
        return rgba, t, surf


# ---------------------------------------------------------------------------------------------
# renderer helpers without a scene
# ---------------------------------------------------------------------------------------------
JOB_DTYPE = np.dtype(
    [("rays", "<u8"), ("n", "<i8"), ("ld", "<i8"), ("rows_out", "<u8"), ("rows_cap", "<i8"),
     ("rows_per_generation", "<u8"), ("total", "<i8")]
)  # prt_trace_job
assert JOB_DTYPE.itemsize == 56


class TraceBatch:
    """A prepared ``prt_trace_batch`` call: the job table, record blocks, workspaces and streams of a
    sequence of traces of one scene, `depth` of them in flight.  ``run()`` traces them all (again);
    ``results()`` are the (rows, counts) of the last run.  outs: record blocks, (15, cap) each -- one per
    ray set, or fewer (at least `depth`) to be reused in turn when only the last results are wanted."""

    def __init__(self, scene, ray_sets, generation_limit, depth=2, outs=None, ray_offset=DEFAULT_RAY_OFFSET, flags=0):
        torch = _torch()
        self.scene, self.ray_sets = scene, list(ray_sets)
        self.limit, self.ray_offset = int(generation_limit), float(ray_offset)
        self.flags = int(flags) | scene.trace_flags
        self.depth = depth = max(1, min(int(depth), TRACE_TICKETS))
        count = len(self.ray_sets)
        self.jobs = np.zeros(count, dtype=JOB_DTYPE)
        self.counts = np.zeros((count, max(self.limit, 1)), dtype=np.int64)
        self.totals = None
        if count == 0:
            return
        dev = self.device = self.ray_sets[0].device
        for rays in self.ray_sets:
            assert rays.is_cuda and rays.dtype == torch.float64 and rays.dim() == 2 and rays.device == dev
            assert rays.shape[0] == RAY_ROWS and (rays.shape[1] == 0 or rays.stride(1) == 1)
        if outs is None:
            outs = [torch.empty((RECORD_COLS, max(r.shape[1], 1) * max(self.limit, 1)), dtype=torch.float64, device=dev)
                    for r in self.ray_sets]
        self.outs = list(outs)
        if len(self.outs) < min(depth, count):
            raise ValueError("a batch needs a record block per trace in flight (len(outs) >= depth)")
        for out in self.outs:
            assert out.is_cuda and out.dtype == torch.float64 and out.dim() == 2 and out.device == dev
            assert out.shape[0] == RECORD_COLS and out.is_contiguous()
        lanes_n = [max((r.shape[1] for r in self.ray_sets[lane::depth]), default=0) for lane in range(depth)]
        self.works = [scene._ticket_workspace(torch, lane, n, dev) for lane, n in enumerate(lanes_n)]
        self.streams = scene.ticket_streams(dev, depth)
        self._work_ptrs = (ctypes.c_void_p * depth)(*[w.data_ptr() for w in self.works])
        self._stream_ptrs = (ctypes.c_void_p * depth)(*[s.cuda_stream for s in self.streams])
        for k, rays in enumerate(self.ray_sets):
            out = self.outs[k % len(self.outs)]
            self.jobs[k] = (rays.data_ptr(), rays.shape[1], rays.stride(0), out.data_ptr(), out.shape[1],
                            self.counts[k].ctypes.data, 0)

    def run(self):
        """Trace every ray set of the batch; returns the rows of all of them together."""
        if len(self.ray_sets) == 0:
            return 0
        torch = _torch()
        if self.scene._table_materials:  # (user-defined glasses: see DeviceScene._refresh_tables)
            if self.scene._tables is None and getattr(self.scene, "_stale_tables", None) is None:
                # a first trace: the tables have to cover the wavelengths of EVERY ray set of the batch (a job that meets
                # one they lack ends the batch with PRT_ERR_WAVELENGTH: nothing can rescan in the middle of it)
                self.scene.ensure_tables(np.concatenate([self.scene.distinct_wavelengths(r) for r in self.ray_sets
                                                         if r.shape[1]] or [np.zeros(0)]))
            else:
                self.scene._refresh_tables(self.ray_sets[0])
        current = torch.cuda.current_stream(self.device)
        for stream in self.streams:  # whatever produced the ray sets
            stream.wait_stream(current)
        total = _check(library().prt_trace_batch(
            self.scene.handle, self.device.index or 0, self.jobs.ctypes.data, len(self.jobs), self.limit,
            self.ray_offset, self.depth, self._work_ptrs, self._stream_ptrs, self.flags))
        for stream in self.streams:  # consumers on the caller's stream
            current.wait_stream(stream)
        self.totals = self.jobs["total"].copy()
        return total

    def busy(self):
        """What a run with ``TRACE_BUSY`` in its flags measured (``prt_trace_batch_busy``): milliseconds the device
        had at least one of the batch's traces in flight (``union_ms``), the sum of the traces' own intervals, how
        many there were, and first start to last end."""
        out = (ctypes.c_double * 4)()
        _check(library().prt_trace_batch_busy(self.scene.handle, self.device.index or 0, out))
        return {"union_ms": float(out[0]), "sum_ms": float(out[1]), "traces": int(out[2]), "span_ms": float(out[3])}

    def result(self, k):
        """(rows, counts) of ray set `k` in the last run."""
        assert self.totals is not None, "run() first"
        k = range(len(self.ray_sets))[k]
        per_generation = self.counts[k, :self.limit].tolist()
        while per_generation and per_generation[-1] == 0:
            per_generation.pop()
        return self.outs[k % len(self.outs)][:, :int(self.totals[k])], per_generation

    def results(self):
        return [self.result(k) for k in range(len(self.ray_sets))]


CAMERA_DTYPE = np.dtype(
    [("world", "<f8", (16,)), ("h_pixels", "<i8"), ("v_pixels", "<i8"), ("h_width", "<f8"),
     ("v_width", "<f8")], align=True)
assert CAMERA_DTYPE.itemsize == 160


def _light(light_positions):
    """The single light the Gooch shader supports, as a contiguous float64[3]."""
    spot = np.asarray(light_positions, dtype=float)
    if spot.ndim != 1 or spot.shape[0] < 3:
        # upstream's (k, lights) branch (gooch.py:48-51) only broadcasts for exactly three
        # lights and then pairs coordinates with the wrong lights; it is not reproduced
        raise ValueError("light_positions must be one (x, y, z[, w]) position")
    return np.ascontiguousarray(spot[:3])


def camera_record(camera):
    rec = np.zeros(1, dtype=CAMERA_DTYPE)
    rec["world"][0] = np.asarray(camera.get_world_transform(), dtype=float).reshape(-1)
    rec["h_pixels"], rec["v_pixels"] = camera.get_resolution()
    rec["h_width"], rec["v_width"] = camera.get_span()
    return rec


def default_device():
    torch = _torch()
    return torch.device("cuda", torch.cuda.current_device())


def camera_rays(camera, device=None):
    """CUDA (8, n) float64: origins (rows 0-3) and unit directions (4-7) of every pixel."""
    torch = _torch()
    device = default_device() if device is None else device
    rec = camera_record(camera)
    n = int(rec["h_pixels"][0]) * int(rec["v_pixels"][0])
    out = torch.empty((8, n), dtype=torch.float64, device=device)
    if n:
        _check(library().prt_camera_rays(device.index or 0, rec.ctypes.data, 0, n, out.data_ptr(),
                                         out.stride(0), _stream_ptr(torch, device)))
    return out


def edge_canvas(surf, h_pixels, v_pixels, rings):
    """EdgeRender's picture from the flat (v*h) CUDA int64 surface ids: CUDA (v,h,4) float64."""
    torch = _torch()
    lib = library()
    out = torch.empty((v_pixels, h_pixels, 4), dtype=torch.float64, device=surf.device)
    if h_pixels * v_pixels:
        work = torch.empty(lib.prt_edge_workspace_bytes(h_pixels, v_pixels), dtype=torch.uint8,
                           device=surf.device)
        _check(lib.prt_edge_canvas(surf.device.index or 0, surf.data_ptr(), h_pixels, v_pixels,
                                   int(rings), out.data_ptr(), work.data_ptr(),
                                   _stream_ptr(torch, surf.device)))
    return out


def surface_shade(surface, rays, distances, light_positions):
    """``surface.shade(rays, distances, light_positions=...)`` -> host (4,n) RGBA."""
    rays = np.atleast_3d(np.asarray(rays, dtype=float))
    ds = DeviceScene.from_components([surface])
    try:
        torch = _torch()
        dev = _to_device(rays, 8)
        t = torch.from_numpy(np.array(np.broadcast_to(
            np.asarray(distances, dtype=float), (dev.shape[1],)))).cuda()
        surf = torch.full((dev.shape[1],), surface.get_id(), dtype=torch.int64, device=dev.device)
        return ds.gooch_shade(dev, t, surf, light_positions).cpu().numpy().T
    finally:
        ds.close()


def gooch_mix(material, rays, normals, light_positions):
    """``GoochMaterial.shade(rays, normals, light_positions)`` -> host (4,n) RGBA."""
    torch = _torch()
    points = _to_device(np.atleast_3d(np.asarray(rays, dtype=float))[0], 4)
    normals = np.asarray(normals, dtype=float)
    normals = _to_device(normals.reshape(normals.shape[0], -1))
    n = points.shape[1]
    if normals.shape[1] != n or normals.shape[0] < 3:
        raise ValueError("normals must be (>=3, n) for (2,4,n) rays")
    shade = np.ascontiguousarray(np.concatenate(material.shade_pair()))
    spot = _light(light_positions)
    out = torch.empty((4, n), dtype=torch.float64, device=points.device)
    if n:
        if normals.stride(0) != points.stride(0):
            raise ValueError("points and normals must share a leading dimension")
        _check(library().prt_gooch_mix(points.device.index or 0, points.data_ptr(), normals.data_ptr(),
                                       n, points.stride(0), shade.ctypes.data, spot.ctypes.data,
                                       out.data_ptr(), out.stride(0), _stream_ptr(torch, points.device)))
    return out.cpu().numpy()


# ---------------------------------------------------------------------------------------------
# numpy-facing conveniences used by the object API (component.intersect, get_world_normals,
# Material.trace).  Inputs are host arrays (as in the reference's own tests); they are staged
# through the device.
# ---------------------------------------------------------------------------------------------
def _to_device(array, rows=None):
    torch = _torch()
    host = np.ascontiguousarray(np.asarray(array, dtype=np.float64))
    if rows is not None:
        host = host.reshape(rows, -1)
    return torch.from_numpy(host).cuda()


def component_intersect(component, rays):
    rays = np.atleast_3d(np.asarray(rays, dtype=float))
    ds = DeviceScene.from_components([component])
    try:
        hits, ids = ds.intersect(0, _to_device(rays, 8))
        return hits.cpu().numpy(), ids.cpu().numpy()
    finally:
        ds.close()


def surface_normals(surface, positions):
    positions = np.asarray(positions, dtype=float)
    single = positions.ndim == 1
    ds = DeviceScene.from_components([surface])
    try:
        out = ds.world_normals(0, _to_device(positions.reshape(4, -1))).cpu().numpy()
        return out[:, 0] if single else out
    finally:
        ds.close()


def material_trace(material, surface, ray_set):
    """``material.trace(surface, ray_set)``: shades ``ray_set`` in place (and returns it) using
    ``material`` on ``surface`` regardless of the surface's own material attribute."""
    from .g3d.objects import TracerSurface

    if not isinstance(surface, TracerSurface):
        raise TypeError("material.trace needs a TracerSurface")
    snap = _scene.SceneSnapshot([surface], material_override=material)
    ds = DeviceScene(snap)
    try:
        dev = _to_device(np.asarray(ray_set), RAY_ROWS)
        ds.ensure_tables(np.unique(np.asarray(ray_set, dtype=float)[10]))  # (a user-defined glass: its index_at, evaluated here)
        ds.material_trace(0, dev)
        ray_set[...] = dev.cpu().numpy()
        return ray_set
    finally:
        ds.close()


# ---------------------------------------------------------------------------------------------
# device-side sources (include/prt.h prt_generate_rays)
# ---------------------------------------------------------------------------------------------
SOURCE_DTYPE = np.dtype(
    [("kind", "<i4"), ("reserved", "<i4"), ("params", "<f8", (4,)), ("wavelength", "<f8"),
     ("world", "<f8", (16,)), ("seed", "<u8")], align=True)
assert SOURCE_DTYPE.itemsize == 184


def sources_on_device(sources):
    """True if every source can emit its rays on the GPU (has a ``device_spec``)."""
    return all(getattr(src, "device_spec", None) is not None and src.device_spec() is not None
               for src in sources)


def generate_rays(sources, rays_per_source, device, lo=0, hi=None, specs=None):
    """The concatenated initial ray set of ``sources`` (pyrayt/_pyrayt.py:356-365), columns
    [lo, hi) of it, built directly in HBM: a CUDA (13, hi-lo) float64 tensor with consecutive
    ids.  Nothing crosses PCIe but the ~200-byte source descriptions."""
    torch = _torch()
    lib = library()
    n = int(rays_per_source)
    total = n * len(sources)
    hi = total if hi is None else hi
    out = torch.empty((RAY_ROWS, max(0, hi - lo)), dtype=torch.float64, device=device)
    for k, src in enumerate(sources):
        a, b = max(lo, k * n), min(hi, (k + 1) * n)
        if b <= a:
            continue
        # (specs: what RayTracer already asked the sources for -- a Lamp draws a new seed every time it is asked)
        kind, params, seed = src.device_spec() if specs is None else specs[k][:3]
        rec = np.zeros(1, dtype=SOURCE_DTYPE)
        rec["kind"], rec["wavelength"], rec["seed"] = kind, src.wavelength, seed
        rec["params"][0, : len(params)] = params
        rec["world"][0] = np.asarray(src.get_world_transform(), dtype=float).reshape(-1)
        _check(lib.prt_generate_rays(device.index or 0, rec.ctypes.data, n, a - k * n, b - a, a,
                                     out.data_ptr(), out.stride(0), a - lo, _stream_ptr(torch, device)))
    return out


# ---------------------------------------------------------------------------------------------
# tinygfx/g3d/operations.py as functions (host arrays in, host arrays out)
# ---------------------------------------------------------------------------------------------
def _ops_rows(matrix):
    if not 1 <= matrix.shape[0] <= 4:
        raise ValueError("vectors must have between 1 and 4 components")
    return matrix.shape[0]


def ops_reflect(vectors, normals):
    torch = _torch()
    v, nrm = _to_device(vectors), _to_device(normals)
    out = torch.empty_like(v)
    n = v.shape[1]
    _check(library().prt_reflect(v.device.index or 0, v.data_ptr(), nrm.data_ptr(), _ops_rows(v), n,
                                 v.stride(0), out.data_ptr(), out.stride(0), _stream_ptr(torch, v.device)))
    return out.cpu().numpy()


def ops_refract(vectors, normals, n1, n2, n_global):
    torch = _torch()
    v, nrm = _to_device(vectors), _to_device(normals)
    a, b = _to_device(n1, 1)[0], _to_device(n2, 1)[0]
    out = torch.empty_like(v)
    n = v.shape[1]
    index = torch.empty(n, dtype=torch.float64, device=v.device)
    _check(library().prt_refract(v.device.index or 0, v.data_ptr(), nrm.data_ptr(), a.data_ptr(), b.data_ptr(),
                                 n_global, _ops_rows(v), n, v.stride(0), out.data_ptr(), out.stride(0),
                                 index.data_ptr(), _stream_ptr(torch, v.device)))
    return out.cpu().numpy(), index.cpu().numpy(), v.cpu().numpy()


def ops_polynomial(entry, a, b, c):
    torch = _torch()
    da, db, dc = (_to_device(x, 1)[0] for x in (a, b, c))
    n = da.shape[0]
    lib = library()
    if entry == "prt_binomial_root":
        out = torch.empty((2, n), dtype=torch.float64, device=da.device)
        _check(lib.prt_binomial_root(da.device.index or 0, da.data_ptr(), db.data_ptr(), dc.data_ptr(), n,
                                     out.data_ptr(), out.stride(0), _stream_ptr(torch, da.device)))
    else:
        out = torch.empty(n, dtype=torch.float64, device=da.device)
        _check(lib.prt_smallest_positive_root(da.device.index or 0, da.data_ptr(), db.data_ptr(), dc.data_ptr(),
                                              n, out.data_ptr(), _stream_ptr(torch, da.device)))
    return out.cpu().numpy()


def ops_dot(m1, m2, axis):
    torch = _torch()
    a, b = _to_device(m1), _to_device(m2)
    rows, cols = a.shape
    if axis == 0:
        reduce_len, reduce_stride, out_len, out_stride = rows, a.stride(0), cols, 1
    else:
        reduce_len, reduce_stride, out_len, out_stride = cols, 1, rows, a.stride(0)
    out = torch.empty(out_len, dtype=torch.float64, device=a.device)
    _check(library().prt_dot(a.device.index or 0, a.data_ptr(), b.data_ptr(), reduce_len, reduce_stride,
                             out_len, out_stride, out.data_ptr(), _stream_ptr(torch, a.device)))
    return out.cpu().numpy()


def ops_array_csg(left, right, op, sort_output):
    torch = _torch()
    a, b = _to_device(left), _to_device(right)
    if a.shape[1] != b.shape[1]:
        raise ValueError("both arrays must describe the same rays (equal trailing dimension)")
    n = a.shape[1]
    out = torch.empty((a.shape[0] + b.shape[0], n), dtype=torch.float64, device=a.device)
    _check(library().prt_array_csg(a.device.index or 0, a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], n,
                                   max(a.stride(0), n), int(op), int(bool(sort_output)), out.data_ptr(),
                                   max(out.stride(0), n), _stream_ptr(torch, a.device)))
    return out.cpu().numpy()


def ops_primitive(entry, kind, params, block, out_rows):
    """prt_primitive_intersect / prt_primitive_normal on a host (rows, n) block."""
    torch = _torch()
    data = _to_device(block)
    n = data.shape[1]
    packed = np.ascontiguousarray(np.asarray(params, dtype=float))
    out = torch.empty((out_rows, n), dtype=torch.float64, device=data.device)
    _check(getattr(library(), entry)(data.device.index or 0, int(kind), packed.ctypes.data, data.data_ptr(), n,
                                     max(data.stride(0), n), out.data_ptr(), max(out.stride(0), n),
                                     _stream_ptr(torch, data.device)))
    return out.cpu().numpy()
