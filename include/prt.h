/*
 * prt.h -- C-ABI of libprt_hip.so, the MI355X (gfx950) ray-propagation engine that sits
 * under PyRayT's RayTracer.trace() hot path.
 *
 * Every entry point is plain C: pointers, sizes, ints.  No torch / C++ types cross this
 * boundary.  Device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()).
 * Streams are passed as void* (a hipStream_t; NULL = the default stream).
 *
 * Each function cites the reference interface (path:line under the PyRayT tree) it replaces.
 * The reference has no FFI layer of its own (it is pure Python + numpy), so the seam is the
 * duck-typed Python interface of pyrayt/_pyrayt.py, tinygfx/g3d/world_objects.py,
 * tinygfx/g3d/csg.py and pyrayt/materials.py.  INTEGRATION.md shows the ctypes stubs a
 * maintainer would add on the reference side.
 *
 * Ownership: the caller owns every buffer.  The library copies the scene description at
 * prt_scene_create() and never retains caller pointers beyond a call (stream-ordered work
 * excepted: buffers must stay alive until the stream has drained).
 * Errors: 0 = OK, negative = error; prt_last_error() returns a thread-local message.
 * Threads: the stateless entry points (prt_reflect ... prt_generate_rays, prt_place_rows,
 * prt_frame_reduce) may be called from any thread; a prt_scene (like the reference's RayTracer,
 * _pyrayt.py:227-246, which holds its state between trace() calls) carries per-scene state between
 * calls -- trace statistics, the dense-mode hints, the host mirror the kernels publish to -- and
 * serves one caller at a time.  Concurrent traces use one scene object per host thread
 * (tools/two_stream_probe.py); a prt_comm likewise belongs to one thread.
 */
#ifndef PRT_H
#define PRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_VERSION 220 /* 0.2.2: record plans (prt_record_plan, prt_trace_set_plan); per-tile records retired (PRT_TRACE_NO_TILE_RECORDS ignored, telemetry slots 8 / 9 count plan launches / misses). 0.2.1: prt_frame_mean_square, PRT_TRACE_BUSY / prt_trace_batch_busy, prt_comm_info.  0.2.0: prt_interact takes the caller-shaded state, PRT_MAT_TABLE / PRT_MAT_HOST,
                           prt_scene_set_index_tables, prt_gather_hits / prt_scatter_shaded, prt_unique_values,
                           prt_frame_stats_sharded / prt_frame_pivots / prt_frame_finish, prt_trace_telemetry fills 12 slots.  A caller built against another version must not load this library:
                           prt_version() is there to be compared with this constant (pyrayt_amd.engine.library does). */

/* ---- ray buffer layout: pyrayt/_pyrayt.py:13-144 (RaySet) -------------------------------
 * A ray set is a row-major (13, n) float64 matrix with leading dimension `ld` (elements
 * between consecutive rows, ld >= n): one contiguous row per field, i.e. SoA. */
enum {
  PRT_ROW_OX = 0, PRT_ROW_OY = 1, PRT_ROW_OZ = 2, PRT_ROW_OW = 3,
  PRT_ROW_DX = 4, PRT_ROW_DY = 5, PRT_ROW_DZ = 6, PRT_ROW_DW = 7,
  PRT_ROW_GENERATION = 8, PRT_ROW_INTENSITY = 9, PRT_ROW_WAVELENGTH = 10,
  PRT_ROW_INDEX = 11, PRT_ROW_ID = 12,
  PRT_RAY_ROWS = 13
};

/* ---- result rows: pyrayt/_pyrayt.py:154-165 (_RayTraceDataframe.df_columns) --------------
 * A record block is a row-major (15, cap) float64 matrix (one contiguous row per DataFrame
 * column) with leading dimension ld_rows. */
enum {
  PRT_COL_GENERATION = 0, PRT_COL_INTENSITY = 1, PRT_COL_WAVELENGTH = 2, PRT_COL_INDEX = 3,
  PRT_COL_ID = 4, PRT_COL_SURFACE = 5, PRT_COL_X0 = 6, PRT_COL_Y0 = 7, PRT_COL_Z0 = 8,
  PRT_COL_X1 = 9, PRT_COL_Y1 = 10, PRT_COL_Z1 = 11, PRT_COL_XTILT = 12, PRT_COL_YTILT = 13,
  PRT_COL_ZTILT = 14,
  PRT_RECORD_COLS = 15
};

/* ---- scene snapshot ---------------------------------------------------------------------- */

/* primitive kinds: tinygfx/g3d/primitives.py (Sphere :220, Cylinder :621, Plane :422,
 * Cube :501, Paraboloid :299) */
enum {
  PRT_PRIM_SPHERE = 0,     /* params: radius */
  PRT_PRIM_CYLINDER = 1,   /* params: radius, h_min, h_max */
  PRT_PRIM_PLANE = 2,      /* params: width (x), length (y) */
  PRT_PRIM_CUBE = 3,       /* params: xmin, xmax, ymin, ymax, zmin, zmax */
  PRT_PRIM_PARABOLOID = 4  /* params: focus, height */
};

/* material kinds: pyrayt/materials.py (:41 absorber, :53 mirror, :102 BasicRefractor,
 * :121 SellmeierRefractor).  PRT_MAT_NONE is a surface whose material has no trace()
 * (the reference's default GoochMaterial, world_objects.py:341): hitting it is an error. */
enum {
  PRT_MAT_NONE = 0,
  PRT_MAT_ABSORBER = 1,
  PRT_MAT_MIRROR = 2,
  PRT_MAT_CONST_INDEX = 3, /* coef[0] = n */
  PRT_MAT_SELLMEIER = 4,   /* coef = b1, b2, b3, c1, c2, c3 (wavelength in um) */
  /* The reference's extension points (pyrayt/materials.py:26-37, :88-99, docs/source/reference/materials.rst:17-19):
   * a user's Glass subclass supplies index_at(wavelength), a user's TracableMaterial subclass supplies trace(). */
  PRT_MAT_TABLE = 5,       /* a glass (Glass.trace, materials.py:70-75) whose index_at() is arbitrary host code: the
                              caller evaluates it on the distinct wavelengths of its rays and hands the library the
                              sorted (wavelength, index) pairs (prt_scene_set_index_tables); the kernels look a ray's
                              wavelength up by exact value.  coef[3] = the index for a NaN wavelength; the rest of
                              coef is owned by the library.  A wavelength that is not in the table is an error
                              (PRT_ERR_WAVELENGTH), never a guess. */
  PRT_MAT_HOST = 6         /* a material whose trace() is arbitrary host code: the caller shades the rays that hit it
                              itself (prt_gather_hits -> its trace() -> prt_scatter_shaded) and prt_interact takes
                              their post-interaction state as given.  prt_trace cannot serve such a surface: a ray
                              that hits one there is PRT_ERR_UNTRACABLE. */
};

/* CSG operations: tinygfx/g3d/csg.py:7-10 (Operation) */
enum { PRT_NODE_LEAF = 0, PRT_NODE_UNION = 1, PRT_NODE_INTERSECT = 2, PRT_NODE_DIFFERENCE = 3 };

/* one TracerSurface (world_objects.py:338-422): primitive + cached world->object matrix */
typedef struct prt_prim {
  int32_t type;          /* PRT_PRIM_* */
  int32_t material;      /* index into the material table */
  int32_t normal_scale;  /* +1 / -1: Intersectable._normal_scale (world_objects.py:305,319) */
  int32_t reserved;
  int64_t surface_id;    /* CountedObject id (world_objects.py:26-40); written to `surface` */
  double params[6];
  double minv[16];       /* row-major _object_coordinate_transform (world_objects.py:122) */
} prt_prim;

/* one node of a component tree: leaf = TracerSurface, inner = CSGSurface (csg.py:64-91) */
typedef struct prt_node {
  int32_t op;            /* PRT_NODE_* */
  int32_t left, right;   /* child node indices (inner nodes) */
  int32_t prim;          /* primitive index (leaf nodes) */
  double aabb[6];        /* inner nodes: world-space CSGSurface._aobb axis spans
                            xmin,xmax,ymin,ymax,zmin,zmax (csg.py:93-116) */
} prt_node;

typedef struct prt_material {
  int32_t kind;          /* PRT_MAT_* */
  int32_t reserved;
  double coef[6];
} prt_material;

typedef struct prt_scene prt_scene; /* opaque */

/* ---- library ------------------------------------------------------------------------------ */

int prt_version(void);
const char* prt_last_error(void);
/* number of visible HIP devices (0 if none / no driver) */
int prt_device_count(void);

/* How a scene is compiled and which kernels serve it.  Every field's zero is the default (and the
 * product's choice); the others exist for A/B measurements and for the tests, which run the parity
 * suites under each of them.  None of them changes a result.  No counterpart upstream.  The library
 * reads nothing from the environment: these options and the PRT_TRACE_* flags are all there is. */
typedef struct prt_scene_options {
  int32_t struct_size;  /* sizeof(prt_scene_options) as the caller knows it */
  int32_t no_chain;     /* 1: components run on the step interpreter, none as a register-only chain step */
  int32_t no_cull;      /* 1: no component cull steps in the trace program, no line-of-sight steps in the render program */
  int32_t cull_min;     /* components from which cull steps are compiled in (0 = the default, 3) */
  int32_t no_groups;    /* 1: no hierarchy of cull steps over groups of components */
  int32_t no_implied;   /* 1: every CSG node evaluates upstream's cull box (csg.py:126-128) exactly */
  int32_t hit_lanes;    /* nearest-hit kernel of prt_propagate and of PRT_TRACE_UNFUSED: lanes per ray,
                           0 / 1 = one ray per lane (default), 4 / 8 / 16 = surface-parallel with a
                           wavefront shuffle (t, component order) min-reduce */
  int32_t hit_staged;   /* 1: those kernels read the program from an LDS copy instead of the scalar cache */
  int32_t list_order_groups; /* 1: the cull-step hierarchy groups components in list order only (the
                           round-2 form); 0: by position in space when that is tighter */
  int32_t one_direction;     /* 1: a grouped trace program is stored once, not also in mirror image */
  int32_t no_intervals;      /* 1: chain steps whose nodes are all INTERSECT run the general CSG node algebra instead of
                                the interval form (an intersection of the leaves' [enter, exit] intervals) */
  int32_t no_clearance;      /* 1: the cylinder that cuts a lens chain to its aperture is evaluated for every wave (0: not for a
                                wave all of whose chords run inside it by a margin: a convexity argument) */
  int32_t reserved[4];
} prt_scene_options;

/* Build a scene from a snapshot.  roots[] lists the node index of every top-level component
 * in RayTracer._components order (pyrayt/_pyrayt.py:229-239); the surface look-up table of
 * _pyrayt.py:257-260 is the depth-first leaf order of those roots.  options: NULL = defaults. */
int prt_scene_create(const prt_prim* prims, int n_prims, const prt_node* nodes, int n_nodes,
                     const int32_t* roots, int n_roots, const prt_material* mats, int n_mats,
                     const prt_scene_options* options, prt_scene** out);
void prt_scene_destroy(prt_scene* scene);
/* The same components with other numbers in them -- a part moved, a radius or a glass changed: what a
 * design loop does between two RayTracer.trace() calls (examples/lens_design.ipynb; upstream simply
 * walks the mutated Python objects again, _pyrayt.py:377).  Recompiles on the host and overwrites the
 * scene's tables in place: device buffers, pinned memory, events and what the scene learnt from its
 * previous trace stay.  Returns 0, or 1 -- scene untouched -- when the snapshot does not have the old
 * one's shape (a table or program would change size): build a new scene then.  Synchronises the
 * device (the previous trace may still be reading the tables); refused while a trace is in flight.
 * options: NULL = keep the scene's. */
int prt_scene_update(prt_scene* scene, const prt_prim* prims, int n_prims, const prt_node* nodes, int n_nodes,
                     const int32_t* roots, int n_roots, const prt_material* mats, int n_mats,
                     const prt_scene_options* options);
/* Index tables of the scene's PRT_MAT_TABLE materials (Glass.index_at of a user-defined glass,
 * pyrayt/materials.py:88-99, evaluated by the caller): material m looks wavelengths up in entries
 * [ranges[2m], ranges[2m] + ranges[2m+1]) of the two HOST arrays (`total` entries each), ascending in wavelength
 * without duplicates inside a material's range; ranges of other kinds of material are ignored.  Copied; replaces
 * the previous tables; a TABLE material without entries misses every look-up.  Synchronises the device (a trace
 * may still be reading the old tables); refused while a trace is in flight. */
int prt_scene_set_index_tables(prt_scene* scene, const int64_t* ranges, int n_mats, const double* wavelengths,
                               const double* indices, int64_t total);
/* rows of the hit list component `root` returns from intersect(): 2 * (#leaves under it) */
int prt_scene_component_rows(const prt_scene* scene, int root);
/* what the scene compiled to (no counterpart upstream; host-only, needs no GPU):
 * out10 = { primitives, components, step slots of the trace program, LDS hit-list slots per ray of the
 * trace program, component cull steps in it, steps / slots of the render program, components of the
 * trace program compiled to a single register-only chain step, 1 if the cull steps are grouped by the
 * components' position in space rather than by their place in the list, 1 if the trace program is stored
 * in both directions (waves that run against its axis take the mirror image) } */
int prt_scene_info(const prt_scene* scene, int64_t* out10);

/* ---- per-state entry points (drop-ins for the reference's Python methods) ------------------ */

/* component.intersect(rays) -> (hits (m,n) sorted ascending, inf = miss ; surface ids (m,n)):
 * TracerSurface.intersect world_objects.py:360-383, CSGSurface.intersect csg.py:118-160.
 * rays: device (8+, n) matrix, rows 0-7 used.  hits_out / ids_out: device (m, n) with leading
 * dimension ld_out, m = prt_scene_component_rows().  ids are defined where the hit is finite
 * and -1 elsewhere. */
int prt_intersect(prt_scene* scene, int device, int root, const double* rays, int64_t n,
                  int64_t ld, double* hits_out, int64_t* ids_out, int64_t ld_out, void* stream);

/* RayTracer._st_propagate (pyrayt/_pyrayt.py:370-392): nearest positive hit over all
 * components.  t_out (n) float64 (inf = no hit), surf_out (n) int64 surface id (-1 = none). */
int prt_propagate(prt_scene* scene, int device, const double* rays, int64_t n, int64_t ld,
                  double* t_out, int64_t* surf_out, void* stream);

/* TracerSurface.get_world_normals (world_objects.py:401-418) for primitive `prim`:
 * points (4,k) -> normals (4,k), both device, row-major with leading dimension ld. */
int prt_world_normals(prt_scene* scene, int device, int prim, const double* points, int64_t k,
                      int64_t ld, double* normals_out, void* stream);

/* surface.material.trace(surface, ray_set) (pyrayt/materials.py:47-50, :58-62, :70-75):
 * shades, in place, every ray of the (13,k) set as having hit primitive `prim` at its current
 * origin. */
int prt_material_trace(prt_scene* scene, int device, int prim, double* rays, int64_t k,
                       int64_t ld, void* stream);

/* RayTracer._st_interact (pyrayt/_pyrayt.py:394-452) + _RayTraceDataframe.insert (:168-186):
 * advance hit rays to their hit point, shade, drop dead rays (order preserving), append one
 * record row per live ray, set generation+1, and (unless generation+1 == generation_limit)
 * re-launch by ray_offset along the new direction.
 *   rays_in (13,n) ld_in ; t/surf from prt_propagate ; rays_out (13, >= n) ld_out
 *   rows_out (15, >= n) ld_rows ; n_live_out: device int64[1] (a negative PRT_ERR_* if the device raised one)
 *   shaded (13, n) ld_shaded, or NULL: column i holds the state of ray i as the caller's own material.trace()
 *     left it (_pyrayt.py:408-410 assigns what trace() returns to all 13 rows); read only for rays that hit a
 *     PRT_MAT_HOST surface -- origin, direction, intensity, wavelength, index and id of the next state are taken
 *     from it, the generation is set as for any ray (:437) -- and ignored elsewhere
 *   workspace: device scratch of prt_interact_workspace_bytes(n) bytes
 * If every ray is dead nothing is written and *n_live_out = 0 (_pyrayt.py:424-425). */
int64_t prt_interact_workspace_bytes(int64_t n);
int prt_interact(prt_scene* scene, int device, const double* rays_in, int64_t n, int64_t ld_in,
                 const double* t, const int64_t* surf, double* rays_out, int64_t ld_out,
                 int generation, int generation_limit, double ray_offset, double* rows_out,
                 int64_t ld_rows, int64_t* n_live_out, const double* shaded, int64_t ld_shaded,
                 void* workspace, void* stream);

/* The two halves of a host-shaded interaction (pyrayt/_pyrayt.py:401-410: `surface.material.trace(surface,
 * next_ray_set[..., surface_mask])` with a user-defined trace()).
 *   prt_gather_hits     the rays whose nearest hit (surf from prt_propagate) is `surface_id`, in ray order, all 13
 *                       rows, origins advanced to the hit point (o += d t, :404-407) -> subset_out (13, >= count)
 *                       ld_subset; index_out (>= count): their columns in `rays`; *count_out: HOST, how many
 *                       (the call synchronises the stream).  subset_out / index_out may hold n columns at most.
 *                       workspace: prt_interact_workspace_bytes(n) device bytes.
 *   prt_scatter_shaded  column j of subset (13, k) -> column index[j] of shaded (13, >= max index + 1): the block
 *                       prt_interact reads.  Stream-ordered. */
int prt_gather_hits(int device, const double* rays, int64_t n, int64_t ld, const double* t, const int64_t* surf,
                    int64_t surface_id, double* subset_out, int64_t ld_subset, int64_t* index_out,
                    int64_t* count_out, void* workspace, void* stream);
int prt_scatter_shaded(int device, const double* subset, int64_t k, int64_t ld_subset, const int64_t* index,
                       double* shaded, int64_t ld_shaded, void* stream);

/* The distinct values of a device array of doubles (the wavelength row of a ray set: what Glass.index_at of a
 * user-defined glass has to be evaluated on, pyrayt/materials.py:70-75), compared bit for bit, in no particular
 * order: out (cap) device, *count_out HOST = how many there are (the call synchronises the stream); when that
 * exceeds cap only the first cap found are in `out`.  workspace: prt_unique_workspace_bytes(cap) device bytes. */
int64_t prt_unique_workspace_bytes(int64_t cap);
int prt_unique_values(int device, const double* values, int64_t n, double* out, int64_t cap, int64_t* count_out,
                      void* workspace, void* stream);

/* ---- ray sources (next to the hot path) --------------------------
 * Source.generate_rays(n) (pyrayt/components.py:481-496): object-space pattern -> 4x4 world
 * transform -> unit directions, written straight into a device ray set so that the initial
 * RaySet never crosses PCIe.  Patterns: LineOfRays :511-530, CircleOfRays :533-558,
 * ConeOfRays :561-585, WedgeOfRays :588-613 (closed form, same arithmetic as upstream) and
 * Lamp :616-654 (upstream draws from numpy's global RNG; here a counter-based generator keyed
 * by `seed` and the ray number: same distribution, not the same stream). */
enum {
  PRT_SRC_LINE = 0,   /* params: spacing */
  PRT_SRC_CIRCLE = 1, /* params: diameter */
  PRT_SRC_CONE = 2,   /* params: half angle (radians) */
  PRT_SRC_WEDGE = 3,  /* params: full angle (radians) */
  PRT_SRC_LAMP = 4    /* params: width, length, max angle (radians) */
};
typedef struct prt_source {
  int32_t kind;      /* PRT_SRC_* */
  int32_t reserved;
  double params[4];
  double wavelength; /* um */
  double world[16];  /* row-major _world_coordinate_transform (world_objects.py:97-99) */
  uint64_t seed;     /* PRT_SRC_LAMP only */
} prt_source;

/* Write rays [first, first+count) of the n_total rays this source emits into columns
 * [col_offset, col_offset+count) of the device (13, >= col_offset+count) ray set `rays_out`
 * (leading dimension ld); ray k gets id id_first + (k - first), generation 0, index 1,
 * intensity 100 (Lamp: 100 cos(theta)). */
int prt_generate_rays(int device, const prt_source* source, int64_t n_total, int64_t first,
                      int64_t count, int64_t id_first, double* rays_out, int64_t ld,
                      int64_t col_offset, void* stream);

/* ---- the whole hot loop -------------------------------------------------------------------
 * RayTracer.trace() minus source generation and DataFrame construction
 * (pyrayt/_pyrayt.py:329-339 driving :370-452).  Runs every generation on the device.
 *   rays (13,n) ld: initial ray set (read only)
 *   rows_out (15, rows_cap) ld_rows = rows_cap: generation-major record rows
 *   rows_per_generation: HOST int64[generation_limit], rows recorded by each generation
 *   workspace: device scratch of prt_trace_workspace_bytes(n) bytes.  A scene that is traced again
 *     with the same workspace address, ray count and generation limit finds the control words in it as
 *     its previous trace left them and does not re-initialise them; the library notices when another
 *     scene traced with that address in between (then it does), but not when something else wrote
 *     there: do not hand the block to anything but prt_trace between two traces
 *   flags: PRT_TRACE_* bits
 * Returns the total number of rows (>= 0) or a negative error.  PRT_ERR_ROWS_CAP if rows_cap
 * is too small (n * generation_limit rows always suffices).
 * On return the COUNTS are on the host; the record rows and the workspace are only stream-ordered: the
 * generation kernels publish their counts to host-mapped memory themselves, and the call returns as
 * soon as it has seen them -- possibly while the last kernel is still storing rows.  Whatever consumes
 * rows_out (or frees / reuses rows_out, rays or the workspace) must run on `stream` or synchronise with
 * it first; PRT_TRACE_SYNC makes the call do that itself before it returns. */
#define PRT_TRACE_KEEP_ABSORBED 1 /* carry zero-direction (absorbed) rays into the next
                                     generation exactly like _pyrayt.py:415-428 (Q3) instead of
                                     dropping them when they are absorbed; the rows are
                                     identical either way */
#define PRT_TRACE_UNFUSED 2       /* run the generation as propagate + interact kernels */
#define PRT_TRACE_NO_HINTS 4      /* do not use the dense-mode hints of the scene's previous trace
                                     (every generation runs the general look-back path, as in a first trace) */
#define PRT_TRACE_FULL_ROWS 8     /* carry all 13 state rows between generations (see "compact state") */
#define PRT_TRACE_PUBLISH_KERNEL 16 /* counts reach the host through a one-block kernel behind each batch
                                     instead of from inside the generation kernels (A/B, tests) */
#define PRT_TRACE_TEST_STALL 32   /* test hook: one tile reports an expired look-back, so that the
                                     fallback to the three-kernel path can be exercised */
#define PRT_TRACE_SYNC 64         /* hipStreamSynchronize(stream) before returning */
#define PRT_TRACE_NO_TIMING 256   /* do not bracket the generation launches with HIP events (prt_trace_stats then
                                     reports 0 ms of kernel time): two event records and one event query less per
                                     trace, which is most of what a 125k-ray trace costs the host */
/* (512 was PRT_TRACE_NO_TILE_RECORDS: the per-tile records were retired in 0.2.2 -- worth under 2 % wherever they applied
 * once the sparse-loss forms existed, profiles/r6/ab_round6.txt; the bit is ignored) */
#define PRT_TRACE_BUSY 2048         /* prt_trace_batch only: bracket every job's launches with a pair of HIP events of its own (on the job's stream) and merge the intervals behind the batch: prt_trace_batch_busy */
#define PRT_TRACE_NO_SPARSE_KEEP 1024 /* do not launch sparse-loss generations dense with their absorbed rays kept
                                     (see prt_trace_telemetry); A/B, tests */
#define PRT_TRACE_COUNT_PATHS 128 /* count, in prt_trace_telemetry, the rays that are not well formed and
                                     the CSG node evaluations that took an exact path (see there) */
int64_t prt_trace_workspace_bytes(int64_t n);
int64_t prt_trace(prt_scene* scene, int device, const double* rays, int64_t n, int64_t ld,
                  int generation_limit, double ray_offset, double* rows_out, int64_t rows_cap,
                  int64_t* rows_per_generation, void* workspace, int flags, void* stream);

/* The same in two halves, so that host work and GPU work overlap (no counterpart upstream:
 * pyrayt/_pyrayt.py:329-339 is a blocking loop).  prt_trace_begin enqueues the generations the scene's
 * previous trace needed (a first trace: a batch of four) and returns at once; prt_trace_end waits for
 * their counts, enqueues whatever the trace still needs (more generations; a repeat when a hint did
 * not hold) and returns what prt_trace returns.  prt_trace is begin + end on ticket 0.
 * A scene has PRT_TRACE_TICKETS tickets per device: traces of different tickets may be in flight
 * together.  Every ticket in flight needs its own workspace and its own record block; on one stream they
 * execute one after the other, on two streams their kernels overlap on the device (the workgroups of a
 * generation leave the chip partly idle while they start up and drain).  rays, rows_out and the
 * workspace must stay valid until prt_trace_end.  Results are those of prt_trace, bit for bit. */
#define PRT_TRACE_TICKETS 4
int prt_trace_begin(prt_scene* scene, int device, int ticket, const double* rays, int64_t n, int64_t ld,
                    int generation_limit, double ray_offset, double* rows_out, int64_t rows_cap,
                    void* workspace, int flags, void* stream);
int64_t prt_trace_end(prt_scene* scene, int device, int ticket, int64_t* rows_per_generation);

/* A sequence of traces of one scene -- a tolerance run, a source sweep, the shards of a rank -- with `depth`
 * of them in flight: job k runs on ticket k % depth with workspaces[k % depth] and on streams[k % depth]
 * (NULL: all on the null stream), exactly as a caller of prt_trace_begin / prt_trace_end would run them,
 * without the caller's interpreter between two launches (no counterpart upstream: pyrayt/_pyrayt.py:329-339
 * traces one ray set per call).  It is a convenience and keeps a caller's interpreter out of the loop; it is
 * not faster than a tight loop over the two entry points (125k-ray traces, three in flight: 25 us each either
 * way -- what bounds them is the chain of dependent launches on the device, not the host).
 *   jobs: per trace the ray set, its record block and HOST rows_per_generation[generation_limit]; `total`
 *     receives what prt_trace would have returned for it.  Jobs `depth` apart may share a record block if
 *     the caller wants only the last results (jobs in flight together may not).
 *   workspaces: `depth` blocks, each of prt_trace_workspace_bytes(largest n it will see) bytes.
 *   The ray sets must be complete before the call, or be produced on the stream their job runs on.
 * Returns the rows of all jobs together, or the first error (jobs already in flight are collected, later
 * ones are not started).  As with prt_trace the counts are on the host on return and the rows are ordered
 * on their job's stream (PRT_TRACE_SYNC: every job synchronises its stream when it is collected). */
typedef struct prt_trace_job {
  const double* rays;  /* (13, n) with leading dimension ld */
  int64_t n, ld;
  double* rows_out;    /* (15, rows_cap) */
  int64_t rows_cap;
  int64_t* rows_per_generation;
  int64_t total;       /* out */
} prt_trace_job;
int64_t prt_trace_batch(prt_scene* scene, int device, prt_trace_job* jobs, int64_t count, int generation_limit,
                        double ray_offset, int depth, void* const* workspaces, void* const* streams, int flags);
/* What the last prt_trace_batch of this scene on `device` that was given PRT_TRACE_BUSY measured (for bench.py's
 * roofline: traces in flight together overlap on the device, so the sum of their kernel times says nothing about
 * the region): out4[0] = milliseconds during which at least one of the batch's traces had launches in flight -- the
 * union of the jobs' [first launch enqueued ... last launch finished] intervals, HIP events on each job's own
 * stream --, out4[1] = the sum of those intervals, out4[2] = how many there were, out4[3] = from the earliest
 * start to the latest end.  Zeros if no such batch ran.  (A job that had to repeat an attempt is represented by its
 * last attempt.) */
int prt_trace_batch_busy(const prt_scene* scene, int device, double* out4);

/* ---- record plans: what a trace records (round 6) ------------------------------------------------------
 * The reference appends one row per live ray and generation (_RayTraceDataframe.insert, pyrayt/_pyrayt.py:168-186)
 * and the very next line of its users throws most of them away: `results.loc[results['surface'] ==
 * imager.get_id()]` (examples/lens_design.ipynb cells 11, 19, 38), `results.loc[results['generation'] ==
 * np.max(results['generation'])]` (cells 12, 15, 20) -- to look at a spot size, a focus, a merit function.  A plan
 * tells the generation kernel about it while the row is still in registers:
 *   n_surfaces > 0    only rows whose surface id is one of surfaces[] pass (0: the rows of every surface pass);
 *   store_rows        the rows that pass are stored in rows_out, generation-major / id-ascending as ever: the frame
 *                     is exactly frame.loc[frame.surface.isin(surfaces)] of the unfiltered trace, and
 *                     rows_per_generation / the return value count the stored rows; `columns` picks which of the fifteen
 *                     columns those rows write (the notebook's spot diagrams look at two).  0: nothing is stored at all
 *                     (rows_out may be NULL, rows_cap 0; the counts are 0) -- the caller wants the sums only;
 *   n_groups > 0      the rows that pass are also summed, per generation and per group (group = floor(id /
 *                     rays_per_source), pyrayt/_pyrayt.py:349-354; rays_per_source <= 0: one group), into
 *                     sums_out[(generation * n_groups + group) * PRT_SINK_STATS + k] (device memory, overwritten by
 *                     every trace of the ticket, complete when the trace's stream reaches the end of prt_trace_end):
 *                       k = 0..8  the sums of prt_frame_reduce -- count, sum (y1 - py), sum (z1 - pz), sum of their
 *                                 squares, sum (f - pf), sum (f - pf)^2 with f = the axis intercept x0 - x_tilt y0 /
 *                                 y_tilt over the rows that have one, sum wavelength, sum intensity, rows with an
 *                                 intercept -- about the caller's pivots (device (n_groups, 3): py, pz, pf; NULL:
 *                                 zeros.  A design loop passes the previous iteration's means: second moments about
 *                                 a point near the mean are well conditioned);
 *                       k = 9..11 the sums of prt_frame_mean_square for ms_quantity (a frame column 0..14 or
 *                                 PRT_FRAME_AXIS_INTERCEPT; < 0: none), ms_transform (0 none, 1 sin) and ms_about:
 *                                 rows with a finite v, sum v, sum v^2.
 *                     Sums are additive over generations (one set of pivots): "the rows of the last generation" are
 *                     the block of the highest generation whose count is not zero; prt_frame_finish turns a block's
 *                     first nine into the statistics of prt_frame_stats.
 * A plan belongs to a ticket (prt_trace = ticket 0) of a scene on a device and stays in force until replaced; NULL
 * removes it.  Traces under a plan run on the fused path only (PRT_TRACE_UNFUSED / COUNT_PATHS: PRT_ERR_ARG) and
 * learn their own dense-mode hints; without a plan nothing changes -- those kernels do not know about plans.
 * generation_limit: the largest generation_limit a trace under this plan will be given (sizes sums_out). */
#define PRT_SINK_STATS 12
typedef struct prt_record_plan {
  int32_t struct_size;     /* sizeof(prt_record_plan) */
  int32_t n_surfaces;      /* 0 .. 8 */
  int32_t store_rows;
  int32_t n_groups;
  int64_t surfaces[8];     /* surface ids (TracerSurface.get_id()) */
  double rays_per_source;
  double* sums_out;        /* DEVICE (generation_limit, n_groups, PRT_SINK_STATS) float64, or NULL when n_groups == 0 */
  const double* pivots;    /* DEVICE (n_groups, 3) or NULL */
  int32_t ms_quantity, ms_transform;
  double ms_about;
  int32_t generation_limit;
  int32_t columns;         /* which record columns a stored row writes: bit k = column PRT_COL_k; 0 = all fifteen.  A caller
                              that will look at x1, y1 only -- a spot diagram -- asks for those (0x600): the other rows of the
                              (15, rows_cap) block are left as they were, and only two columns need to cross PCIe */
} prt_record_plan;
int prt_trace_set_plan(prt_scene* scene, int device, int ticket, const prt_record_plan* plan);

/* ---- frame re-assembly across the GPUs of a node ----------------------------
 * No counterpart upstream (pyrayt/_pyrayt.py:329-339 is one Python thread).  Rank r traces the
 * contiguous id range [r n/G, (r+1) n/G) with no communication; these entry points put the per-rank
 * record blocks back into the row order of pyrayt/_pyrayt.py:168-186 (generation-major, ascending
 * ray id = rank-major inside a generation).  One process per GPU; the communicator is RCCL's
 * (resolved with dlopen at first use), bootstrapped like any NCCL communicator: rank 0 draws a
 * 128-byte id, the caller broadcasts it (torch.distributed, MPI, a file ...), every rank creates.
 *   prt_allgather_counts   (limit) rows-per-generation of every rank -> counts_all[r * limit + g];
 *                          synchronises the stream (the host sizes the assembled frame from it)
 *   prt_allgather_rows     15 grouped ncclAllGather (one per record column, straight out of the
 *                          (15, ld_rows) block) + the placement kernel; stream-ordered, no host sync;
 *                          ld_rows must be >= the largest per-rank row total
 *   prt_place_rows         the placement kernel alone, for a staging area filled by another
 *                          transport: element (column k, rank r, position p) at
 *                          staging[k * stride_col + r * stride_rank + p]                              */
typedef struct prt_comm prt_comm;
int prt_comm_unique_id(char* id128);
int prt_comm_create(int device, int world, int rank, const char* id128, prt_comm** out);
void prt_comm_destroy(prt_comm* comm);
/* what RCCL reports about the communicator: out3 = { ncclCommCount, ncclCommUserRank, device } */
int prt_comm_info(const prt_comm* comm, int* out3);
int prt_allgather_counts(prt_comm* comm, const int64_t* counts_local, int limit, int64_t* counts_all,
                         void* stream);
int64_t prt_allgather_workspace_bytes(int world, int limit, int64_t pad_rows);
int prt_allgather_rows(prt_comm* comm, const double* rows, int64_t ld_rows, const int64_t* counts_all,
                       int limit, double* out, int64_t ld_out, void* workspace, void* stream);
int64_t prt_place_workspace_bytes(int world, int limit);
int prt_place_rows(int device, const double* staging, int64_t stride_rank, int64_t stride_col, int world,
                   const int64_t* counts_all, int limit, double* out, int64_t ld_out, void* workspace,
                   void* stream);

/* ---- result sink: reductions over the record block on the device ---------------------------------
 * The (15, R) record block is the frame of pyrayt/_pyrayt.py:147-186 in columns.  What upstream's
 * examples compute from that frame (examples/lens_design.ipynb cells 11-16, 19-20, 38) is: keep
 * the rows of one surface and / or generation, group them by source (id // rays_per_source,
 * _pyrayt.py:349-354), and reduce spot positions (y1, z1) and axis intercepts
 * x0 - x_tilt * y0 / y_tilt per group.  One pass over the block; out is (n_groups, 9) float64 on
 * the device: count, sum(y1 - py), sum(z1 - pz), sum((y1 - py)^2 + (z1 - pz)^2), sum(focus - pf),
 * sum((focus - pf)^2), sum(wavelength), sum(intensity), number of rows with a finite axis intercept (the
 * two focus sums run over those: a ray parallel to the axis has none, and pandas' mean skips such a
 * NaN), with pivots = DEVICE (n_groups, 3) float64, per
 * group the (py, pz, pf) its rows are measured from (a second pass about the first pass's means gives
 * well-conditioned second moments), or NULL for 0.
 * surface / generation: NaN selects every row; rays_per_source <= 0: a single group.
 * Every wave keeps the sums of the group it is in in registers (rows are ordered by id, so groups come
 * in runs) and adds them to the output when the group changes; with few groups the waves spread
 * those atomics over 64 copies of the output in a stream-ordered scratch block (hipMallocAsync) that
 * a second kernel folds.  Stream-ordered; no host synchronisation. */
int prt_frame_reduce(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                     double generation, double rays_per_source, int n_groups, const double* pivots,
                     double* out, void* stream);
/* The statistics themselves in one stream-ordered call (two prt_frame_reduce passes, the second about
 * the first one's per-group means, computed on the device): out = (n_groups, 8) float64 on the device,
 * per group count, mean y1, mean z1, rms spot radius about that centroid, mean axis intercept, its
 * standard deviation (both over the rows that have an intercept, NaN if none has), mean wavelength, mean
 * intensity (NaN in 1..7 for a group without rows).
 * workspace: prt_frame_stats_workspace_bytes(n_groups) device bytes. */
int64_t prt_frame_stats_workspace_bytes(int n_groups);
int prt_frame_stats(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                    double generation, double rays_per_source, int n_groups, double* out, void* workspace,
                    void* stream);
/* The same statistics of a frame that is spread over the ranks of a communicator -- a sharded trace whose rows were
 * NOT re-assembled (each rank holds the rows of its own id range): every rank reduces its own rows and the
 * (n_groups, 9) sums of each pass are added across the ranks with one ncclAllReduce (the second pass runs about the
 * whole frame's means, which every rank then holds).  Every rank receives the statistics of the whole frame; what
 * crosses xGMI is 72 bytes per group and pass instead of the frame (315 MB into every GPU for the north-star job).
 * Stream-ordered; same `out` and `workspace` as prt_frame_stats.  Row order plays no part: any partition of the
 * rows gives the same sums (up to the rounding of the additions). */
int prt_frame_stats_sharded(prt_comm* comm, const double* rows, int64_t ld, int64_t n_rows, double surface,
                            double generation, double rays_per_source, int n_groups, double* out, void* workspace,
                            void* stream);
/* The steps between and behind the two passes on their own, for sums added across ranks by another transport
 * (prt_frame_reduce on every rank -> add -> prt_frame_pivots -> prt_frame_reduce about them -> add ->
 * prt_frame_finish): pivots_out (n_groups, 3) = per group the means (y1, z1, axis intercept) of a first pass's sums;
 * out (n_groups, 8) = the statistics from a second pass's sums and the pivots it ran about.  All device pointers. */
int prt_frame_pivots(int device, const double* sums, int n_groups, double* pivots_out, void* stream);
int prt_frame_finish(int device, const double* sums, const double* pivots, int n_groups, double* out, void* stream);
/* Mean squares of one quantity of the frame: the merit functions of examples/lens_design.ipynb, all of the form
 * np.mean(np.square(f(rows) - c)) over a selection of rows (cell 20, the coma metric: np.sin(ray_set['y_tilt']) -
 * np.sin(angle) over the rows of the last generation; cells 28 / 32: the axis intercept minus the design focus).
 * quantity: a frame column (PRT_COL_*, 0..14) or PRT_FRAME_AXIS_INTERCEPT = x0 - x_tilt * y0 / y_tilt (cells 12, 15);
 * transform: 0 none, 1 sin; v = transform(quantity) - about.  out = (n_groups, 3) float64 on the device, per group
 * the rows counted, sum v, sum v^2, over the rows that pass the surface / generation filter (NaN = every row) and
 * whose v is finite (pandas' mean skips a NaN: a ray parallel to the axis has no intercept); the sums are additive
 * over any partition of the rows, so a sharded frame adds them across ranks before dividing.  rays_per_source <= 0:
 * one group.  Stream-ordered. */
#define PRT_FRAME_AXIS_INTERCEPT 15
int prt_frame_mean_square(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                          double generation, double rays_per_source, int n_groups, int quantity, int transform,
                          double about, double* out, void* stream);

/* statistics of the trace of this scene that ended last (prt_trace / prt_trace_end; for bench.py's roofline):
 * out[0] = generations that found rays, out[1] = sum over generations of rays alive at entry,
 * out[2] = GPU milliseconds spent in generation kernels (hipEvent, on the trace stream),
 * out[3] = number of generation-kernel launches, out[4] = sum of rows recorded,
 * out[5] = sum of rays handed to the next generation,
 * out[6] = traces of this scene, so far, whose look-back gave up and which were re-run on the
 *          three-kernel path (telemetry: such a trace is correct but about twice as slow),
 * out[7] = PRT_VARIANT_* the last trace ran on. */
#define PRT_VARIANT_FUSED 1       /* one ray per lane, one fused kernel per generation */
#define PRT_VARIANT_UNFUSED 2     /* propagate / scan / interact kernels per generation */
#define PRT_VARIANT_KLANES 3      /* ... with the surface-parallel nearest-hit kernel (K lanes per ray,
                                     shuffle min-reduce; prt_scene_options.hit_lanes = 4 | 8 | 16) */
int prt_trace_stats(const prt_scene* scene, double* out8);
/* counters of this scene since it was created: out12 = { traces re-run on the three-kernel path after a
 * look-back gave up, traces repeated because a dense-mode hint did not hold, generation launches made in
 * dense mode, traces repeated with all 13 state rows (see below; at most one per scene),
 * and from the traces run with PRT_TRACE_COUNT_PATHS: how many such traces, ray-generations whose ray was
 * not well formed (see "shortcuts" in DESIGN.md: such a ray takes none), CSG node evaluations with
 * survivors under an implied cull box, ... of which evaluated upstream's box test exactly;
 * generation launches made under a record plan, traces under a plan repeated because one of the plan's own dense
 * hints did not hold (slots 8 and 9 counted the per-tile records of 0.2.0-0.2.1, retired in 0.2.2),
 * dense-mode launches that kept their absorbed rays (sparse loss, below), launches under a plan that ran dense }.
 * Dense mode: a generation in which the previous trace of the same scene (whatever its ray count)
 * recorded every ray and carried all or none of them on is launched on the assumption that it will
 * again -- every tile then knows its output position without the look-back; each tile checks the
 * assumption on its own counts and a miss repeats the trace without assumptions (results are exact
 * either way; PRT_TRACE_NO_HINTS turns the hints off for a call).
 * Sparse loss: a generation that recorded every ray and carried on all but a few absorbed ones (at most 1 in 64)
 * is launched dense as well, with PRT_TRACE_KEEP_ABSORBED in force for that launch: the absorbed rays go on,
 * direction zeroed, the way upstream carries them (_pyrayt.py:415-428), and the next generation -- which
 * compacts -- finds them dead, records nothing for them and drops them.  The rows are the same; what is saved is
 * the look-back of a generation whose few odd rays make their tiles the slowest ones, so that every tile waited
 * for a straggler (PRT_TRACE_NO_SPARSE_KEEP turns it off for a call).  Such a launch notes, per tile, how many rays it kept (a dead list
 * in the workspace), and the generation behind it -- if it loses no ray of its own -- is launched on that list:
 * its tiles' positions are "tile index x tile size minus the dead rays in front", again without a look-back.
 * Compact state: between the generations of a trace the ray state goes without its rows 3, 7 and 8
 * (origin w, direction w, generation): in a ray set that starts like RaySet's defaults
 * (pyrayt/_pyrayt.py:29-36: w = 1 / 0, generation 0) they hold 1, +0 and the generation's number in every
 * generation, bit for bit, so they are neither written nor read (24 of 104 B each way).  Generation 0
 * checks the caller's rows, every generation checks the rays it hands on; the first ray that differs
 * makes the library repeat the trace with all rows and keep doing so for this scene
 * (PRT_TRACE_FULL_ROWS forces that form for a call).
 * Lean segments (0.2.1): three more of those rows never change on this path (no built-in material touches a ray's
 * intensity, wavelength or id) and are redundant within a wave of a ray set as sources emit it.  A wave of a
 * generation whose 64 outputs land, in lane order, on the 64 columns one wave of the next generation reads checks that
 * its rays share one intensity and one wavelength (bit for bit) and that their ids count up by one from an integer
 * in [0, 2^48); then its first lane alone writes, into the first three entries of the segment's id row, that first id
 * boxed into a NaN with the tag 0x7ffb in its top 16 bits, the intensity and the wavelength -- rows 9 and 10 of the
 * segment are not written -- and the reader takes the three numbers from there with one scalar load (24 of the
 * remaining 80 B each way).  Any other wave hands the rows on as they are, so every ray set is served.  A ray whose
 * genuine id carries that tag would be misread: it raises the same repeat with all 13 rows. */
int prt_trace_telemetry(const prt_scene* scene, int64_t* out12);

/* ---- renderers (second consumer of the intersect path) ----------
 * tinygfx/g3d/renderers.py: an OrthographicCamera grid (world_objects.py:499-537) is pushed
 * through the components, then Gooch-shaded (ShadedRenderer :129-248) or edge-detected
 * (EdgeRender :11-126).  Pixels are numbered row-major over the v_pixels x h_pixels picture;
 * images are (v, h, 4) float64 RGBA, the layout both renderers return. */
typedef struct prt_camera {
  double world[16];  /* row-major camera -> world transform (world_objects.py:97-99) */
  int64_t h_pixels, v_pixels;
  double h_width, v_width; /* camera-space span of the grid along y (h) and z (v) */
} prt_camera;

/* OrthographicCamera.generate_rays (world_objects.py:519-537): rays of pixels
 * [first, first+count) into columns [0, count) of the device (8, ld) block rays_out
 * (rows 0-3 origin, 4-7 unit direction). */
int prt_camera_rays(int device, const prt_camera* camera, int64_t first, int64_t count,
                    double* rays_out, int64_t ld, void* stream);

/* EdgeRender._st_propagate / ShadedRenderer._st_propagate (renderers.py:70-92, 187-209).
 * Differs from prt_propagate in one rule: a component without a positive hit offers its
 * smallest parameter (gather from the unmasked list at the argmin of the masked one,
 * :79-83), so t_out may be negative.  rays (>= 8, n) ld; t_out (n); surf_out (n), -1 = none. */
int prt_render_hits(prt_scene* scene, int device, const double* rays, int64_t n, int64_t ld,
                    double* t_out, int64_t* surf_out, void* stream);

/* ShadedRenderer._st_interact (renderers.py:211-236) = TracerSurface.shade
 * (world_objects.py:385-399) + GoochMaterial.shade (materials/gooch.py:30-65) with one light:
 *   gooch: device (n_prims, 8) = shade_warm[4] | shade_cool[4] per primitive (gooch.py:36-37)
 *   light: HOST double[3], world position of the light
 *   rgba_out: device (n, 4); pixels with surf == -1 get (0,0,0,0).
 * Upstream's multi-light branch (gooch.py:48-51) does not broadcast for any light count
 * but 3, where it mixes up coordinates and lights; it is not offered. */
int prt_gooch_shade(prt_scene* scene, int device, const double* rays, int64_t n, int64_t ld,
                    const double* t, const int64_t* surf, const double* gooch,
                    const double* light, double* rgba_out, void* stream);

/* GoochMaterial.shade(rays, normals, light_positions) (materials/gooch.py:30-65) on caller
 * supplied points and normals, one light:  points, normals: device (>= 3, n) ld;
 * shade: HOST double[8] = shade_warm | shade_cool; light: HOST double[3];
 * rgba_out: device (4, n) ld_out -- the (4, n) layout the method returns. */
int prt_gooch_mix(int device, const double* points, const double* normals, int64_t n, int64_t ld,
                  const double* shade, const double* light, double* rgba_out, int64_t ld_out,
                  void* stream);

/* ShadedRenderer.render() / the propagate half of EdgeRender.render() for pixels
 * [first, first+count) in one kernel: camera ray -> nearest hit -> Gooch colour, nothing read
 * from HBM but the scene.  Any of rgba_out (count,4) / t_out (count) / surf_out (count) may
 * be NULL; gooch and light are needed only with rgba_out. */
int prt_render(prt_scene* scene, int device, const prt_camera* camera, int64_t first,
               int64_t count, const double* gooch, const double* light, double* rgba_out,
               double* t_out, int64_t* surf_out, void* stream);

/* EdgeRender._st_interact (renderers.py:94-116): edge = surface id differs from the left or
 * upper neighbour (outside = -1), grown by `rings` 3x3 binary dilations (upstream:
 * max(1, int(max(v, h) / 300))); rgba_out (v, h, 4): edges (0,0,0,1), the rest (1,1,1,0).
 * workspace: device scratch of prt_edge_workspace_bytes(h, v) bytes. */
int64_t prt_edge_workspace_bytes(int64_t h_pixels, int64_t v_pixels);
int prt_edge_canvas(int device, const int64_t* surf, int64_t h_pixels, int64_t v_pixels,
                    int rings, double* rgba_out, void* workspace, void* stream);

/* ---- tinygfx/g3d/operations.py as functions -------------
 * The vector helpers the primitives and materials are written with, callable on their own like
 * upstream's (`cg.reflect`, `cg.refract`, ...).  Column vectors: (rows, n) blocks with leading
 * dimension ld, rows in 1..4 (homogeneous 4-vectors upstream). */

/* reflect(vectors, normals) (operations.py:86-107): out = v - 2 n (v.n) */
int prt_reflect(int device, const double* vectors, const double* normals, int rows, int64_t n,
                int64_t ld, double* out, int64_t ld_out, void* stream);

/* refract(vectors, normals, n1, n2, n_global) (operations.py:110-162).  `vectors` is normalised
 * IN PLACE, as upstream does (:125); n1, n2: device (n) per-ray indices; out: refracted (or
 * totally reflected) unit directions; index_out (n): n2 (n_global when leaving) or n1 on TIR. */
int prt_refract(int device, double* vectors, const double* normals, const double* n1,
                const double* n2, double n_global, int rows, int64_t n, int64_t ld, double* out,
                int64_t ld_out, double* index_out, void* stream);

/* binomial_root(a, b, c) (operations.py:28-63) -> roots_out (2, n): the degenerate branches of
 * the reference included (|a| <= 1e-8 linear, then |b| <= 1e-8 -> +-inf by the sign of c). */
int prt_binomial_root(int device, const double* a, const double* b, const double* c, int64_t n,
                      double* roots_out, int64_t ld_out, void* stream);

/* smallest_positive_root(a, b, c) (operations.py:4-25) -> out (n), +inf where there is none */
int prt_smallest_positive_root(int device, const double* a, const double* b, const double* c,
                               int64_t n, double* out, void* stream);

/* element_wise_dot(m1, m2, axis) (operations.py:66-83) in strided form:
 * out[o] = sum_{r < reduce_len} m1[o*out_stride + r*reduce_stride] * m2[same], o < out_len.
 * axis 0 of a (k, n) block: reduce_len k, reduce_stride ld, out_len n, out_stride 1. */
int prt_dot(int device, const double* m1, const double* m2, int64_t reduce_len,
            int64_t reduce_stride, int64_t out_len, int64_t out_stride, double* out, void* stream);

/* csg.array_csg(array1, array2, operation, sort_output) (tinygfx/g3d/csg.py:13-61): the interval
 * algebra of a CSG node on two (m, n) blocks of ascending hit lists (m even: enter/exit pairs), one
 * column per ray; op = PRT_NODE_UNION / INTERSECT / DIFFERENCE.  out (m_left + m_right, n): with
 * sort_output the surviving entries ascending followed by +inf, otherwise the merged order with
 * rejected entries set to +inf.  Ties merge left-first (the stable order of SURVEY.md Q8). */
int prt_array_csg(int device, const double* left, int m_left, const double* right, int m_right,
                  int64_t n, int64_t ld, int op, int sort_output, double* out, int64_t ld_out,
                  void* stream);

/* primitive.intersect(rays) / primitive.normal(points) in OBJECT space (tinygfx/g3d/primitives.py:
 * Sphere :220-296, Paraboloid :299-419, Plane :422-498, Cube :501-602, Cylinder :621-741), i.e. the
 * shape routines without a world transform: type = PRT_PRIM_*, params as in prt_prim.
 *   rays (>= 7, n) ld (rows 0-2 origin, 4-6 direction; w rows are not read) -> hits_out (2, n):
 *   the raw pair in upstream's order -- not sorted, NaN where upstream yields NaN
 *   points (>= 3, n) ld -> normals_out (4, n): unit normal, w = 0 */
int prt_primitive_intersect(int device, int type, const double* params, const double* rays,
                            int64_t n, int64_t ld, double* hits_out, int64_t ld_out, void* stream);
int prt_primitive_normal(int device, int type, const double* params, const double* points,
                         int64_t n, int64_t ld, double* normals_out, int64_t ld_out, void* stream);

/* error codes */
#define PRT_OK 0
#define PRT_ERR_ARG (-1)
#define PRT_ERR_HIP (-2)
#define PRT_ERR_SCENE (-3)
#define PRT_ERR_ROWS_CAP (-4)
#define PRT_ERR_UNTRACABLE (-5) /* a ray hit a PRT_MAT_NONE surface (AttributeError upstream) */
#define PRT_ERR_WAVELENGTH (-6) /* a ray's wavelength is not in the index table of the PRT_MAT_TABLE glass it hit */

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */
