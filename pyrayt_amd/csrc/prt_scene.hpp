// prt_scene.hpp -- host side of libprt_hip: error plumbing, the scene object behind prt_scene*,
// the scene compiler (component trees -> linear step programs with LDS slot allocation, cull
// boxes, root rules) and the per-device upload.  Included by prt_kernels.hip (one translation
// unit: the kernels and the C-ABI follow there).
#pragma once
// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

static int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(PRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
static const int kMaxBatch = 64;  // generations launched between two host looks at the counts
struct SceneDev {
  const DevPrim* prims;
  const DevInstr* code;
  int n_instr;
  int lds_slots;
  unsigned long long absorber_mask = 0;  // bit p: primitive p (< 64) has the absorbing material
};

struct Program {
  std::vector<DevInstr> code;
  int lds_slots = 0;
  std::vector<int32_t> component_first;  // per component: first and one-past-last step slot (its own cull step included)
  int mirror_steps = 0;                  // trace programs stored in both directions: slots of [PICK][mirror image][JUMP] in front of the program proper
};

// One trace in flight (prt_trace_begin ... prt_trace_end).  A scene has PRT_TRACE_TICKETS of them per
// device, each with its own host mirror, events and record of the control words it left in its
// workspace, so that the host can enqueue the next trace while the GPU still runs the previous one.
static const int kMaxGenerationSlots = 1024;
struct TraceTicket {
  bool active = false;       // begun, not ended yet
  // the call's arguments
  const double* rays = nullptr;
  int64_t n = 0, ld = 0;
  int limit = 0;
  double ray_offset = 0;
  double* rows_out = nullptr;
  int64_t rows_cap = 0;
  char* w = nullptr;
  int flags = 0;
  hipStream_t st = nullptr;
  // the attempt in flight (fused path): generations [g, g + batch) are enqueued and publish `epoch`
  bool launched = false, allow_hints = true, use_hints = false, compact = true, publish_in_kernel = true;
  int g = 0, batch = 0, test_stall = 0, n_seen = 0;
  int64_t total_rows = 0;
  char seen_mode[kMaxGenerationSlots];
  char seen_sparse[kMaxGenerationSlots];         // per generation seen: bit 0 sparse loss, bit 1 dense enough to stall a look-back, bit 2 few enough for a dead list, bits 3 / 4 more than 1 ray in 32 / 16 not recorded
  char launch_mode[kMaxGenerationSlots] = {0};  // how each generation of the attempt was launched: assume value, 4 = dense with absorbed rays kept
  double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // include/prt.h prt_trace_stats
  // resources of the ticket
  int64_t* host_pinned = nullptr;               // 16 x int64 pinned staging (three-kernel path: control block, row count)
  struct HostMirror* mirror = nullptr;          // host-mapped: the device publishes a batch's counts here
  struct HostMirror* mirror_dev = nullptr;      // the same memory as the device addresses it
  unsigned long long epoch = 0;                 // number of batches published so far
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing_pending = false;                  // ev0..ev1 of the last batch not yet added to stats[2]
  hipEvent_t busy0 = nullptr, busy1 = nullptr;  // PRT_TRACE_BUSY: this trace's own pair (prt_trace_batch lends them)
  char* busy_recorded = nullptr;                // ... and where to note that the pair was recorded (a job that launches nothing leaves it 0)
  // control words already initialised on the stream for a next trace of this shape (see fused_start)
  const void* ready_workspace = nullptr;
  int64_t ready_n = -1;
  int ready_slots = 0, ready_stall = 0;
  int flip = 0;  // which of the two tile-status buffers the next generation launch works on
  unsigned long long user = 0;  // identity of this ticket in the workspace registry (0: not drawn yet)
  unsigned long long update_seen = 0;   // the scene update this ticket's stream has waited for ...
  hipStream_t update_seen_stream = nullptr;  // ... and which stream that was
  bool traced = false;                  // launched something since the scene's last update
  // record plan of this ticket (prt_trace_set_plan; prt_trace_kernels.hpp PlanDev): what the PLAN kernels are told
  bool plan_active = false;
  bool plan_dirty = false;            // the device copy is older than plan_host
  bool sink_unclean = false;          // the slotted sums hold what an attempt that was not folded left there
  struct PlanDev* plan_host = nullptr;  // (owned; the device copy sits in plan_dev)
  struct PlanDev* plan_dev = nullptr;
  double* sink_slots = nullptr;       // (slots, limit, n_groups, SINK_STATS): where the waves add
  size_t sink_slot_bytes = 0;
  double* sums_out = nullptr;         // the caller's (limit, n_groups, SINK_STATS) block
  unsigned long long plan_key = 0;    // identifies the plan the scene's plan hints were learnt under
};

struct DeviceCopy {
  // the scene tables of this device: one allocation (`block`), the pointers below point into it.  prt_scene_update
  // overwrites the whole image with ONE stream-ordered copy out of pinned staging memory (see there).
  char* block = nullptr;
  size_t block_bytes = 0;
  char* staging = nullptr;                      // pinned host image of `block`
  hipStream_t update_stream = nullptr;          // the copies of prt_scene_update run here ...
  hipEvent_t update_event = nullptr, tail_event = nullptr;  // ... behind everything the tickets enqueued; consumers wait for update_event
  unsigned long long update_serial = 0;         // updates so far (a ticket's stream waits once per update)
  bool update_unsettled = false;                // an update's copy may still be in flight (entry points other than the tickets' wait on the host)
  bool untracked_use = false;                   // something other than a ticket's trace used the tables since the last update
  DevPrim* prims = nullptr;
  double* tables = nullptr;                     // index tables of the PRT_MAT_TABLE materials: wavelengths | indices
  DevInstr* trace_code = nullptr;               // all components, each reduced to its candidate hit
  int32_t* trace_component_first = nullptr;     // [2 * components] step ranges of trace_code (k-lanes kernels)
  DevInstr* render_code = nullptr;              // same with the renderers' selection rule
  std::vector<DevInstr*> component_code;        // one program per component, no I_ROOT
  TraceTicket ticket[PRT_TRACE_TICKETS];
  std::vector<hipEvent_t> busy_events;          // PRT_TRACE_BUSY: two per job of the largest batch so far (kept, reused)
  std::vector<char> busy_recorded;              // ... per job of the current batch: its pair was recorded by a launch
  double busy[4] = {0, 0, 0, 0};                // prt_trace_batch_busy: what the last such batch measured
  bool ready = false;                           // everything above is in place (a copy whose upload failed half way is not)
};

struct prt_scene {
  prt_scene_options options;                    // as given to prt_scene_create / prt_scene_update (zeros = defaults)
  std::vector<prt_prim> prims;
  std::vector<prt_node> nodes;
  std::vector<int32_t> roots;
  std::vector<prt_material> mats;
  std::vector<DevPrim> dev_prims;
  Program trace_program;
  Program render_program;
  std::vector<Program> component_programs;
  std::vector<Operand> component_result;        // where each component program leaves its list
  std::vector<DeviceCopy> per_device;
  // dense-mode hints of traces under a record plan (their own: what a plan stores changes which generations are dense)
  std::vector<char> plan_hint_mode;
  unsigned long long plan_hint_key = 0;
  int plan_hint_rest = 0, plan_hint_misses_in_a_row = 0;
  long long plan_launches = 0, plan_dense_launches = 0, plan_misses = 0, plan_traces = 0;
  bool has_untracable = false;                  // a shading error can be raised at store time (a surface without a
                                                // traceable material, a caller-shaded one, a table glass that may miss
                                                // a wavelength): such traces publish behind the batch (k_fused_reinit)
  // prt_scene_set_index_tables: per material (first, count) into the two arrays below
  std::vector<int64_t> table_ranges;
  std::vector<double> table_wavelengths, table_indices;
  bool has_cull_steps = false;                  // the trace program carries I_BOX steps (k_generation<CULL>)
  bool spatial_groups = false;                  // ... grouped by position: components are visited out of list order
  int stats_device = -1, stats_ticket = 0;      // whose statistics prt_trace_stats reports: the trace ended last
  int last_generations = 0;  // working generations of the previous trace: sizes the first batch
  long lookback_fallbacks = 0;  // traces of this scene that fell back to the three-kernel path
  // what the previous trace's generations looked like (dense-mode hints for the next one, see k_generation)
  int64_t hint_n = -1;          // ray count of the trace the hints were recorded from (-1: no hints)
  int hint_keep_absorbed = 0;
  std::vector<char> hint_mode;  // per generation: 0 general, 1 every ray recorded and carried on, 2 recorded, none carried,
                                // 4 recorded, all but a few (absorbed) carried on: runs as 1 with the absorbed rays kept
  long speculation_misses = 0;  // traces that had to be repeated because a hint did not hold
  bool full_rows = false;       // a ray set of this scene needed the state rows the compact form leaves out
  long full_rows_fallbacks = 0; // traces that had to be repeated for that reason (at most one per scene)
  long dense_launches = 0;      // generation launches made in dense mode so far
  long sparse_keep_launches = 0;  // ... of which kept their absorbed rays to stay dense (hint mode 4)
  // PRT_TRACE_COUNT_PATHS: traces counted, rays that were not well formed, CSG node evaluations under an
  // implied cull box that had survivors, ... of which took upstream's exact box test
  long long path_counts[4] = {0, 0, 0, 0};
  int hint_holdoff = 0;         // traces still to run without hints after a miss (doubles with every miss in a row)
  std::vector<char> missed_mode;            // the hint modes of an attempt that missed, until its repeat has been looked at
  std::vector<int> hint_rest, hint_rest_span;  // per generation: traces for which its dense hint is not offered / the span of its last rest
  int hint_misses_in_a_row = 0;
};

static int leaves_under(const prt_scene* s, int node) {
  const prt_node& n = s->nodes[node];
  if (n.op == PRT_NODE_LEAF) return 1;
  return leaves_under(s, n.left) + leaves_under(s, n.right);
}

// Linearise one component (post-order) and place its hit lists in LDS.
//
// Slots are handed out downwards from a ceiling (slot numbers are negative while compiling and
// shifted afterwards).  A leaf that feeds the CSG step right after it stays in registers
// (REGA / REGB); every CSG result goes to LDS.  With a register operand of length 2 and the
// other operand at [b, b+m) the result is written to [b-2, b+m): the merge writes slot
// b-2+k only after it has consumed k-2 entries of that operand, so it never overwrites an
// unread entry, and a chain of k leaves needs just 2k slots.  Two LDS operands keep the right
// one a further len(right) slots down so the same argument holds for the left one.
static bool solid_bounds(const prt_scene* s, int node, double* box);

static double short_direction_bound(const prt_scene* s, int node);

struct Compiler {
  const prt_scene* s;
  std::vector<DevInstr>& out;
  int lowest = 0;
  bool positive_only = false;  // trace programs: only positive hits are ever looked at

  static DevInstr blank(int kind) {
    DevInstr in;
    std::memset(&in, 0, sizeof(in));
    in.kind = kind;
    return in;
  }

  Operand leaf(int node, int mode, int ceiling) {
    DevInstr in = blank(I_LEAF);
    in.a0 = s->nodes[node].prim;
    in.a1 = mode;
    const prt_prim& pr = s->prims[in.a0];
    in.type = pr.type;
    for (int k = 0; k < 6; ++k) in.data[k] = pr.params[k];
    for (int k = 0; k < 12; ++k) in.data[6 + k] = pr.minv[k];  // rows 0..2 of M^-1
    Operand o = {mode, 0, 2};
    if (mode == OPER_LDS) {
      o.base = ceiling - 2;
      in.a2 = o.base;
      lowest = std::min(lowest, o.base);
    }
    out.push_back(in);
    return o;
  }

  // does the upstream cull box of `node` provably contain the node's solid?  (see csg_keep)
  // Structurally it does for INTERSECT / DIFFERENCE over UNION-free subtrees -- *if* the box is the one
  // csg.py:93-116 would compute now.  Upstream caches it, and the cache of the outer node of a
  // right-nested tree moved after construction is stale (world_objects.py:315-317): the snapshot
  // carries that stale box, so the box is also checked against the solid's real bounds (found by the
  // 500-seed fuzz tier: a stale, empty box culls every ray upstream).
  bool box_contains_solid(int node) const {
    const prt_node& n = s->nodes[node];
    if (s->options.no_implied) return false;  // test knob: every node tests its cull box exactly
    if (n.op == PRT_NODE_LEAF) return true;
    // the argument needs every leaf to see |d_obj|^2 >= 1e-3 for every well-formed ray, i.e. for world
    // directions with |d|^2 >= kWellFormedLen2Lo (csg_keep sends every other ray to the exact test)
    if (short_direction_bound(s, node) > kWellFormedLen2Lo) return false;
    bool structural = false;
    if (n.op == PRT_NODE_INTERSECT) structural = box_contains_solid(n.left) && box_contains_solid(n.right);
    if (n.op == PRT_NODE_DIFFERENCE) structural = box_contains_solid(n.left);
    if (!structural) return false;  // UNION of disjoint operands keeps only the first operand's span (csg.py:98-109)
    double real[6];
    if (!solid_bounds(s, node, real)) return false;
    for (int k = 0; k < 3; ++k) {
      if (!(real[2 * k] <= real[2 * k + 1])) continue;  // the solid has no extent on this axis: nothing to contain
      const double tol = 1e-12 * (1.0 + std::fabs(real[2 * k]) + std::fabs(real[2 * k + 1]));
      if (!(n.aabb[2 * k] <= real[2 * k] + tol && n.aabb[2 * k + 1] >= real[2 * k + 1] - tol)) return false;
    }
    return true;
  }

  Operand emit(int node, int ceiling) {
    const prt_node& n = s->nodes[node];
    if (n.op == PRT_NODE_LEAF) return leaf(node, OPER_LDS, ceiling);
    const bool l_leaf = s->nodes[n.left].op == PRT_NODE_LEAF;
    const bool r_leaf = s->nodes[n.right].op == PRT_NODE_LEAF;
    Operand L, R;
    int base;
    if (l_leaf && r_leaf) {
      L = leaf(n.left, OPER_REGA, 0);
      R = leaf(n.right, OPER_REGB, 0);
      base = ceiling - 4;
    } else if (r_leaf) {
      L = emit(n.left, ceiling);
      R = leaf(n.right, OPER_REGB, 0);
      base = L.base - 2;
    } else if (l_leaf) {
      R = emit(n.right, ceiling);  // evaluation order of pure children is free
      L = leaf(n.left, OPER_REGA, 0);
      base = R.base - 2;
    } else {
      L = emit(n.left, ceiling);
      const int m_r = 2 * leaves_under(s, n.right);
      R = emit(n.right, L.base - m_r);
      base = L.base - m_r;
    }
    // INTERSECT / DIFFERENCE results lie inside the left solid: when the left list has no positive
    // entry the node cannot contribute a positive hit whatever the right operand is (entries that
    // are not positive come in enter/exit pairs and leave the depth seen by positive entries
    // unchanged), so a right leaf evaluated just before the node is marked skippable.
    if (positive_only && r_leaf && !out.empty() && out.back().kind == I_LEAF && out.back().a1 == OPER_REGB &&
        (n.op == PRT_NODE_INTERSECT || n.op == PRT_NODE_DIFFERENCE)) {
      DevInstr& rl = out.back();
      rl.pad[0] = 1;
      rl.a3 = L.mode; rl.a4 = L.base; rl.a5 = L.len;
    }
    DevInstr in = blank(I_CSG);
    in.pad[1] = box_contains_solid(node) ? 1 : 0;
    in.a0 = n.op;
    in.a1 = L.mode; in.a2 = L.base; in.a3 = L.len;
    in.a4 = R.mode; in.a5 = R.base; in.a6 = R.len;
    in.a7 = base;
    for (int k = 0; k < 6; ++k) in.data[k] = n.aabb[k];
    out.push_back(in);
    lowest = std::min(lowest, base);
    return Operand{OPER_LDS, base, L.len + R.len};
  }
};

static void shift_slots(std::vector<DevInstr>& code, size_t from, int shift) {
  for (size_t k = from; k < code.size(); ++k) {
    DevInstr& in = code[k];
    if (in.kind == I_CHAIN) { k += CHAIN_SLOTS - 1; continue; }  // no lists; the next slots are raw data
    if (in.kind == I_LEAF && in.a1 == OPER_LDS) in.a2 += shift;
    if (in.kind == I_LEAF && in.pad[0] == 1 && in.a3 == OPER_LDS) in.a4 += shift;
    if (in.kind == I_CSG) {
      if (in.a1 == OPER_LDS) in.a2 += shift;
      if (in.a4 == OPER_LDS) in.a5 += shift;
      in.a7 += shift;
    }
    if (in.kind == I_ROOT && in.a0 == OPER_LDS) in.a1 += shift;
  }
}

// compile component `root_node`; returns where its result list ends up
// ---- world-space bounds of a component's leaf surfaces (for the I_BOX cull step) ---------------
static bool invert4(const double* m, double* out) {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
    if (!(std::fabs(a[piv][col]) > 1e-300)) return false;
    for (int c = 0; c < 8; ++c) std::swap(a[col][c], a[piv][c]);
    const double inv = 1.0 / a[col][col];
    for (int c = 0; c < 8; ++c) a[col][c] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      const double f = a[r][col];
      for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
    }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = a[r][4 + c];
  return true;
}

// object-space bounds of a primitive: the extents its intersect routine clips to
static bool prim_bounds(const prt_prim& p, double lo[3], double hi[3]) {
  const double* q = p.params;
  switch (p.type) {
    case PRT_PRIM_SPHERE: { const double r = std::fabs(q[0]); for (int k = 0; k < 3; ++k) { lo[k] = -r; hi[k] = r; } } return true;
    case PRT_PRIM_CYLINDER: { const double r = std::fabs(q[0]); lo[0] = lo[1] = -r; hi[0] = hi[1] = r;
      lo[2] = std::min(q[1], q[2]); hi[2] = std::max(q[1], q[2]); } return true;
    case PRT_PRIM_PLANE: lo[0] = -std::fabs(q[0]) / 2; hi[0] = std::fabs(q[0]) / 2; lo[1] = -std::fabs(q[1]) / 2;
      hi[1] = std::fabs(q[1]) / 2; lo[2] = hi[2] = 0.0; return true;
    case PRT_PRIM_CUBE: for (int k = 0; k < 3; ++k) { lo[k] = std::min(q[2 * k], q[2 * k + 1]); hi[k] = std::max(q[2 * k], q[2 * k + 1]); } return true;
    case PRT_PRIM_PARABOLOID: { if (!(q[0] > 0) || !(q[1] > 0)) return false; const double rim = std::sqrt(4 * q[0] * q[1]);
      lo[0] = lo[1] = -rim; hi[0] = hi[1] = rim; lo[2] = 0.0; hi[2] = q[1]; } return true;
  }
  return false;
}

// World-space box (xmin,xmax,ymin,ymax,zmin,zmax) of the solid of `node`: a leaf's is the box of
// its transformed object-space corners; A&B lies in both operands' boxes, A-B in A's, A|B in their
// union.  Every entry a node's hit list keeps is a parameter at which the ray crosses the boundary
// of that solid (array_csg keeps exactly the depth changes into and out of "inside"), hence a point
// of this box -- which, unlike upstream's own cull box (csg.py:93-116), is only ever used to skip
// work that cannot matter, never to change a result.
static bool solid_bounds(const prt_scene* s, int node, double* box) {
  const prt_node& n = s->nodes[node];
  if (n.op != PRT_NODE_LEAF) {
    double l[6], r[6];
    if (!solid_bounds(s, n.left, l)) return false;
    if (n.op == PRT_NODE_DIFFERENCE) { std::memcpy(box, l, sizeof(l)); return true; }
    if (!solid_bounds(s, n.right, r)) return false;
    for (int k = 0; k < 3; ++k) {
      const bool both = n.op == PRT_NODE_INTERSECT;
      box[2 * k] = both ? std::max(l[2 * k], r[2 * k]) : std::min(l[2 * k], r[2 * k]);
      box[2 * k + 1] = both ? std::min(l[2 * k + 1], r[2 * k + 1]) : std::max(l[2 * k + 1], r[2 * k + 1]);
    }
    return true;
  }
  const prt_prim& p = s->prims[n.prim];
  double lo[3], hi[3], world[16];
  if (!prim_bounds(p, lo, hi) || !invert4(p.minv, world)) return false;
  for (int k = 0; k < 3; ++k) { box[2 * k] = HUGE_VAL; box[2 * k + 1] = -HUGE_VAL; }
  for (int corner = 0; corner < 8; ++corner) {
    const double v[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
    const double w = world[12] * v[0] + world[13] * v[1] + world[14] * v[2] + world[15];
    for (int k = 0; k < 3; ++k) {
      const double x = (world[4 * k] * v[0] + world[4 * k + 1] * v[1] + world[4 * k + 2] * v[2] + world[4 * k + 3]) / w;
      if (!std::isfinite(x)) return false;
      box[2 * k] = std::min(box[2 * k], x);
      box[2 * k + 1] = std::max(box[2 * k + 1], x);
    }
  }
  return true;
}

enum { ROOT_NONE = 0, ROOT_TRACE = 1, ROOT_RENDER = 2 };

// widen a cull box by 1e-3 of its diagonal (plus a floor relative to its distance from the origin)
static void pad_box(double* box) {
  double diag = 0, reach = 0;
  for (int k = 0; k < 3; ++k) {
    const double side = std::max(0.0, box[2 * k + 1] - box[2 * k]);  // an empty overlap has no extent
    diag += side * side;
    reach = std::max(reach, std::max(std::fabs(box[2 * k]), std::fabs(box[2 * k + 1])));
  }
  const double pad = 1e-3 * std::sqrt(diag) + 1e-9 * reach + 1e-12;
  for (int k = 0; k < 3; ++k) { box[2 * k] -= pad; box[2 * k + 1] += pad; }
}

// The cull steps argue geometrically: a finite positive entry of a component's hit list is a parameter at
// which the ray is on one of the component's surfaces, hence inside its box.  That holds as long as no
// leaf takes one of upstream's degenerate branches *spuriously* -- `isclose(a, 0)` / `isclose(d_z, 0)` with
// their absolute 1e-8 thresholds on object-space quantities (primitives.py:346, 531, 683) fire for any ray
// whose object-space direction is short, not only for one that really is parallel, and what the
// branch then reports (e.g. the paraboloid's -c / b) is not a point of the surface.  So a component
// has a bound on the squared world-space direction length below which that can happen for one of its
// leaves -- 1e-3 over the smallest squared singular value of the leaf's M^-1, i.e. |d_obj|^2 >= 1e-3:
// a branch that fires there (|d_obj,xy|^2 <= 1e-8) means sin^2 of the angle to the axis <= 1e-5, and
// the point its linear root names is off the surface by about that times the squared distance
// travelled, far inside the 1e-3-of-the-diagonal padding for anything an optical bench holds.  The
// kernels have ONE gate for all their shortcuts (well_formed(): w = 1 / 0 and |d|^2 >= kWellFormedLen2Lo,
// prt_device.hpp), so the rule is applied here, at compile time: a component gets cull steps only if its
// bound is at most kWellFormedLen2Lo, and may_reach lets every ray that is not well formed through.  For
// objects of ordinary size the bound is far below that; an object scaled up by more than ~30 exceeds it
// and is simply never culled.  (Found by the adv_still fixture.)
static double min_singular_sq(const double* minv) {
  // smallest eigenvalue of B = A^T A, A = the 3x3 linear part of the row-major 4x4 M^-1 (analytic
  // symmetric 3x3 eigenvalues; the result is halved to stay below rounding)
  double a[3][3], b[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[r][c] = minv[4 * r + c];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) b[i][j] = a[0][i] * a[0][j] + a[1][i] * a[1][j] + a[2][i] * a[2][j];
  const double p1 = b[0][1] * b[0][1] + b[0][2] * b[0][2] + b[1][2] * b[1][2];
  const double q = (b[0][0] + b[1][1] + b[2][2]) / 3.0;
  if (p1 == 0.0) return 0.5 * std::min(b[0][0], std::min(b[1][1], b[2][2]));
  const double p2 = (b[0][0] - q) * (b[0][0] - q) + (b[1][1] - q) * (b[1][1] - q) + (b[2][2] - q) * (b[2][2] - q) + 2 * p1;
  const double p = std::sqrt(p2 / 6.0);
  double c[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i][j] = (b[i][j] - (i == j ? q : 0.0)) / p;
  double det = c[0][0] * (c[1][1] * c[2][2] - c[1][2] * c[2][1]) - c[0][1] * (c[1][0] * c[2][2] - c[1][2] * c[2][0]) +
               c[0][2] * (c[1][0] * c[2][1] - c[1][1] * c[2][0]);
  const double rr = std::max(-1.0, std::min(1.0, det / 2.0));
  const double phi = std::acos(rr) / 3.0;
  const double smallest = q + 2 * p * std::cos(phi + 2.0943951023931953);  // + 2 pi / 3: the smallest root
  return 0.5 * std::max(0.0, smallest);
}
static double short_direction_bound(const prt_scene* s, int node) {
  const prt_node& n = s->nodes[node];
  if (n.op != PRT_NODE_LEAF) return std::max(short_direction_bound(s, n.left), short_direction_bound(s, n.right));
  const double sigma2 = min_singular_sq(s->prims[n.prim].minv);
  return sigma2 > 0 ? 1e-3 / sigma2 : HUGE_VAL;
}

// count the steps of one kind in a program (a chain record spans CHAIN_SLOTS raw slots)
static int count_steps(const std::vector<DevInstr>& code, int kind) {
  int n = 0;
  for (size_t k = 0; k < code.size(); ++k) {
    n += (code[k].kind == kind && !(kind == I_BOX && code[k].a1 != BOX_TEST)) ? 1 : 0;  // (the PICK / JUMP frame is not a cull step)
    if (code[k].kind == I_CHAIN) k += CHAIN_SLOTS - 1;
  }
  return n;
}

// Is component `root_node` a left-deep chain of two or three leaves whose primitive types have a
// compiled chain body?  Fills the record if so.
static bool chain_record(const prt_scene* s, int root_node, const Compiler& c, DevChain* out) {
  if (s->options.no_chain) return false;  // A/B and test knob: the step interpreter instead
  const prt_node& top = s->nodes[root_node];
  if (top.op == PRT_NODE_LEAF || s->nodes[top.right].op != PRT_NODE_LEAF) return false;
  int leaves[3], n_leaves, nodes[2];
  const prt_node& l = s->nodes[top.left];
  if (l.op == PRT_NODE_LEAF) {
    n_leaves = 2;
    leaves[0] = top.left; leaves[1] = top.right;
    nodes[0] = root_node; nodes[1] = -1;
  } else {
    if (s->nodes[l.left].op != PRT_NODE_LEAF || s->nodes[l.right].op != PRT_NODE_LEAF) return false;
    n_leaves = 3;
    leaves[0] = l.left; leaves[1] = l.right; leaves[2] = top.right;
    nodes[0] = top.left; nodes[1] = root_node;
  }
  int types[3] = {-1, -1, -1};
  for (int k = 0; k < n_leaves; ++k) types[k] = s->prims[s->nodes[leaves[k]].prim].type;
  static const struct { int shape, t0, t1, t2; } table[] = {
      {CHAIN_SSC, PRIM_SPHERE, PRIM_SPHERE, PRIM_CYLINDER}, {CHAIN_SSQ, PRIM_SPHERE, PRIM_SPHERE, PRIM_CUBE},
      {CHAIN_CSS, PRIM_CYLINDER, PRIM_SPHERE, PRIM_SPHERE}, {CHAIN_QSS, PRIM_CUBE, PRIM_SPHERE, PRIM_SPHERE},
      {CHAIN_QQQ, PRIM_CUBE, PRIM_CUBE, PRIM_CUBE},
      {CHAIN_SC, PRIM_SPHERE, PRIM_CYLINDER, -1}, {CHAIN_SQ, PRIM_SPHERE, PRIM_CUBE, -1},
      {CHAIN_CS, PRIM_CYLINDER, PRIM_SPHERE, -1}, {CHAIN_QS, PRIM_CUBE, PRIM_SPHERE, -1},
      {CHAIN_CB, PRIM_CYLINDER, PRIM_PARABOLOID, -1}, {CHAIN_QB, PRIM_CUBE, PRIM_PARABOLOID, -1},
      {CHAIN_PC, PRIM_PLANE, PRIM_CYLINDER, -1}, {CHAIN_PQ, PRIM_PLANE, PRIM_CUBE, -1},
      {CHAIN_BC, PRIM_PARABOLOID, PRIM_CYLINDER, -1}};
  int shape = -1;
  for (const auto& row : table)
    if (row.t0 == types[0] && row.t1 == types[1] && row.t2 == types[2]) shape = row.shape;
  if (shape < 0) return false;
  std::memset(out, 0, sizeof(*out));
  out->kind = I_CHAIN;
  out->shape = shape;
  out->n_leaves = n_leaves;
  out->op1 = s->nodes[nodes[0]].op;
  out->implied1 = c.box_contains_solid(nodes[0]) ? 1 : 0;
  for (int k = 0; k < 6; ++k) out->box1[k] = s->nodes[nodes[0]].aabb[k];
  if (n_leaves == 3) {
    out->op2 = s->nodes[nodes[1]].op;
    out->implied2 = c.box_contains_solid(nodes[1]) ? 1 : 0;
    for (int k = 0; k < 6; ++k) out->box2[k] = s->nodes[nodes[1]].aabb[k];
  }
  for (int k = 0; k < n_leaves; ++k) {
    const int prim = s->nodes[leaves[k]].prim;
    const prt_prim& pr = s->prims[prim];
    out->prim[k] = prim;
    for (int j = 0; j < 6; ++j) out->leaf[k][j] = pr.params[j];
    for (int j = 0; j < 12; ++j) out->leaf[k][6 + j] = pr.minv[j];  // rows 0..2 of M^-1
  }
  // a cylinder as third leaf may be cleared without being evaluated (chain_candidate, chord_inside_cylinder)
  out->clearance = s->options.no_clearance ? 0 : 1;
  // every node an INTERSECT whose cull box is implied: the chain is an intersection of intervals (chain_candidate)
  out->intervals = (out->op1 == PRT_NODE_INTERSECT && out->implied1 && (n_leaves == 2 || (out->op2 == PRT_NODE_INTERSECT && out->implied2))) ? 1 : 0;
  if (s->options.no_intervals) out->intervals = 0;
  return true;
}
static Operand compile_component(const prt_scene* s, int root_node, std::vector<DevInstr>& code,
                                 int root_rule, int* slots) {
  // a scene of several components gets a cull step in front of each: most rays can reach only one
  // or two of them (see may_reach); with one or two components the test would cost more than it saves
  size_t box_at = (size_t)-1;
  const bool cull_off = s->options.no_cull != 0;  // A/B and test knobs (prt_scene_options)
  const int cull_min = s->options.cull_min > 0 ? s->options.cull_min : 3;
  // render programs: every component behind a LINE test (the pictures' rays are coherent and most of a picture
  // is background or another part: a wave whose lines of sight all miss a part's box skips the part)
  const bool line_cull = root_rule == ROOT_RENDER && !cull_off;
  if ((root_rule == ROOT_TRACE && (int)s->roots.size() >= cull_min && !cull_off) || line_cull) {
    DevInstr in = Compiler::blank(I_BOX);
    in.a1 = line_cull ? BOX_LINE : BOX_TEST;
    double* box = in.data;
    // (a cull step argues about well-formed rays only; a component so large in object space that even
    // a unit direction may be "short" to one of its leaves -- short_direction_bound -- gets none)
    if (solid_bounds(s, root_node, box) && short_direction_bound(s, root_node) <= kWellFormedLen2Lo) {
      pad_box(box);
      box_at = code.size();
      code.push_back(in);
    }
  }
  const size_t from = code.size();
  Compiler c{s, code};
  c.positive_only = root_rule == ROOT_TRACE;
  Operand res;
  DevChain chain;
  if (root_rule == ROOT_TRACE && chain_record(s, root_node, c, &chain)) {
    // the whole component is one register-only step: no lists, no LDS slots
    DevInstr raw[CHAIN_SLOTS];
    std::memcpy(raw, &chain, sizeof(chain));
    for (const DevInstr& slot : raw) code.push_back(slot);
    if (box_at != (size_t)-1) code[box_at].a0 = CHAIN_SLOTS;
    return Operand{OPER_REGA, 0, 0};
  }
  if (s->nodes[root_node].op == PRT_NODE_LEAF)
    res = c.leaf(root_node, OPER_REGA, 0);
  else
    res = c.emit(root_node, 0);
  if (root_rule != ROOT_NONE && s->nodes[root_node].op == PRT_NODE_LEAF) {
    code.back().pad[0] = 2;  // a bare surface: the leaf step itself yields the component's candidate
  } else if (root_rule == ROOT_TRACE && code.size() > from && code.back().kind == I_CSG) {
    code.back().pad[0] = 1;  // the node reduces straight to its nearest positive survivor
  } else if (root_rule != ROOT_NONE) {
    // the renderers may pick a non-positive entry, so their lists are always materialised
    DevInstr in = Compiler::blank(I_ROOT);
    in.a0 = res.mode; in.a1 = res.base; in.a2 = res.len;
    in.pad[0] = root_rule == ROOT_RENDER ? 2 : 0;
    code.push_back(in);
  }
  if (box_at != (size_t)-1) code[box_at].a0 = (int)(code.size() - from);  // steps to jump over
  const int shift = -c.lowest;
  shift_slots(code, from, shift);
  if (res.mode == OPER_LDS) res.base += shift;
  *slots = std::max(*slots, shift);
  return res;
}

// The trace program: every component's steps, and for scenes of many components a hierarchy of cull steps
// over GROUPS of components: a group of more than kGroupFanout components is split into kGroupFanout
// sub-groups, each led by an I_BOX step holding the union of its members' solid boxes whose jump spans
// the whole sub-group (the interpreter's I_BOX semantic nests as it is).  A coherent wave then tests
// O(fanout * log n) boxes instead of n: the groups behind it, or beyond its nearest hit so far, go with
// one test each.
// Two ways to form the groups.  LIST ORDER: runs of consecutive components -- tight when the parts are
// listed the way an optical train is written down, along the axis.  BY POSITION: the components are split
// at the median of their box centres along the longest axis of those centres, twice per level (four
// sub-groups), so the boxes are tight whatever order the caller listed the parts in; the program then
// visits components out of list order, which the tie rule of pyrayt/_pyrayt.py:380-386 (the first
// component in list order among those with the smallest t) survives because the running minimum is then
// kept lexicographically on (t, primitive index) -- primitives are numbered component by component, so
// that IS (t, list index) (Ray8::lex, `beats`).  The spatial form is taken when its boxes are clearly
// smaller (summed half surface area below 0.9 of the list-order form's); prt_scene_options.list_order_groups
// keeps the list-order form.
static const int kGroupFanout = 4;
static const int kGroupMinComponents = 8;

struct Group {
  std::vector<int> comps;    // list indices of the components under this node, ascending
  std::vector<Group> kids;   // empty: a single component
  bool boxed = false;        // led by its own cull step (a sub-group of several components)
};
struct ComponentBox {
  bool ok = false;           // solid_bounds() succeeded and the cull argument applies (short_direction_bound)
  double box[6] = {0, 0, 0, 0, 0, 0};
};

static bool union_box(const std::vector<ComponentBox>& boxes, const std::vector<int>& comps, double* out) {
  for (int k = 0; k < 3; ++k) { out[2 * k] = HUGE_VAL; out[2 * k + 1] = -HUGE_VAL; }
  for (int c : comps) {
    if (!boxes[c].ok) return false;
    for (int k = 0; k < 3; ++k) {
      out[2 * k] = std::min(out[2 * k], boxes[c].box[2 * k]);
      out[2 * k + 1] = std::max(out[2 * k + 1], boxes[c].box[2 * k + 1]);
    }
  }
  return !comps.empty();
}

// split `comps` into up to kGroupFanout parts: consecutive runs, or by position (median splits)
static std::vector<std::vector<int>> split_group(const std::vector<ComponentBox>& boxes, const std::vector<int>& comps,
                                                 bool spatial) {
  std::vector<std::vector<int>> parts;
  const int n = (int)comps.size();
  if (!spatial) {
    const int step = (n + kGroupFanout - 1) / kGroupFanout;
    for (int at = 0; at < n; at += step) parts.emplace_back(comps.begin() + at, comps.begin() + std::min(n, at + step));
    return parts;
  }
  auto halve = [&](const std::vector<int>& in, std::vector<int>& lo, std::vector<int>& hi) {
    double cmin[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, cmax[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (int c : in)
      for (int k = 0; k < 3; ++k) {
        const double mid = 0.5 * (boxes[c].box[2 * k] + boxes[c].box[2 * k + 1]);
        cmin[k] = std::min(cmin[k], mid);
        cmax[k] = std::max(cmax[k], mid);
      }
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (cmax[k] - cmin[k] > cmax[axis] - cmin[axis]) axis = k;
    std::vector<int> sorted = in;
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) {
      return boxes[a].box[2 * axis] + boxes[a].box[2 * axis + 1] < boxes[b].box[2 * axis] + boxes[b].box[2 * axis + 1];
    });
    const size_t half = (sorted.size() + 1) / 2;
    lo.assign(sorted.begin(), sorted.begin() + half);
    hi.assign(sorted.begin() + half, sorted.end());
  };
  std::vector<int> lo, hi, q[4];
  halve(comps, lo, hi);
  if (lo.size() > 1) halve(lo, q[0], q[1]); else q[0] = lo;
  if (hi.size() > 1) halve(hi, q[2], q[3]); else q[2] = hi;
  for (auto& part : q)
    if (!part.empty()) {
      std::sort(part.begin(), part.end());  // inside a group: list order
      parts.push_back(part);
    }
  return parts;
}

static Group build_groups(const std::vector<ComponentBox>& boxes, const std::vector<int>& comps, bool grouped,
                          bool spatial) {
  Group g;
  g.comps = comps;
  if (comps.size() <= 1) return g;
  std::vector<std::vector<int>> parts;
  if (grouped && (int)comps.size() > kGroupFanout) {
    parts = split_group(boxes, comps, spatial);
  } else {
    for (int c : comps) parts.push_back({c});
  }
  for (const auto& part : parts) {
    Group kid = build_groups(boxes, part, grouped, spatial);
    double unused[6];
    kid.boxed = part.size() > 1 && union_box(boxes, part, unused);
    g.kids.push_back(std::move(kid));
  }
  return g;
}

// summed half surface area of the group boxes: what a ray has to get past, per level
static double group_cost(const std::vector<ComponentBox>& boxes, const Group& g) {
  double cost = 0.0;
  for (const Group& kid : g.kids) {
    double b[6];
    if (kid.boxed && union_box(boxes, kid.comps, b)) {
      const double dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
      cost += dx * dy + dy * dz + dz * dx;
    }
    cost += group_cost(boxes, kid);
  }
  return cost;
}

// (root_rule ROOT_TRACE: into the trace program; ROOT_RENDER: into the render program, whose group steps test
// the LINE of the ray like its component steps, see compile_component)
static void emit_groups(prt_scene* s, const std::vector<ComponentBox>& boxes, const Group& g, bool mirrored,
                        int root_rule = ROOT_TRACE) {
  const bool tracing = root_rule == ROOT_TRACE;
  Program& prog = tracing ? s->trace_program : s->render_program;
  if (g.kids.empty()) {
    for (int c : g.comps) {  // (one component)
      if (!mirrored && tracing) prog.component_first[2 * c] = (int32_t)prog.code.size();
      compile_component(s, s->roots[c], prog.code, root_rule, &prog.lds_slots);
      if (!mirrored && tracing) prog.component_first[2 * c + 1] = (int32_t)prog.code.size();
    }
    return;
  }
  for (size_t at = 0; at < g.kids.size(); ++at) {
    const Group& kid = g.kids[mirrored ? g.kids.size() - 1 - at : at];
    size_t box_at = (size_t)-1;
    if (kid.boxed) {  // a sub-group of several components: its own cull step in front
      DevInstr in = Compiler::blank(I_BOX);
      in.a1 = tracing ? BOX_TEST : BOX_LINE;
      if (union_box(boxes, kid.comps, in.data)) {
        pad_box(in.data);
        box_at = prog.code.size();
        prog.code.push_back(in);
      }
    }
    const size_t from = prog.code.size();
    emit_groups(s, boxes, kid, mirrored, root_rule);
    if (box_at != (size_t)-1) prog.code[box_at].a0 = (int)(prog.code.size() - from);
  }
}

static void first_and_last_prim(const prt_scene* s, int node, int* lo, int* hi) {
  const prt_node& n = s->nodes[node];
  if (n.op == PRT_NODE_LEAF) { *lo = std::min(*lo, n.prim); *hi = std::max(*hi, n.prim); return; }
  first_and_last_prim(s, n.left, lo, hi);
  first_and_last_prim(s, n.right, lo, hi);
}

static void compile_trace_program(prt_scene* s) {
  Program& prog = s->trace_program;
  const int n = (int)s->roots.size();
  prog.component_first.assign((size_t)2 * n, 0);
  s->spatial_groups = false;
  if (n == 0) return;
  const bool cull_off = s->options.no_cull != 0 || s->options.no_groups != 0;
  const bool grouped = n >= kGroupMinComponents && !cull_off;
  std::vector<ComponentBox> boxes(n);
  std::vector<int> all(n);
  bool every_box = true, prims_follow_components = true;
  int last_prim = -1;
  for (int c = 0; c < n; ++c) {
    all[c] = c;
    boxes[c].ok = solid_bounds(s, s->roots[c], boxes[c].box) && short_direction_bound(s, s->roots[c]) <= kWellFormedLen2Lo;
    every_box = every_box && boxes[c].ok;
    int lo = INT32_MAX, hi = -1;
    first_and_last_prim(s, s->roots[c], &lo, &hi);
    prims_follow_components = prims_follow_components && lo > last_prim;  // (t, primitive index) == (t, list index)
    last_prim = std::max(last_prim, hi);
  }
  Group tree = build_groups(boxes, all, grouped, false);
  if (grouped && every_box && prims_follow_components && !s->options.list_order_groups) {
    Group by_position = build_groups(boxes, all, grouped, true);
    if (group_cost(boxes, by_position) < 0.9 * group_cost(boxes, tree)) {
      tree = std::move(by_position);
      s->spatial_groups = true;
    }
  }
  // The order in which a program visits the groups decides how much the cull steps save: a ray that meets
  // its nearest part FIRST culls everything beyond it (may_reach looks at (0, best_t]), one that meets it
  // last has tested every box on its way with best_t = inf.  A beam running down a lens train against the
  // order of the program -- reflected by a mirror at the far end, or simply launched from the other side --
  // pays for all 32 lenses (measured: 987 against 111 us per 1M rays).  So a grouped program is stored a
  // second time in mirror image (groups and components in the opposite order at every level), and a wave
  // takes that copy when most of its rays run against the program's axis: the coordinate axis along which
  // the centres of the first and the last top-level group lie furthest apart.  Any order gives the same
  // result: the running minimum of such programs is lexicographic on (t, list index), see `beats`.
  // In the program this is two more steps of the I_BOX kind (the interpreter's "jump if no lane wants what
  // follows"), PICK and JUMP, see the layout below.
  int pick_axis = -1;
  if (grouped && every_box && prims_follow_components && tree.kids.size() > 1 && !s->options.one_direction) {
    double first[6], last[6];
    if (union_box(boxes, tree.kids.front().comps, first) && union_box(boxes, tree.kids.back().comps, last)) {
      double axis[3];
      int major = 0;
      for (int k = 0; k < 3; ++k) {
        axis[k] = 0.5 * (last[2 * k] + last[2 * k + 1]) - 0.5 * (first[2 * k] + first[2 * k + 1]);
        if (std::fabs(axis[k]) > std::fabs(axis[major])) major = k;
      }
      if (std::fabs(axis[major]) > 0.0 && std::isfinite(axis[major])) pick_axis = major + (axis[major] < 0.0 ? 4 : 0);
    }
  }
  if (pick_axis < 0) {
    emit_groups(s, boxes, tree, false);
    return;
  }
  // layout: [PICK][mirror image][JUMP -> end][program proper]: a wave that runs along the axis jumps from
  // PICK straight to the program proper and pays one extra step; one that runs against it falls into the
  // mirror image and pays two
  DevInstr pick = Compiler::blank(I_BOX);
  pick.a1 = BOX_PICK;
  pick.a2 = pick_axis;  // 0 / 1 / 2: the program proper runs along +x / +y / +z, 4 / 5 / 6: along -x / -y / -z
  prog.code.push_back(pick);
  emit_groups(s, boxes, tree, true);
  DevInstr jump = Compiler::blank(I_BOX);
  jump.a1 = BOX_JUMP;
  prog.code.push_back(jump);
  const int mirror_steps = (int)prog.code.size();
  prog.code[0].a0 = mirror_steps - 1;                // PICK: over the mirror image and its JUMP, onto the program proper
  emit_groups(s, boxes, tree, false);
  prog.code[mirror_steps - 1].a0 = (int)prog.code.size() - mirror_steps;  // JUMP: over the program proper
  prog.mirror_steps = mirror_steps;
}

// Walk a compiled trace program the way the interpreter does (nearest_hit_n) and check what the kernels take
// for granted: every jump lands on the start of a step inside its own region, a chain record is never cut,
// and each direction of the program yields exactly one candidate per component.  A compiler bug then is an
// error from prt_scene_create, not a wave reading steps from beyond the table.
static const char* verify_program(const prt_scene* s, const Program& program, bool tracing) {
  const std::vector<DevInstr>& code = program.code;
  const int size = (int)code.size(), mirror = program.mirror_steps, n = (int)s->roots.size();
  const int box_kind = tracing ? BOX_TEST : BOX_LINE;  // (a render program's steps test the line of the ray)
  if (mirror < 0 || mirror > size) return "mirror image longer than the program";
  std::vector<char> starts((size_t)size + 1, 0);
  for (int pc = 0; pc < size; ++pc) {
    starts[pc] = 1;
    if (code[pc].kind == I_CHAIN) {
      if (pc + CHAIN_SLOTS > size) return "a chain record runs past the end";
      pc += CHAIN_SLOTS - 1;
    }
  }
  starts[size] = 1;
  struct Region { int from, to; };
  std::vector<Region> regions;
  if (mirror > 0) {
    if (mirror < 2 || code[0].kind != I_BOX || code[0].a1 != BOX_PICK || code[mirror - 1].kind != I_BOX ||
        code[mirror - 1].a1 != BOX_JUMP)
      return "a program in both directions without its PICK / JUMP frame";
    if (code[0].a0 + 1 != mirror || mirror - 1 + code[mirror - 1].a0 + 1 != size) return "PICK / JUMP land in the wrong place";
    if ((code[0].a2 & 3) > 2) return "PICK without an axis";
    regions.push_back({1, mirror - 1});
  }
  regions.push_back({mirror, size});
  for (const Region& r : regions) {
    int candidates = 0;
    for (int pc = r.from; pc < r.to; ++pc) {
      if (!starts[pc]) return "a step inside a chain record";
      const DevInstr& in = code[pc];
      if (in.kind == I_CHAIN) {
        candidates += 1;
        pc += CHAIN_SLOTS - 1;
      } else if (in.kind == I_BOX) {
        if (in.a1 != box_kind) return "a cull step of the wrong kind (PICK / JUMP inside a program, or the other program's test)";
        const int target = pc + in.a0 + 1;
        if (in.a0 < 1 || target > r.to || !starts[target]) return "a cull step jumps out of its region or into a record";
      } else if (in.kind == I_LEAF) {
        candidates += in.pad[0] == 2 ? 1 : 0;
      } else if (in.kind == I_CSG) {
        candidates += in.pad[0] == 1 ? 1 : 0;
      } else if (in.kind == I_ROOT) {
        candidates += 1;
      } else {
        return "a step of unknown kind";
      }
    }
    if (candidates != n) return "a direction of the program does not yield one candidate per component";
  }
  for (int c = 0; tracing && c < n; ++c) {
    const int from = program.component_first[2 * c], to = program.component_first[2 * c + 1];
    if (from < mirror || to > size || from >= to || !starts[from] || !starts[to]) return "a component's step range is off";
  }
  return nullptr;
}

// The render program: every component behind its line-of-sight step, and from kGroupMinComponents components on
// the same hierarchy of group steps as the trace program -- in LIST ORDER only: the renderers keep the first
// component among equal parameters (renderers.py:84, strict '<'), and their programs have no lexicographic
// running minimum that would let the visit leave that order.
static void compile_render_program(prt_scene* s) {
  const int n = (int)s->roots.size();
  if (n == 0) return;
  const bool grouped = n >= kGroupMinComponents && !s->options.no_cull && !s->options.no_groups;
  std::vector<ComponentBox> boxes(n);
  std::vector<int> all(n);
  for (int c = 0; c < n; ++c) {
    all[c] = c;
    boxes[c].ok = solid_bounds(s, s->roots[c], boxes[c].box) && short_direction_bound(s, s->roots[c]) <= kWellFormedLen2Lo;
  }
  emit_groups(s, boxes, build_groups(boxes, all, grouped, false), false, ROOT_RENDER);
}

static int validate_tree(const prt_scene* s, int node, int depth, std::vector<char>& seen) {
  if (node < 0 || node >= (int)s->nodes.size() || depth > 64 || seen[node]) return PRT_ERR_SCENE;
  seen[node] = 1;
  const prt_node& n = s->nodes[node];
  if (n.op == PRT_NODE_LEAF) {
    if (n.prim < 0 || n.prim >= (int)s->prims.size()) return PRT_ERR_SCENE;
    return PRT_OK;
  }
  if (n.op < PRT_NODE_UNION || n.op > PRT_NODE_DIFFERENCE) return PRT_ERR_SCENE;
  int rc = validate_tree(s, n.left, depth + 1, seen);
  if (rc) return rc;
  return validate_tree(s, n.right, depth + 1, seen);
}

static const size_t kMaxLdsBytes = 150 * 1024;  // one workgroup per CU at the very most (160 KiB LDS)

// options as the library keeps them: the caller's struct may be shorter (older header) or longer
static int read_options(const prt_scene_options* given, prt_scene_options* out) {
  std::memset(out, 0, sizeof(*out));
  if (given) {
    if (given->struct_size < (int32_t)(2 * sizeof(int32_t)))
      return fail(PRT_ERR_ARG, "prt_scene_options.struct_size is not set");
    std::memcpy(out, given, std::min((size_t)given->struct_size, sizeof(*out)));
  }
  out->struct_size = (int32_t)sizeof(*out);
  const int lanes = out->hit_lanes;
  if (!(lanes == 0 || lanes == 1 || lanes == 4 || lanes == 8 || lanes == 16))
    return fail(PRT_ERR_ARG, "prt_scene_options.hit_lanes must be 0, 1, 4, 8 or 16");
  if (out->cull_min < 0) return fail(PRT_ERR_ARG, "prt_scene_options.cull_min is negative");
  return PRT_OK;
}

extern "C" int prt_scene_create(const prt_prim* prims, int n_prims, const prt_node* nodes,
                                int n_nodes, const int32_t* roots, int n_roots,
                                const prt_material* mats, int n_mats, const prt_scene_options* options,
                                prt_scene** out) {
  if (!out) return fail(PRT_ERR_ARG, "out is null");
  *out = nullptr;
  if (n_prims < 0 || n_nodes < 0 || n_roots < 0 || n_mats < 0 ||
      (n_prims && !prims) || (n_nodes && !nodes) || (n_roots && !roots) || (n_mats && !mats))
    return fail(PRT_ERR_ARG, "null array with non-zero count");
  prt_scene_options opts;
  int rc_opts = read_options(options, &opts);
  if (rc_opts) return rc_opts;
  prt_scene* s = new prt_scene();
  s->options = opts;
  s->prims.assign(prims, prims + n_prims);
  s->nodes.assign(nodes, nodes + n_nodes);
  s->roots.assign(roots, roots + n_roots);
  s->mats.assign(mats, mats + n_mats);
  std::vector<char> seen(n_nodes, 0);
  for (int r : s->roots) {
    if (validate_tree(s, r, 0, seen) != PRT_OK) {
      delete s;
      return fail(PRT_ERR_SCENE, "malformed component tree (bad index, cycle, shared node or op)");
    }
  }
  for (const prt_prim& p : s->prims) {
    if (p.type < PRT_PRIM_SPHERE || p.type > PRT_PRIM_PARABOLOID || p.material < 0 ||
        p.material >= n_mats) {
      delete s;
      return fail(PRT_ERR_SCENE, "primitive with unknown type or material index");
    }
    const prt_material& m = s->mats[p.material];
    if (m.kind < PRT_MAT_NONE || m.kind > PRT_MAT_HOST) {
      delete s;
      return fail(PRT_ERR_SCENE, "unknown material kind");
    }
    DevPrim d;
    std::memset(&d, 0, sizeof(d));
    std::memcpy(d.minv, p.minv, sizeof(d.minv));
    std::memcpy(d.params, p.params, sizeof(d.params));
    std::memcpy(d.coef, m.coef, sizeof(d.coef));
    d.surface_id = (double)p.surface_id;
    d.type = p.type;
    d.mat_kind = m.kind;
    d.normal_scale = p.normal_scale < 0 ? -1 : 1;
    if (m.kind == PRT_MAT_NONE || m.kind == PRT_MAT_HOST || m.kind == PRT_MAT_TABLE) s->has_untracable = true;
    if (m.kind == PRT_MAT_TABLE) {  // no entries until prt_scene_set_index_tables: every look-up misses
      const double nan_index = m.coef[3];
      std::memset(d.coef, 0, sizeof(d.coef));
      d.coef[3] = nan_index;
    }
    s->dev_prims.push_back(d);
  }
  compile_trace_program(s);
  if (const char* wrong = verify_program(s, s->trace_program, true)) {
    const std::string message = std::string("internal error, the compiled trace program is malformed: ") + wrong;
    delete s;
    return fail(PRT_ERR_SCENE, message);
  }
  s->has_cull_steps = count_steps(s->trace_program.code, I_BOX) > 0;
  compile_render_program(s);
  if (const char* wrong = verify_program(s, s->render_program, false)) {
    const std::string message = std::string("internal error, the compiled render program is malformed: ") + wrong;
    delete s;
    return fail(PRT_ERR_SCENE, message);
  }
  for (int r : s->roots) {
    Program p;
    Operand res = compile_component(s, r, p.code, ROOT_NONE, &p.lds_slots);
    s->component_programs.push_back(p);
    s->component_result.push_back(res);
  }
  const size_t lds = (size_t)std::max(s->trace_program.lds_slots, s->render_program.lds_slots) * PRT_BLOCK * 12;
  if (lds > kMaxLdsBytes) {
    delete s;
    return fail(PRT_ERR_SCENE, "a component has too many surfaces for the per-lane LDS hit lists");
  }
  *out = s;
  return PRT_OK;
}

// The scene tables of one device as ONE image: [prims | trace code | component step ranges | render code | one program
// per component], each part at a 256-byte boundary.  `write` == nullptr: sizes only.  Returns the image's size; the
// offsets of the parts go to `at` (5 + components entries).
static size_t scene_image(const prt_scene* s, char* write, std::vector<size_t>* at) {
  size_t pos = 0;
  auto part = [&](const void* src, size_t bytes) {
    const size_t here = pos;
    if (write && bytes) std::memcpy(write + here, src, bytes);
    pos += (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    if (at) at->push_back(here);
  };
  if (at) at->clear();
  part(s->dev_prims.data(), s->dev_prims.size() * sizeof(DevPrim));
  part(s->trace_program.code.data(), s->trace_program.code.size() * sizeof(DevInstr));
  part(s->trace_program.component_first.data(), s->trace_program.component_first.size() * sizeof(int32_t));
  part(s->render_program.code.data(), s->render_program.code.size() * sizeof(DevInstr));
  for (const Program& p : s->component_programs) part(p.code.data(), p.code.size() * sizeof(DevInstr));
  return pos;
}

// what a ticket needs on the host side, made when the ticket traces for the first time (most scenes only ever use
// ticket 0, and a scene's first trace should not pay for four)
static int ticket_resources(TraceTicket& t) {
  if (t.mirror) return PRT_OK;
  HIP_TRY(hipHostMalloc((void**)&t.host_pinned, 16 * sizeof(int64_t), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&t.mirror, 64 + (kMaxBatch + 4) * 64, hipHostMallocMapped | hipHostMallocCoherent));  // header + GenCtrl slots of 64 B
  std::memset((void*)t.mirror, 0, 64 + (kMaxBatch + 4) * 64);
  HIP_TRY(hipHostGetDevicePointer((void**)&t.mirror_dev, (void*)t.mirror, 0));
  HIP_TRY(hipEventCreate(&t.ev0));
  HIP_TRY(hipEventCreate(&t.ev1));
  return PRT_OK;
}

// give back whatever a device copy holds (the device is current)
static void release_device_copy(DeviceCopy& c) {
  (void)hipFree(c.block);  // (prims, trace_code, trace_component_first, render_code, component_code[] point into it)
  (void)hipFree(c.tables);
  if (c.staging) (void)hipHostFree(c.staging);
  if (c.update_stream) (void)hipStreamDestroy(c.update_stream);
  if (c.update_event) (void)hipEventDestroy(c.update_event);
  if (c.tail_event) (void)hipEventDestroy(c.tail_event);
  for (TraceTicket& t : c.ticket) {
    if (t.host_pinned) (void)hipHostFree(t.host_pinned);
    if (t.mirror) (void)hipHostFree(t.mirror);
    if (t.ev0) (void)hipEventDestroy(t.ev0);
    if (t.ev1) (void)hipEventDestroy(t.ev1);
    t.busy0 = t.busy1 = nullptr;
    free((void*)t.plan_host);  // (malloc'ed by prt_trace_set_plan)
    if (t.plan_dev) (void)hipFree(t.plan_dev);
    if (t.sink_slots) (void)hipFree(t.sink_slots);
  }
  for (hipEvent_t e : c.busy_events) (void)hipEventDestroy(e);
  c = DeviceCopy();
}

extern "C" void prt_scene_destroy(prt_scene* s) {
  if (!s) return;
  for (size_t d = 0; d < s->per_device.size(); ++d) {
    DeviceCopy& c = s->per_device[d];
    if (!c.block) continue;
    (void)hipSetDevice((int)d);
    // a trace that was begun and never ended, or one whose counts are out while its last stores are not,
    // still reads the tables and writes the host mirror freed below
    (void)hipDeviceSynchronize();
    release_device_copy(c);
  }
  delete s;
}

// Put the scene's index tables on a device copy (the device is current and idle) and point the records of the
// primitives with a table glass at them: coef[0] / coef[2] = device addresses of the material's wavelengths /
// indices (as bits), coef[1] = how many; the copy's primitive table is rewritten from s->dev_prims.
static int apply_tables(prt_scene* s, DeviceCopy& c) {
  const size_t total = s->table_wavelengths.size();
  bool any = false;
  for (const DevPrim& d : s->dev_prims) any = any || d.mat_kind == MAT_TABLE;
  if (!any) return PRT_OK;
  (void)hipFree(c.tables);
  c.tables = nullptr;
  HIP_TRY(hipMalloc((void**)&c.tables, std::max<size_t>(1, 2 * total) * sizeof(double)));
  if (total) {
    HIP_TRY(hipMemcpy(c.tables, s->table_wavelengths.data(), total * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.tables + total, s->table_indices.data(), total * sizeof(double), hipMemcpyHostToDevice));
  }
  std::vector<DevPrim> patched(s->dev_prims);
  for (size_t p = 0; p < patched.size(); ++p) {
    DevPrim& d = patched[p];
    if (d.mat_kind != MAT_TABLE) continue;
    const size_t m = (size_t)s->prims[p].material;
    int64_t first = 0, count = 0;
    if (2 * m + 1 < s->table_ranges.size()) { first = s->table_ranges[2 * m]; count = s->table_ranges[2 * m + 1]; }
    const unsigned long long lam = (unsigned long long)(c.tables + first), idx = (unsigned long long)(c.tables + total + first);
    std::memcpy(&d.coef[0], &lam, sizeof(double));
    std::memcpy(&d.coef[2], &idx, sizeof(double));
    d.coef[1] = (double)count;
  }
  HIP_TRY(hipMemcpy(c.prims, patched.data(), patched.size() * sizeof(DevPrim), hipMemcpyHostToDevice));
  return PRT_OK;
}

extern "C" int prt_scene_set_index_tables(prt_scene* s, const int64_t* ranges, int n_mats, const double* wavelengths,
                                          const double* indices, int64_t total) {
  if (!s) return fail(PRT_ERR_ARG, "scene is null");
  if (n_mats != (int)s->mats.size() || !ranges || total < 0 || (total && (!wavelengths || !indices)))
    return fail(PRT_ERR_ARG, "ranges must hold (first, count) for every material of the scene");
  for (const DeviceCopy& c : s->per_device)
    for (const TraceTicket& tk : c.ticket)
      if (tk.active) return fail(PRT_ERR_ARG, "a trace of this scene is in flight (prt_trace_end it first)");
  for (int m = 0; m < n_mats; ++m) {
    if (s->mats[m].kind != PRT_MAT_TABLE) continue;
    const int64_t first = ranges[2 * m], count = ranges[2 * m + 1];
    if (first < 0 || count < 0 || first + count > total || count >= (1ll << 31))
      return fail(PRT_ERR_ARG, "index table range out of bounds");
    for (int64_t k = first + 1; k < first + count; ++k)
      if (!(wavelengths[k - 1] < wavelengths[k]))
        return fail(PRT_ERR_ARG, "index table wavelengths must be strictly ascending (and not NaN) inside a material's range");
  }
  s->table_ranges.assign(ranges, ranges + 2 * (size_t)n_mats);
  // (only the ranges checked above are kept: a slot that is not a table now holds (0, 0), so a later
  // prt_scene_update that turns it into one finds an empty table -- every look-up misses -- not an unchecked range)
  for (int m = 0; m < n_mats; ++m)
    if (s->mats[m].kind != PRT_MAT_TABLE) s->table_ranges[2 * m] = s->table_ranges[2 * m + 1] = 0;
  s->table_wavelengths.assign(wavelengths, wavelengths + total);
  s->table_indices.assign(indices, indices + total);
  for (size_t d = 0; d < s->per_device.size(); ++d) {
    DeviceCopy& c = s->per_device[d];
    if (!c.ready) continue;
    HIP_TRY(hipSetDevice((int)d));
    HIP_TRY(hipDeviceSynchronize());  // the last trace may still be draining: it reads the old tables
    int rc = apply_tables(s, c);
    if (rc) return rc;
  }
  return PRT_OK;
}

// The same scene with other numbers in it (a part moved, a radius or a glass changed): recompile on
// the host and overwrite the tables in place -- device buffers, the host mirror, the events, the
// hints of the previous trace and the telemetry all stay.  Returns 1 (and leaves the scene as it
// was) when the new description does not have the old one's shape, i.e. when a program or table
// would change size: the caller then builds a new scene.
extern "C" int prt_scene_update(prt_scene* s, const prt_prim* prims, int n_prims, const prt_node* nodes,
                                int n_nodes, const int32_t* roots, int n_roots, const prt_material* mats,
                                int n_mats, const prt_scene_options* options) {
  if (!s) return fail(PRT_ERR_ARG, "scene is null");
  for (const DeviceCopy& c : s->per_device)
    for (const TraceTicket& tk : c.ticket)
      if (tk.active) return fail(PRT_ERR_ARG, "a trace of this scene is in flight (prt_trace_end it first)");
  prt_scene* t = nullptr;
  const int rc = prt_scene_create(prims, n_prims, nodes, n_nodes, roots, n_roots, mats, n_mats,
                                  options ? options : &s->options, &t);
  if (rc) return rc;
  auto same_program = [](const Program& a, const Program& b) {
    return a.code.size() == b.code.size() && a.lds_slots == b.lds_slots &&
           a.component_first.size() == b.component_first.size() && a.mirror_steps == b.mirror_steps;
  };
  bool same = t->dev_prims.size() == s->dev_prims.size() && same_program(t->trace_program, s->trace_program) &&
              same_program(t->render_program, s->render_program) &&
              t->component_programs.size() == s->component_programs.size() &&
              t->has_untracable == s->has_untracable;
  for (size_t k = 0; same && k < t->component_programs.size(); ++k)
    same = same_program(t->component_programs[k], s->component_programs[k]);
  if (!same) {
    prt_scene_destroy(t);
    return 1;
  }
  bool has_tables = false;
  for (const DevPrim& d : t->dev_prims) has_tables = has_tables || d.mat_kind == MAT_TABLE;
  for (const DevPrim& d : s->dev_prims) has_tables = has_tables || d.mat_kind == MAT_TABLE;
  for (size_t d = 0; d < s->per_device.size(); ++d) {
    DeviceCopy& c = s->per_device[d];
    if (!c.ready) continue;
    HIP_TRY(hipSetDevice((int)d));
    if (scene_image(t, nullptr, nullptr) != c.block_bytes) { prt_scene_destroy(t); return fail(PRT_ERR_SCENE, "scene image changed its size"); }
    // The new image goes over the old one with ONE copy out of pinned memory on a stream of the copy's own, ordered
    // BEHIND whatever the tickets have enqueued (the last kernel of a trace whose counts are out may still be reading
    // the tables) and IN FRONT of whatever they enqueue next (prt_trace_begin makes its stream wait for update_event):
    // no device-wide synchronisation, nothing the host waits for -- a design loop moves a part before every trace.
    // Entry points outside the ticket runtime (prt_propagate, the renderers ...) are not tracked stream by stream: if
    // one of them used the tables since the last update, or the scene has index tables to re-apply, the device is
    // synchronised first, as it always was.
    if (!c.staging) {  // the first update of this device copy
      HIP_TRY(hipHostMalloc((void**)&c.staging, c.block_bytes, hipHostMallocDefault));
      std::memset(c.staging, 0, c.block_bytes);
      HIP_TRY(hipStreamCreateWithFlags(&c.update_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&c.update_event, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c.tail_event, hipEventDisableTiming));
    }
    if (c.update_unsettled) {  // (the previous update's copy reads the staging memory written below: long done as a rule)
      if (hipEventQuery(c.update_event) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(c.update_event));
      }
      c.update_unsettled = false;
    }
    bool ordered = !c.untracked_use && !has_tables;
    if (ordered) {
      for (TraceTicket& tk : c.ticket) {
        if (!tk.traced) continue;
        if (hipEventRecord(c.tail_event, tk.st) != hipSuccess || hipStreamWaitEvent(c.update_stream, c.tail_event, 0) != hipSuccess) {
          (void)hipGetLastError();
          ordered = false;  // (e.g. a stream the caller has destroyed since)
          break;
        }
      }
    }
    if (!ordered) HIP_TRY(hipDeviceSynchronize());
    for (TraceTicket& tk : c.ticket) tk.traced = false;
    c.untracked_use = false;
    scene_image(t, c.staging, nullptr);
    if (ordered) {
      HIP_TRY(hipMemcpyAsync(c.block, c.staging, c.block_bytes, hipMemcpyHostToDevice, c.update_stream));
      HIP_TRY(hipEventRecord(c.update_event, c.update_stream));
      c.update_serial += 1;
      c.update_unsettled = true;
    } else {
      HIP_TRY(hipMemcpy(c.block, c.staging, c.block_bytes, hipMemcpyHostToDevice));
      c.update_unsettled = false;
    }
  }
  s->prims.swap(t->prims);
  s->nodes.swap(t->nodes);
  s->roots.swap(t->roots);
  s->mats.swap(t->mats);
  s->dev_prims.swap(t->dev_prims);
  std::swap(s->trace_program, t->trace_program);
  std::swap(s->render_program, t->render_program);
  s->options = t->options;
  s->has_cull_steps = t->has_cull_steps;
  s->spatial_groups = t->spatial_groups;
  s->component_programs.swap(t->component_programs);
  s->component_result.swap(t->component_result);
  prt_scene_destroy(t);  // (never reached a device: host memory only)
  // (the primitive records were overwritten: the table glasses get their table addresses back; a material list
  // of another shape leaves them without entries until the caller sets the tables again)
  if (s->table_ranges.size() != 2 * s->mats.size()) {
    s->table_ranges.clear(); s->table_wavelengths.clear(); s->table_indices.clear();
  }
  for (size_t d = 0; d < s->per_device.size(); ++d) {
    DeviceCopy& c = s->per_device[d];
    if (!c.ready) continue;
    HIP_TRY(hipSetDevice((int)d));
    int rc_tables = apply_tables(s, c);
    if (rc_tables) return rc_tables;
  }
  return PRT_OK;
}

extern "C" int prt_scene_info(const prt_scene* s, int64_t* out10) {
  int64_t* out8 = out10;
  if (!s || !out10) return fail(PRT_ERR_ARG, "null argument");
  out10[8] = s->spatial_groups ? 1 : 0;
  out10[9] = s->trace_program.mirror_steps > 0 ? 1 : 0;
  const std::vector<DevInstr> forward(s->trace_program.code.begin() + s->trace_program.mirror_steps, s->trace_program.code.end());
  const int64_t culls = count_steps(forward, I_BOX);  // (a grouped program is followed by its mirror image: not counted)
  out8[0] = (int64_t)s->prims.size();
  out8[1] = (int64_t)s->roots.size();
  out8[2] = (int64_t)forward.size();
  out8[3] = s->trace_program.lds_slots;
  out8[4] = culls;
  out8[5] = (int64_t)s->render_program.code.size();
  out8[6] = s->render_program.lds_slots;
  out8[7] = count_steps(forward, I_CHAIN);
  return PRT_OK;
}

extern "C" int prt_scene_component_rows(const prt_scene* s, int root) {
  if (!s || root < 0 || root >= (int)s->roots.size()) return fail(PRT_ERR_ARG, "bad component index");
  return 2 * leaves_under(s, s->roots[root]);
}

static int raise_lds_limits();  // defined after the kernels

// make sure the scene tables exist on `device` and make it current
// tracked: the caller is the ticket runtime, whose launches prt_scene_update orders its copy behind (and which makes
// its streams wait for that copy); any other entry point is noted, and waits on the host for an update in flight
static int on_device(prt_scene* s, int device, DeviceCopy** out, bool tracked = false) {
  if (!s) return fail(PRT_ERR_ARG, "scene is null");
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return fail(PRT_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  if ((int)s->per_device.size() <= device) s->per_device.resize(device + 1);
  DeviceCopy& c = s->per_device[device];
  if (!c.ready) {
    release_device_copy(c);  // (what an earlier attempt that failed half way left behind)
    int rc = PRT_OK;
    std::vector<size_t> at;
    c.block_bytes = scene_image(s, nullptr, &at);
    HIP_TRY(hipMalloc((void**)&c.block, c.block_bytes));
    {
      // (the pinned image, the stream and the events of prt_scene_update are made by the first update: a scene that is
      // never updated -- and a scene's first trace -- does not pay for them)
      std::vector<char> image(c.block_bytes, 0);
      scene_image(s, image.data(), nullptr);
      HIP_TRY(hipMemcpy(c.block, image.data(), c.block_bytes, hipMemcpyHostToDevice));
    }
    c.prims = (DevPrim*)(c.block + at[0]);
    c.trace_code = (DevInstr*)(c.block + at[1]);
    c.trace_component_first = (int32_t*)(c.block + at[2]);
    c.render_code = (DevInstr*)(c.block + at[3]);
    for (size_t k = 0; k < s->component_programs.size(); ++k) c.component_code.push_back((DevInstr*)(c.block + at[4 + k]));
    // (a ticket's pinned staging, host mirror and events are made when it is first used: ticket_resources)
    int rc_lds = raise_lds_limits();
    if (rc_lds) return rc_lds;
    rc = apply_tables(s, c);
    if (rc) return rc;
    c.ready = true;
  }
  if (!tracked) {
    c.untracked_use = true;
    if (c.update_unsettled) {  // (an update's copy may still be on its way: this caller's stream is not known here)
      HIP_TRY(hipEventSynchronize(c.update_event));
      c.update_unsettled = false;
    }
  }
  *out = &c;
  return PRT_OK;
}

