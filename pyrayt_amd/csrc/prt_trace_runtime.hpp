// prt_trace_runtime.hpp -- the host runtime of a whole trace: workspace layout, the four asynchronous tickets of a
// scene and device (prt_trace_begin / prt_trace_end, prt_trace = begin + end, prt_trace_batch), how generations are
// launched in batches without a host round trip, which compaction form each launch is offered from what the scene's
// previous trace did (hints, per-tile records, dead lists) and what happens when a tile refutes it, the three-kernel
// fallback, statistics and telemetry.  Host code; included by prt_kernels.hip.
#pragma once
// workspace of prt_trace:
//   ctrl | fused ctrl | generation slots | rows_per_generation (device) | tile words A | B
//   | block counts | block offsets | hit_t (n f64) | hit_prim (n i32)      [unfused path only]
//   | ray buffer A (13 n) | ray buffer B (13 n)
struct TraceLayout {
  size_t ctrl, fctrl, gen, dead_lists, gen_rows, tiles_a, tiles_b, counts, offsets, hit_t, hit_prim, rays_a,
      rays_b, total;
};
static TraceLayout trace_layout(int64_t n) {
  const size_t nb = blocks_for(n) + 1;
  const size_t nn = (size_t)(n < 1 ? 1 : n);
  TraceLayout l;
  size_t at = 0;
  auto take = [&](size_t bytes) { size_t here = at; at += align_up(bytes, 256); return here; };
  l.ctrl = take(sizeof(TraceCtrl));
  // the fused path's control header sits right in front of its generation slots so that the
  // host reads both back with one copy
  l.fctrl = take(sizeof(GenCtrl) + (kMaxGenerationSlots + 1) * sizeof(GenCtrl));
  l.gen = l.fctrl + sizeof(GenCtrl);
  l.dead_lists = take(3 * sizeof(DeadList));                                   // (at kDeadListOffset from gen[0]: dead_list())
  l.gen_rows = take(kMaxGenerationSlots * sizeof(int64_t));
  l.tiles_a = take(nb * sizeof(unsigned long long));
  l.tiles_b = take(nb * sizeof(unsigned long long));
  l.counts = take(nb * 2 * sizeof(int32_t));
  l.offsets = take(nb * 2 * sizeof(int64_t));
  l.hit_t = take(nn * sizeof(double));
  l.hit_prim = take(nn * sizeof(int32_t));
  l.rays_a = take(nn * PRT_RAY_ROWS * sizeof(double));
  l.rays_b = take(nn * PRT_RAY_ROWS * sizeof(double));
  l.total = at;
  if (l.dead_lists - l.gen != kDeadListOffset) abort();  // (the kernel finds them by this constant)
  return l;
}

extern "C" int64_t prt_trace_workspace_bytes(int64_t n) {
  return (int64_t)trace_layout(n < 0 ? 0 : n).total;
}

static int64_t trace_error(int error) {
  if (error == PRT_ERR_ROWS_CAP) return fail(PRT_ERR_ROWS_CAP, "rows_cap too small");
  if (error == PRT_ERR_UNTRACABLE)
    return fail(PRT_ERR_UNTRACABLE, "a ray hit a surface whose material has no trace() (or one shaded by the caller: "
                                    "PRT_MAT_HOST surfaces are served by prt_propagate / prt_gather_hits / prt_interact)");
  if (error == PRT_ERR_WAVELENGTH)
    return fail(PRT_ERR_WAVELENGTH, "a ray's wavelength is not in the index table of the glass it hit "
                                    "(prt_scene_set_index_tables)");
  if (error == PRT_ERR_STALL) return PRT_ERR_STALL;
  return fail(error, "device error during trace");
}

// Who traced last with a given workspace address: a ticket skips re-initialising the control words only
// if nobody else used the block since its own last trace (another scene, another ticket, or the other
// trace path of the same one marks the block as theirs before touching it).
static std::mutex g_workspace_mutex;
static std::unordered_map<const void*, unsigned long long> g_workspace_user;
static std::atomic<unsigned long long> g_next_user{1};
static bool workspace_taken_over(const void* w, unsigned long long user) {
  std::lock_guard<std::mutex> lock(g_workspace_mutex);
  // (addresses that were freed long ago would stay in here for ever: forgetting everybody is always safe,
  // a forgotten owner just re-initialises its control words once)
  if (g_workspace_user.size() > 4096) g_workspace_user.clear();
  unsigned long long& last = g_workspace_user[w];
  const bool same = last == user;
  last = user;
  return !same;
}

// three kernels per generation + a host round trip (kept for A/B runs and cross-checks)
static int64_t trace_unfused(prt_scene* s, DeviceCopy* c, TraceTicket* t, int64_t* rows_per_generation) {
  const int64_t n = t->n;
  const int generation_limit = t->limit;
  char* w = t->w;
  hipStream_t st = t->st;
  const TraceLayout l = trace_layout(n);
  TraceCtrl* ctrl = (TraceCtrl*)(w + l.ctrl);
  int64_t* gen_rows = (int64_t*)(w + l.gen_rows);
  int32_t* counts = (int32_t*)(w + l.counts);
  int64_t* offsets = (int64_t*)(w + l.offsets);
  double* hit_t = (double*)(w + l.hit_t);
  int32_t* hit_prim = (int32_t*)(w + l.hit_prim);
  double* buf[2] = {(double*)(w + l.rays_a), (double*)(w + l.rays_b)};
  const int keep_absorbed = (t->flags & PRT_TRACE_KEEP_ABSORBED) ? 1 : 0;
  // PRT_TRACE_COUNT_PATHS: the nearest-hit kernel of this path counts (one ray per lane only: the k-lanes
  // kernels evaluate components per lane and would count a ray once per lane group member)
  const bool count_paths = (t->flags & PRT_TRACE_COUNT_PATHS) != 0 && s->options.hit_lanes <= 1 && !s->options.hit_staged;
  if (t->user == 0) t->user = g_next_user.fetch_add(1);
  (void)workspace_taken_over(w, t->user);  // the block is ours now: a fused trace behind this one re-initialises
  t->ready_workspace = nullptr;

  SceneDev sd = trace_scene_dev(s, c);
  hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, st, ctrl, n, t->rows_cap);

  const double* src = t->rays;
  int64_t src_ld = t->ld;
  int64_t n_cur = n, total_rows = 0;
  int error = 0;
  for (int g = 0; g < generation_limit && n_cur > 0; ++g) {
    double* dst = buf[g & 1];
    const unsigned nb = blocks_for(n_cur);
    const int relaunch = (g + 1 != generation_limit) ? 1 : 0;
    HIP_TRY(hipEventRecord(t->ev0, st));
    {
      int rc_hit = launch_hit(s, c, sd, st, src, src_ld, (const TraceCtrl*)ctrl, (int64_t)0, n_cur, hit_t, hit_prim,
                              (int64_t*)nullptr, counts, keep_absorbed, count_paths ? ctrl->paths : nullptr);
      if (rc_hit) return rc_hit;
    }
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, (const int32_t*)counts, offsets, ctrl);
    hipLaunchKernelGGL(k_shade, dim3(nb), dim3(PRT_BLOCK), 0, st, sd, src, src_ld,
                       (const TraceCtrl*)ctrl, (int64_t)0, (const double*)hit_t,
                       (const int32_t*)hit_prim, (const int64_t*)nullptr, (const int64_t*)offsets,
                       dst, n, t->rows_out, t->rows_cap, (int64_t)0, (double)(g + 1), relaunch,
                       t->ray_offset, keep_absorbed, ctrl, (const double*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, ctrl, gen_rows, g);
    HIP_TRY(hipEventRecord(t->ev1, st));
    // the host needs the new ray count to size the next launch
    HIP_TRY(hipMemcpyAsync(t->host_pinned, ctrl, sizeof(TraceCtrl), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(t->host_pinned + 12, gen_rows + g, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev0, t->ev1));
    const TraceCtrl* h = (const TraceCtrl*)t->host_pinned;
    t->stats[0] += 1;
    t->stats[1] += (double)n_cur;
    t->stats[2] += ms;
    t->stats[3] += 4;
    if (h->error) { error = h->error; break; }
    const int64_t live = t->host_pinned[12];
    rows_per_generation[g] = live;
    total_rows += live;
    t->stats[4] += (double)live;
    t->stats[5] += (double)h->n_cur;
    n_cur = h->n_cur;
    src = dst;
    src_ld = n;
  }
  if (error) return trace_error(error);
  if (count_paths) {  // (the control block of the last generation is on the host: the counters are cumulative)
    const TraceCtrl* h = (const TraceCtrl*)t->host_pinned;
    s->path_counts[0] += 1;
    for (int k = 1; k < 4; ++k) s->path_counts[k] += (long long)h->paths[k];
  }
  return total_rows;
}

// one kernel per generation; generations are launched in batches with no host round trip in
// between (a generation whose predecessor left no rays exits in its prologue)
static const int kGenerationBatch = 4;

// add the HIP-event time of the ticket's last batch to its kernel-time statistic (waits for ev1 if
// need be: by the time anybody asks, the batch has long finished)
static int settle_timing(TraceTicket* t) {
  if (!t->timing_pending) return PRT_OK;
  t->timing_pending = false;
  HIP_TRY(hipEventSynchronize(t->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, t->ev0, t->ev1));
  t->stats[2] += ms;
  return PRT_OK;
}

// spin on the epoch word of the host mirror; gives up after ~2 s of polling and lets the stream
// synchronisation report whatever went wrong
static int await_epoch(TraceTicket* t, unsigned long long epoch) {
  volatile unsigned long long* word = &t->mirror->epoch;
  for (long spins = 0; __atomic_load_n(word, __ATOMIC_ACQUIRE) != epoch; ++spins) {
    __builtin_ia32_pause();
    if (spins > (1l << 28)) {
      HIP_TRY(hipStreamSynchronize(t->st));
      if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != epoch) return fail(PRT_ERR_HIP, "trace batch never published its counts");
      break;
    }
  }
  return PRT_OK;
}

#ifdef PRT_HOST_PROFILE
#include <time.h>
static double g_hp[8]; static long g_hp_n;
static inline double hp_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
#define HP(k) g_hp_t[k] = hp_now()
static double g_hp_t[8];
extern "C" void prt_debug_host_profile(double* out9) { for (int k = 0; k < 8; ++k) out9[k] = g_hp[k]; out9[8] = (double)g_hp_n; }
#else
#define HP(k)
#endif

// ---- record plans (include/prt.h prt_record_plan; the device side is PlanDev in prt_trace_kernels.hpp) -------------
// the plan as the kernels read it, put into the ticket's device copy on the trace's stream
__global__ void k_plan_set(PlanDev* dst, PlanDev value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}
// out[e] = the sum over the slots of slotted[slot][e]; the slots are left zeroed for the ticket's next trace
__global__ void __launch_bounds__(PRT_BLOCK)
k_sink_fold(double* __restrict__ slotted, int slots, int n, double* __restrict__ out) {
  const int e = blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (e >= n) return;
  double v = 0.0;
  for (int k = 0; k < slots; ++k) {
    v += slotted[(size_t)k * n + e];
    slotted[(size_t)k * n + e] = 0.0;
  }
  out[e] = v;
}

static unsigned long long plan_key_of(const PlanDev& p) {  // what the plan's dense-mode hints depend on
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
  mix((unsigned long long)p.n_rec);
  mix((unsigned long long)(p.store_rows != 0));
  // (the columns a stored row writes do not enter: the plan's dense forms are about WHICH rays store a row)
  for (int k = 0; k < p.n_rec; ++k) mix((unsigned long long)(unsigned)p.rec_prims[k]);
  return h | 1ull;
}

extern "C" int prt_trace_set_plan(prt_scene* s, int device, int ticket, const prt_record_plan* plan) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c, true);
  if (rc) return rc;
  if (ticket < 0 || ticket >= PRT_TRACE_TICKETS) return fail(PRT_ERR_ARG, "ticket out of range");
  TraceTicket* t = &c->ticket[ticket];
  if (t->active) return fail(PRT_ERR_ARG, "this ticket has a trace in flight (prt_trace_end it first)");
  if (!plan) {
    t->plan_active = false;
    return PRT_OK;
  }
  if (plan->struct_size != (int32_t)sizeof(prt_record_plan)) return fail(PRT_ERR_ARG, "prt_record_plan.struct_size does not match this library");
  if (plan->n_surfaces < 0 || plan->n_surfaces > 8 || plan->n_groups < 0 || plan->generation_limit < 1 ||
      plan->generation_limit > kMaxGenerationSlots)
    return fail(PRT_ERR_ARG, "record plan: 0..8 surfaces, n_groups >= 0, 1 <= generation_limit <= 1024");
  if (plan->n_groups > 0 && !plan->sums_out) return fail(PRT_ERR_ARG, "record plan: n_groups > 0 needs sums_out");
  if (!(plan->rays_per_source > 0) && plan->n_groups > 1) return fail(PRT_ERR_ARG, "record plan: one group without rays_per_source");
  if (plan->ms_quantity > PRT_FRAME_AXIS_INTERCEPT || plan->ms_transform < 0 || plan->ms_transform > 1)
    return fail(PRT_ERR_ARG, "record plan: ms_quantity is a frame column 0..14, 15 (axis intercept) or < 0; ms_transform 0 | 1");
  if (!plan->store_rows && plan->n_groups == 0 && plan->n_surfaces == 0)
    return fail(PRT_ERR_ARG, "record plan: neither rows nor sums are asked for");
  PlanDev p;
  memset(&p, 0, sizeof(p));
  p.n_rec = plan->n_surfaces;
  for (int k = 0; k < plan->n_surfaces; ++k) {
    // (a surface id that is none of the scene's passes no row: a primitive index no ray can hit)
    p.rec_prims[k] = -2;
    for (size_t q = 0; q < s->dev_prims.size(); ++q)
      if ((int64_t)s->dev_prims[q].surface_id == plan->surfaces[k]) p.rec_prims[k] = (int32_t)q;
  }
  p.store_rows = plan->store_rows ? 1 : 0;
  p.n_groups = plan->n_groups;
  p.rays_per_source = plan->rays_per_source;
  p.pivots = plan->pivots;
  p.ms_quantity = plan->ms_quantity < 0 ? -1 : plan->ms_quantity;
  p.ms_transform = plan->ms_transform;
  p.ms_about = plan->ms_about;
  p.limit = plan->generation_limit;
  p.columns = (plan->columns & 0x7fff) ? (plan->columns & 0x7fff) : 0x7fff;
  // the slotted sums: as many copies as stay under 8 MiB, at most 64, a power of two
  p.slots = 1;
  size_t per_slot = (size_t)p.limit * (size_t)(p.n_groups > 0 ? p.n_groups : 0) * SINK_STATS * sizeof(double);
  if (p.n_groups > 0) {
    while (p.slots < 64 && per_slot * (size_t)p.slots * 2 <= ((size_t)8 << 20)) p.slots *= 2;
    const size_t need = per_slot * (size_t)p.slots;
    if (need > t->sink_slot_bytes) {
      HIP_TRY(hipDeviceSynchronize());  // (nothing of an earlier plan's traces may still add to the block given back)
      if (t->sink_slots) (void)hipFree(t->sink_slots);
      t->sink_slots = nullptr;
      t->sink_slot_bytes = 0;
      HIP_TRY(hipMalloc((void**)&t->sink_slots, need));
      t->sink_slot_bytes = need;
    }
    t->sink_unclean = true;  // (cleared on the stream of the first trace under the plan)
  }
  p.sums = t->sink_slots;
  if (!t->plan_dev) HIP_TRY(hipMalloc((void**)&t->plan_dev, sizeof(PlanDev)));
  if (!t->plan_host) t->plan_host = (PlanDev*)malloc(sizeof(PlanDev));
  if (!t->plan_host) return fail(PRT_ERR_HIP, "out of host memory");
  *t->plan_host = p;
  t->sums_out = plan->sums_out;
  t->plan_key = plan_key_of(p);
  t->plan_dirty = true;
  t->plan_active = true;
  return PRT_OK;
}

// behind a trace under a plan that sums: fold the slots into the caller's block (and leave them zeroed)
static int plan_fold(TraceTicket* t) {
  const PlanDev& p = *t->plan_host;
  if (p.n_groups <= 0) return PRT_OK;
  const int n = p.limit * p.n_groups * SINK_STATS;
  hipLaunchKernelGGL(k_sink_fold, dim3((n + PRT_BLOCK - 1) / PRT_BLOCK), dim3(PRT_BLOCK), 0, t->st, t->sink_slots,
                     p.slots, n, t->sums_out);
  HIP_TRY(hipGetLastError());
  t->sink_unclean = false;
  return PRT_OK;
}

// enqueue generations [t->g, t->g + t->batch) of the ticket's trace (one launch each, no host round trip
// in between) and whatever has to run behind them; they publish t->epoch to the ticket's mirror
static int fused_launch_batch(prt_scene* s, DeviceCopy* c, TraceTicket* t) {
  const int64_t n = t->n;
  const TraceLayout l = trace_layout(n);
  char* w = t->w;
  FusedCtrl* ctrl = (FusedCtrl*)(w + l.fctrl);
  GenCtrl* gen = (GenCtrl*)(w + l.gen);
  unsigned long long* tiles[2] = {(unsigned long long*)(w + l.tiles_a), (unsigned long long*)(w + l.tiles_b)};
  double* buf[2] = {(double*)(w + l.rays_a), (double*)(w + l.rays_b)};
  const int keep_absorbed = (t->flags & PRT_TRACE_KEEP_ABSORBED) ? 1 : 0;
  const SceneDev sd = trace_scene_dev(s, c);
  // (shading from an LDS copy of the primitive table was tried in two forms -- per lane without the
  // waterfall, and inside the waterfall -- and measured 2-17 % slower than the batched L1 reads)
  size_t lds = lds_bytes_fused(sd.lds_slots);
  hipStream_t st = t->st;
  int rc = settle_timing(t);
  if (rc) return rc;
  HP(2);
  const unsigned long long epoch = ++t->epoch;
  const bool timed = !(t->flags & PRT_TRACE_NO_TIMING);
  if (timed) HIP_TRY(hipEventRecord(t->ev0, st));
  // PRT_TRACE_BUSY: the trace's own pair, from in front of its first launch to behind its last (a further batch, or a
  // repeated attempt, moves the end -- resp. both -- along)
  if (t->busy0 && t->g == 0) HIP_TRY(hipEventRecord(t->busy0, st));
  HP(3);
  const bool culls = s->has_cull_steps;
  auto kernel = t->compact ? (culls ? k_generation<true, true> : k_generation<false, true>)
                           : (culls ? k_generation<true, false> : k_generation<false, false>);
  if (t->plan_active)
    kernel = t->compact ? (culls ? k_generation<true, true, true> : k_generation<false, true, true>)
                        : (culls ? k_generation<true, false, true> : k_generation<false, false, true>);
  for (int b = 0; b < t->batch; ++b) {
    const int gg = t->g + b;
    const double* src = (gg == 0) ? t->rays : buf[(gg - 1) & 1];
    const int64_t src_ld = (gg == 0) ? t->ld : n;
    if (t->plan_active) {
      // A trace under a record plan: the dense forms it knows are "every ray goes on" / "none does", each with
      // "every ray stores a row" / "none does" (k_generation<.., PLAN>); anything else compacts by look-back.  Bit 5 of a
      // hint (plans that store no rows): the generation lost a few absorbed rays last time -- it is launched dense with
      // its absorbed rays kept, as upstream carries them (_pyrayt.py:415-428): nothing of theirs is stored or summed by
      // the generation behind, which finds them dead on arrival (the product path's sparse-loss form without its dead
      // lists: with no rows to place, dead lanes cost the next generation nothing but their slots).
      int assume = 0;
      if (t->use_hints && gg < (int)s->plan_hint_mode.size()) assume = s->plan_hint_mode[gg];
      t->launch_mode[gg] = (char)assume;
      s->plan_launches += 1;
      s->plan_dense_launches += assume ? 1 : 0;
      const int keep = (keep_absorbed || (assume & 32)) ? 1 : 0;
      assume &= 31;
      hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds, st, sd, src, src_ld, buf[gg & 1], n,
                         t->rows_out, t->rows_cap, ctrl, gen, gg, tiles[t->flip], tiles[t->flip ^ 1], (double)(gg + 1),
                         t->limit, t->ray_offset, keep,
                         t->publish_in_kernel ? t->mirror_dev : (HostMirror*)nullptr, epoch, b,
                         b + 1 == t->batch ? 1 : 0, assume, (const PlanDev*)t->plan_dev);
      t->flip ^= 1;
      continue;
    }
    // dense-mode hint of the previous trace for this generation (the kernel reads the generation's
    // ray count on the device and checks the assumption tile by tile)
    int assume = 0;
    if (t->use_hints && gg < (int)s->hint_mode.size()) assume = s->hint_mode[gg];
    // (a generation whose dense hint was refuted lately keeps compacting for a while: see hint_rest)
    if (assume && gg < (int)s->hint_rest.size() && s->hint_rest[gg] > 0) assume = 0;
    // Mode 4 (sparse loss): last time every ray of this generation was recorded and all but a few of them -- absorbed
    // ones -- were carried on.  Compacting those few away costs every tile a look-back, and the tiles that hold such a
    // ray are as a rule the slow ones (a ray that misses the part it was expected to hit visits the parts behind it):
    // with 1 tile in 60 slow and 1280 in flight, every tile waits for a straggler (profiles/r4/lookback_stragglers.txt).
    // Such a generation runs dense instead, keeping its absorbed rays the way upstream does (PRT_TRACE_KEEP_ABSORBED
    // for this launch only: _pyrayt.py:415-428 carries them, direction zeroed, and drops them a generation later);
    // the next generation finds them dead on arrival, records nothing for them and drops them when it compacts.
    int keep = keep_absorbed;
    if (assume == 4) {
      if (t->flags & PRT_TRACE_NO_SPARSE_KEEP) assume = 0;
      else { assume = 1; keep = 1; }
    }
    // Modes 5 / 6: the generation behind one that keeps its absorbed rays and loses none of its own (5: carries the
    // live ones on, 6: carries none) takes its offsets from the dead list its predecessor wrote (k_generation) --
    // no look-back for a handful of dead rays.  Only right behind such a launch of this attempt (generation 0 keeps
    // rays without listing them); behind anything else nothing dead arrives and the plain dense forms apply.
    if (assume == 5 || assume == 6) {
      const bool kept_before = gg > 0 && t->launch_mode[gg - 1] == 4;
      if (!kept_before) assume -= 4;
      else if (gg == 1) assume = 0;
    }
    t->launch_mode[gg] = (char)(keep && !keep_absorbed ? 4 : assume);
    s->dense_launches += assume ? 1 : 0;
    s->sparse_keep_launches += keep && !keep_absorbed ? 1 : 0;
    // launches, not generations, alternate between the two status buffers, across traces too: every
    // launch works on the one the launch before it left clean and cleans the other (k_generation)
    hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds, st, sd, src, src_ld, buf[gg & 1], n,
                       t->rows_out, t->rows_cap, ctrl, gen, gg, tiles[t->flip], tiles[t->flip ^ 1], (double)(gg + 1),
                       t->limit, t->ray_offset, keep,
                       t->publish_in_kernel ? t->mirror_dev : (HostMirror*)nullptr, epoch, b,
                       b + 1 == t->batch ? 1 : 0, assume, (const PlanDev*)nullptr);
    t->flip ^= 1;
    if (b == 0) HP(4);
  }
  if (timed) {
    HIP_TRY(hipEventRecord(t->ev1, st));
    t->timing_pending = true;
  }
  // Behind the batch, only when the generation kernels did not tell the host themselves: a one-block
  // kernel that does (and re-arms the control words if the batch turns out to end the trace).  In
  // the usual case nothing runs behind the batch: the status buffers recycle each other, a
  // generation slot is always written by the launch before the one that reads it, and the error
  // word of a trace that succeeded is still zero.
  if (!t->publish_in_kernel) {
    hipLaunchKernelGGL(k_fused_reinit, dim3(1), dim3(1024), 0, st, ctrl, gen, t->limit + 1, t->g + t->batch,
                       t->limit, tiles[0], (int64_t)blocks_for(n), n, t->test_stall, t->mirror_dev, epoch,
                       t->g, t->batch + 1);
  }
  if (t->busy1) {
    HIP_TRY(hipEventRecord(t->busy1, st));
    // (both events of the pair are on the stream now -- busy0 was recorded by this attempt's first batch: only such a
    // job enters the merged intervals, whatever path a job that launches nothing took)
    if (t->busy_recorded && t->busy0) *t->busy_recorded = 1;
  }
  HIP_TRY(hipGetLastError());
  HP(5);
  t->launched = true;
  return PRT_OK;
}

// start an attempt of the ticket's trace on the fused path: control words, hints, first batch
static int fused_start(prt_scene* s, DeviceCopy* c, TraceTicket* t) {
  const int64_t n = t->n;
  const TraceLayout l = trace_layout(n);
  char* w = t->w;
  const int n_slots = t->limit + 1;
  const int keep_absorbed = (t->flags & PRT_TRACE_KEEP_ABSORBED) ? 1 : 0;
  t->test_stall = (t->flags & PRT_TRACE_TEST_STALL) ? 1 : 0;
  // A trace leaves the control words (generation slots, tile status buffers, error word) as the next
  // trace of the same shape needs them (see the launch loop); only a first trace, one with another
  // workspace / ray count / limit, or one behind a trace that failed clears them here.
  if (t->user == 0) t->user = g_next_user.fetch_add(1);
  const bool others = workspace_taken_over(w, t->user);
  if (others || !(t->ready_workspace == w && t->ready_n == n && t->ready_slots == n_slots &&
                  t->ready_stall == t->test_stall)) {
    hipLaunchKernelGGL(k_fused_init, dim3(64), dim3(256), 0, t->st, (FusedCtrl*)(w + l.fctrl), (GenCtrl*)(w + l.gen),
                       n_slots, (unsigned long long*)(w + l.tiles_a), (int64_t)blocks_for(n), n, t->test_stall);
    t->flip = 0;
  }
  t->ready_workspace = nullptr;
  // the generation kernels tell the host themselves unless an error can still be raised after the
  // last tile has its totals (only PRT_ERR_UNTRACABLE, at store time) or an experiment kernel runs
  // (the stall test hook raises its fake error from a tile that no successor waits for: same path)
  t->publish_in_kernel = !s->has_untracable && !t->test_stall && !(t->flags & PRT_TRACE_PUBLISH_KERNEL);
  // Hints from the previous trace of this scene with this many rays (PRT_TRACE_NO_HINTS turns them off).
  // After a miss the hints rest for 2, 4, 8 ... 64 traces (a caller that alternates between ray sets of
  // different shapes must not pay a repeat every time).
  for (int& rest : s->hint_rest) rest -= rest > 0 ? 1 : 0;  // (counted in traces of this scene)
  bool allow_hints = t->allow_hints;
  if (allow_hints && s->hint_holdoff > 0) {
    s->hint_holdoff -= 1;
    allow_hints = false;
  }
  // (the hints say how the scene treated the previous ray set -- which generations lost no ray -- and each
  // tile checks them on its own rays, so they serve a ray set of another SIZE as well: a design loop that
  // changes its ray count from call to call keeps them)
  t->use_hints = allow_hints && s->hint_n >= 0 && s->hint_keep_absorbed == keep_absorbed && !t->test_stall;
  if (t->plan_active) {
    // the plan as the kernels read it (once per change), clean sums (after an attempt that was not folded), and the
    // plan's own hints: those of the last trace of this scene under a plan that stores the same rows
    if (t->plan_dirty) {
      hipLaunchKernelGGL(k_plan_set, dim3(1), dim3(1), 0, t->st, t->plan_dev, *t->plan_host);
      t->plan_dirty = false;
    }
    if (t->plan_host->n_groups > 0) {
      if (t->sink_unclean) HIP_TRY(hipMemsetAsync(t->sink_slots, 0, t->sink_slot_bytes, t->st));
      t->sink_unclean = true;  // (until plan_fold has run behind this attempt)
    }
    bool plan_hints = t->allow_hints && s->plan_hint_key == t->plan_key && !t->test_stall;
    if (plan_hints && (++s->plan_traces & 31) == 0) {  // (every 32nd trace under the plan: the kept-rays bits are measured anew)
      for (char& m : s->plan_hint_mode) if (m & 32) m = 0;
    }
    if (plan_hints && s->plan_hint_rest > 0) {
      s->plan_hint_rest -= 1;
      plan_hints = false;
    }
    t->use_hints = plan_hints;
    allow_hints = false;  // (no per-tile records, no dead lists under a plan)
  }
  t->g = 0;
  t->n_seen = 0;
  t->total_rows = 0;
  // Generations are launched blind, a batch at a time, and the host looks at the counts once per
  // batch.  A scene traced before most likely runs as many generations as last time: launching
  // exactly that many first means neither a launch that finds no rays nor a second round trip.
  int want = kGenerationBatch;
  if (s->last_generations > 0) want = std::min(s->last_generations, kMaxBatch);
  t->batch = std::min(want, t->limit);
  HP(1);
  return fused_launch_batch(s, c, t);
}

// wait for the attempt's batch, look at its counts, enqueue further batches until the trace is over.
// Returns the number of rows, or an error (the internal ones included: the caller repeats the attempt).
static int64_t fused_finish(prt_scene* s, DeviceCopy* c, TraceTicket* t, int64_t* rows_per_generation) {
  const int keep_absorbed = (t->flags & PRT_TRACE_KEEP_ABSORBED) ? 1 : 0;
  int error = 0;
  bool done = false;
  while (true) {
    int rc = await_epoch(t, t->epoch);
    if (rc) return rc;
    HP(6);
    const GenCtrl* host_gen = t->mirror->gen;
    t->stats[3] += t->batch;
    error = t->mirror->error;
    if (error) break;
    for (int b = 0; b < t->batch; ++b) {
      if (host_gen[b].n_in == 0) { done = true; break; }
      t->stats[0] += 1;
      t->stats[1] += (double)host_gen[b].n_in;
      t->stats[4] += (double)host_gen[b].n_live;
      t->stats[5] += (double)host_gen[b].n_carry;
      rows_per_generation[t->g + b] = host_gen[b].n_live;
      t->total_rows += host_gen[b].n_live;
      const bool all_live = host_gen[b].n_live == host_gen[b].n_in;
      const int64_t lost = host_gen[b].n_in - host_gen[b].n_carry;
      // (bit 0, sparse loss: every ray recorded, at most 1 in 64 absorbed; bit 1: ... at least one such ray in 128
      // tiles, enough to stall a look-back; bit 2: ... few enough for the generation behind to read them off the dead
      // list -- a tile reads the whole list, and from a few hundred entries on that costs what a look-back without
      // stragglers costs: ab_round4.txt, "mode 7")
      t->seen_sparse[t->n_seen] = !(all_live && host_gen[b].n_carry > 0 && lost > 0 && lost * 64 <= host_gen[b].n_in) ? 0
                                  : (char)(1 | (lost * 128 * PRT_BLOCK >= host_gen[b].n_in ? 2 : 0) | (lost <= 256 ? 4 : 0));
      // (bits 3, 4: more than 1 ray in 32 / in 16 arrived dead or hit nothing.  The first keeps the generation before
      // from starting to keep its absorbed rays -- this one would carry too many dead lanes --, the second makes one
      // that does keep them stop: it is keeping too many by now.  Two thresholds, a sparse loss apart: no flip-flop.)
      if ((host_gen[b].n_in - host_gen[b].n_live) * 32 > host_gen[b].n_in) t->seen_sparse[t->n_seen] |= 8;
      if ((host_gen[b].n_in - host_gen[b].n_live) * 16 > host_gen[b].n_in) t->seen_sparse[t->n_seen] |= 16;
      t->seen_mode[t->n_seen++] = all_live && lost == 0 ? 1 : all_live && host_gen[b].n_carry == 0 ? 2 : 0;
      if (t->plan_active) {
        // under a record plan (n_live counts the rows STORED): a generation is dense when its rays all go on or none
        // does, and all of them store a row or none does (k_generation<.., PLAN>: bits 0..3 | bit 4)
        const int carry_form = lost == 0 ? 1 : host_gen[b].n_carry == 0 ? 2 : 0;
        const bool rec_all = all_live, rec_none = host_gen[b].n_live == 0;
        char mode = (char)(carry_form && (rec_all || rec_none) ? (carry_form | (rec_none ? 16 : 0)) : 0);
        if (!t->plan_host->store_rows && !keep_absorbed) {
          // (no rows to place: a generation that loses only a few absorbed rays keeps them next time and stays dense;
          // one launched that way is seen to carry everything and stays that way -- every 64th trace of the scene under
          // a plan is launched without the bit, so that a generation that has begun to absorb in numbers is found out)
          const bool kept = (t->launch_mode[t->g + b] & 32) != 0;
          const bool sparse = lost > 0 && host_gen[b].n_carry > 0 && lost * 64 <= host_gen[b].n_in;
          if ((kept && carry_form == 1) || (!kept && sparse)) mode = (char)(1 | 16 | 32);
        }
        t->seen_mode[t->n_seen - 1] = mode;
      }
    }
    if (!done && host_gen[t->batch].n_in == 0) done = true;
    t->g += t->batch;
    if (done || t->g >= t->limit) break;
    t->batch = std::min(kGenerationBatch, t->limit - t->g);
    rc = fused_launch_batch(s, c, t);
    if (rc) return rc;
  }
  t->launched = false;
  if (t->plan_active) {
    // a trace under a record plan keeps its own hints and touches none of the scene's others
    if (error == PRT_ERR_SPECULATION || error == PRT_ERR_FULL_ROWS) return error;
    if (error == PRT_ERR_ROWS_CAP && t->use_hints) return PRT_ERR_SPECULATION;  // (too small for the hint, perhaps not for the rows)
    if (error == PRT_ERR_STALL) return PRT_ERR_STALL;
    if (error) return trace_error(error);
    if (!t->publish_in_kernel) t->flip = 0;
    s->last_generations = (int)t->stats[0];
    s->plan_hint_mode.assign(t->seen_mode, t->seen_mode + t->n_seen);
    s->plan_hint_key = t->plan_key;
    if (t->use_hints) s->plan_hint_misses_in_a_row = 0;
    t->ready_workspace = t->w;
    t->ready_n = t->n;
    t->ready_slots = t->limit + 1;
    t->ready_stall = t->test_stall;
    return t->total_rows;
  }
  if (!error) {
    // Which generations keep their absorbed rays next time (mode 4, see the launch loop).  It moves the compaction to
    // the generation behind: free when that one compacts anyway, worth it when the loss is dense enough for its
    // stragglers to hold up most of the tiles in flight (config 3: one lost ray in 61 tiles, -6.6 % on the trace), a
    // loss when it turns a dense generation into a compacting one for a handful of rays (config 2: one in 434 tiles,
    // +3.4 %; profiles/r4/ab_round4.txt).
    const bool sparse_ok = !keep_absorbed && !(t->flags & PRT_TRACE_NO_SPARSE_KEEP);
    // (raw: what the counts say -- 1 every ray recorded and carried, 2 every ray recorded, none carried, 0 neither)
    char raw_next = t->n_seen ? t->seen_mode[0] : 0;
    for (int g = 0; g < t->n_seen; ++g) {
      const char raw = raw_next, was = g < (int)s->hint_mode.size() && t->use_hints ? s->hint_mode[g] : 0;
      raw_next = g + 1 < t->n_seen ? t->seen_mode[g + 1] : 0;
      char mode = raw;
      if (t->launch_mode[g] == 4) {
        // (kept its absorbed rays: how many there were cannot be told from its own counts -- the generation behind
        // tells: when a thirty-second of what it received is dead, this one goes back to compacting and is judged anew)
        const bool flooded = g + 1 < t->n_seen && (t->seen_sparse[g + 1] & 16);
        mode = raw == 1 && !flooded ? 4 : 0;
      } else if (sparse_ok && (t->seen_sparse[g] & 1) && g + 1 < t->n_seen && !(t->seen_sparse[g + 1] & 8) &&
                 ((t->seen_sparse[g] & 2) || raw_next == 0 || (t->seen_sparse[g + 1] & 1) || (g > 0 && (t->seen_sparse[g] & 4)))) {
        // (g > 0 and few: whatever the generation behind looks like, it can take the dead list -- modes 5 / 6)
        mode = 4;
      } else if (sparse_ok && was == 4 && raw == 1) {
        mode = 4;  // nothing absorbed this time: the form that covers both stays
      }
      // The generation behind one that keeps its absorbed rays finds them dead among its own.  If it loses none of
      // its own it runs on the dead list (5 / 6; generation 0 writes no list); otherwise it compacts.
      if (g > 0 && t->seen_mode[g - 1] == 4) {
        // (the list is worth reading when it is short: known when the generation before was seen without keeping)
        const bool listed = g > 1 && t->launch_mode[g - 1] != 4 && (t->seen_sparse[g - 1] & 4);
        if (t->launch_mode[g] == 5 || t->launch_mode[g] == 6) mode = t->launch_mode[g];  // held, tile by tile
        else if (listed && raw == 1) mode = 5;
        else if (listed && raw == 2) mode = 6;
        else mode = 0;
      }
      t->seen_mode[g] = mode;
    }
  }
  if (error == PRT_ERR_SPECULATION || error == PRT_ERR_FULL_ROWS) return error;
  // a record block that looked too small to a generation launched on a hint may only have been too small
  // for the hint: the caller repeats without hints before it reports it
  if (error == PRT_ERR_ROWS_CAP && t->use_hints) return PRT_ERR_SPECULATION;
  if (error == PRT_ERR_STALL) return PRT_ERR_STALL;
  if (error) return trace_error(error);
  if (!t->publish_in_kernel) t->flip = 0;  // (k_fused_reinit cleared buffer 0; the next launch cleans buffer 1)
  s->last_generations = (int)t->stats[0];
  s->hint_n = t->n;
  s->hint_keep_absorbed = keep_absorbed;
  if (!s->missed_mode.empty()) {
    // This trace is the repeat of an attempt whose dense hints did not hold: the generations that were offered a
    // hint and turned out otherwise are the ones whose rays are lost differently from trace to trace (a ray set
    // that loses a near-axial ray in one generation where the previous one lost none).  Such a generation is not
    // offered its dense hint for the next 32, 64 ... 4096 traces (it compacts, by look-back or on its per-tile
    // record); the hints of the other generations were not refuted and stay in use -- a loop that alternates
    // between such ray sets pays one repeat per rest, not one every other trace.
    bool found = false;
    if (s->hint_rest.size() < s->missed_mode.size()) { s->hint_rest.resize(s->missed_mode.size(), 0); s->hint_rest_span.resize(s->missed_mode.size(), 0); }
    for (size_t g = 0; g < s->missed_mode.size(); ++g) {
      const char now = g < (size_t)t->n_seen ? t->seen_mode[g] : 0;
      const char offered = s->missed_mode[g];
      // (offered a plain dense form, found to absorb a few rays or to sit behind a generation that does: the hint it
      // gets now covers both cases, nothing to rest)
      if ((offered == 1 && (now == 4 || now == 5)) || (offered == 2 && now == 6)) { found = true; continue; }
      // (offered a form that also covers what the repeat saw: not the one that missed)
      if ((offered == 4 && now == 1) || (offered == 5 && now == 1) || (offered == 6 && now == 2)) continue;
      if (offered != 0 && offered != now) {
        s->hint_rest_span[g] = s->hint_rest_span[g] ? std::min(s->hint_rest_span[g] * 2, 4096) : 32;
        s->hint_rest[g] = s->hint_rest_span[g];
        found = true;
      }
    }
    if (!found) {
      // Nobody looks different in the repeat: if generations ran on a dead list, the list it is (more tiles kept
      // rays than it holds -- a reader cannot tell from the counts of a trace without hints): those rest.
      for (size_t g = 0; g < s->missed_mode.size(); ++g) {
        if (s->missed_mode[g] != 5 && s->missed_mode[g] != 6) continue;
        s->hint_rest_span[g] = s->hint_rest_span[g] ? std::min(s->hint_rest_span[g] * 2, 4096) : 32;
        s->hint_rest[g] = s->hint_rest_span[g];
        found = true;
      }
    }
    if (found) { s->hint_holdoff = 0; s->hint_misses_in_a_row = 0; }  // (the culprit rests by itself)
    s->missed_mode.clear();
  }
  s->hint_mode.assign(t->seen_mode, t->seen_mode + t->n_seen);
  if (t->use_hints) s->hint_misses_in_a_row = 0;
  // the control words are as a next trace of this shape needs them (see the launch loop)
  t->ready_workspace = t->w;
  t->ready_n = t->n;
  t->ready_slots = t->limit + 1;
  t->ready_stall = t->test_stall;
  return t->total_rows;
}

static void reset_stats(prt_scene* s, TraceTicket* t, int variant) {
  for (double& v : t->stats) v = 0;
  t->stats[6] = (double)s->lookback_fallbacks;
  t->stats[7] = variant;
}

extern "C" int prt_trace_begin(prt_scene* s, int device, int ticket, const double* rays, int64_t n, int64_t ld,
                               int generation_limit, double ray_offset, double* rows_out, int64_t rows_cap,
                               void* workspace, int flags, void* stream) {
  HP(0);
  DeviceCopy* c;
  int rc = on_device(s, device, &c, true);
  if (rc) return rc;
  if (ticket < 0 || ticket >= PRT_TRACE_TICKETS) return fail(PRT_ERR_ARG, "ticket out of range");
  TraceTicket* t = &c->ticket[ticket];
  if (t->active) return fail(PRT_ERR_ARG, "this ticket has a trace in flight (prt_trace_end it first)");
  if (n < 0 || ld < n || generation_limit < 0 || generation_limit > kMaxGenerationSlots ||
      rows_cap < 0 || !workspace || (n && !rays) || (rows_cap && !rows_out))
    return fail(PRT_ERR_ARG, "bad buffers (generation_limit must be <= 1024)");
  if (n >= (1ll << 31)) return fail(PRT_ERR_ARG, "at most 2^31-1 rays per call");
  for (int k = 0; k < PRT_TRACE_TICKETS; ++k)
    if (k != ticket && c->ticket[k].active && n && c->ticket[k].n &&
        (c->ticket[k].w == (char*)workspace || c->ticket[k].rows_out == rows_out))
      return fail(PRT_ERR_ARG, "traces in flight together need their own workspace and record block");
  rc = ticket_resources(*t);
  if (rc) return rc;
  rc = settle_timing(t);  // events of the ticket's previous trace, before they are recorded again
  if (rc) return rc;
  t->rays = rays; t->n = n; t->ld = ld; t->limit = generation_limit; t->ray_offset = ray_offset;
  t->rows_out = rows_out; t->rows_cap = rows_cap; t->w = (char*)workspace; t->flags = flags;
  t->st = (hipStream_t)stream;
  // a scene update whose copy may still be on its way (prt_scene_update): this stream waits for it, once
  if (c->update_serial != 0 && (t->update_seen != c->update_serial || t->update_seen_stream != t->st)) {
    HIP_TRY(hipStreamWaitEvent(t->st, c->update_event, 0));
    t->update_seen = c->update_serial;
    t->update_seen_stream = t->st;
  }
  t->traced = true;
  t->launched = false;
  t->allow_hints = !(flags & PRT_TRACE_NO_HINTS);
  t->compact = !s->full_rows && !(flags & PRT_TRACE_FULL_ROWS);
  t->active = true;
  if (n == 0 || generation_limit == 0) { reset_stats(s, t, PRT_VARIANT_FUSED); return PRT_OK; }
  if (t->plan_active) {
    if (flags & (PRT_TRACE_UNFUSED | PRT_TRACE_COUNT_PATHS)) {
      t->active = false;
      return fail(PRT_ERR_ARG, "a trace under a record plan runs on the fused path only (PRT_TRACE_UNFUSED / COUNT_PATHS)");
    }
    if (generation_limit > t->plan_host->limit) {
      t->active = false;
      return fail(PRT_ERR_ARG, "generation_limit exceeds the record plan's");
    }
    if (t->plan_host->store_rows && rows_cap > 0 && !rows_out) { t->active = false; return fail(PRT_ERR_ARG, "bad buffers"); }
  }
  if (flags & PRT_TRACE_COUNT_PATHS) t->flags |= PRT_TRACE_UNFUSED;  // the counting nearest-hit kernel lives on that path
  if (t->flags & PRT_TRACE_UNFUSED) {  // host round trip per generation: everything happens in prt_trace_end
    reset_stats(s, t, s->options.hit_lanes > 1 ? PRT_VARIANT_KLANES : PRT_VARIANT_UNFUSED);
    return PRT_OK;
  }
  reset_stats(s, t, PRT_VARIANT_FUSED);
  rc = fused_start(s, c, t);
  if (rc) t->active = false;
  return rc;
}

extern "C" int64_t prt_trace_end(prt_scene* s, int device, int ticket, int64_t* rows_per_generation) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c, true);
  if (rc) return rc;
  if (ticket < 0 || ticket >= PRT_TRACE_TICKETS || !rows_per_generation)
    return fail(PRT_ERR_ARG, "bad ticket / null rows_per_generation");
  TraceTicket* t = &c->ticket[ticket];
  if (!t->active) return fail(PRT_ERR_ARG, "no trace in flight on this ticket");
  t->active = false;
  s->stats_device = device;
  s->stats_ticket = ticket;
  for (int g = 0; g < t->limit; ++g) rows_per_generation[g] = 0;
  if (t->n == 0 || t->limit == 0) {
    if (t->plan_active && t->plan_host->n_groups > 0)  // (no ray, no row: the sums of this trace are zeros)
      HIP_TRY(hipMemsetAsync(t->sums_out, 0, (size_t)t->plan_host->limit * t->plan_host->n_groups * SINK_STATS * sizeof(double), t->st));
    return 0;
  }
  int64_t rc64 = 0;
  if (t->flags & PRT_TRACE_UNFUSED) {
    rc64 = trace_unfused(s, c, t, rows_per_generation);
  } else {
    for (int attempt = 0;; ++attempt) {
      rc64 = fused_finish(s, c, t, rows_per_generation);
      if ((rc64 != PRT_ERR_SPECULATION && rc64 != PRT_ERR_FULL_ROWS) || attempt == 4) break;
      if (rc64 == PRT_ERR_SPECULATION && t->plan_active) {
        // (the plan's own hints: forgotten, learnt again from the repeat, rested 2, 4 .. 64 traces after misses in a row)
        s->plan_misses += 1;
        s->plan_hint_misses_in_a_row = std::min(s->plan_hint_misses_in_a_row + 1, 6);
        s->plan_hint_rest = 1 << s->plan_hint_misses_in_a_row;
        s->plan_hint_key = 0;
        t->allow_hints = false;
      } else if (rc64 == PRT_ERR_SPECULATION) {
        // a generation assumed dense was not (the rays or the scene changed since the hints were taken):
        // nothing of this attempt is kept; run again without assumptions, which also renews the hints
        s->speculation_misses += 1;
        s->missed_mode = s->hint_mode;  // (which generations were offered what: compared with the repeat's outcome)
        s->hint_misses_in_a_row = std::min(s->hint_misses_in_a_row + 1, 6);
        s->hint_holdoff = 1 << s->hint_misses_in_a_row;
        s->hint_n = -1;
        t->allow_hints = false;
      } else {
        // a ray needs the state rows the compact form leaves out (homogeneous w other than 1 / +0, or a
        // ray set that does not start at generation 0): this scene traces with all 13 rows from now on
        s->full_rows = true;
        s->full_rows_fallbacks += 1;
        t->compact = false;
      }
      for (int g = 0; g < t->limit; ++g) rows_per_generation[g] = 0;
      rc = settle_timing(t);
      if (rc) return rc;
      reset_stats(s, t, PRT_VARIANT_FUSED);
      rc = fused_start(s, c, t);
      if (rc) return rc;
    }
    if (rc64 == PRT_ERR_SPECULATION || rc64 == PRT_ERR_FULL_ROWS)
      rc64 = fail(PRT_ERR_HIP, "trace kept failing its own assumptions");
    if (rc64 == PRT_ERR_STALL && t->plan_active)
      rc64 = fail(PRT_ERR_HIP, "the look-back of a trace under a record plan gave up (no three-kernel path under a plan)");
    if (rc64 == PRT_ERR_STALL) {  // never observed outside the test hook; see lookback()
      for (int g = 0; g < t->limit; ++g) rows_per_generation[g] = 0;
      rc = settle_timing(t);
      if (rc) return rc;
      s->lookback_fallbacks += 1;  // telemetry: a box that falls back silently would just look 2x slow
      reset_stats(s, t, PRT_VARIANT_UNFUSED);
      rc64 = trace_unfused(s, c, t, rows_per_generation);
    }
  }
  if (rc64 >= 0 && t->plan_active && !(t->flags & PRT_TRACE_UNFUSED)) {
    rc = plan_fold(t);  // (the sums of this trace, on its stream, behind its last kernel)
    if (rc) return rc;
  }
  if (rc64 >= 0 && (t->flags & PRT_TRACE_SYNC)) HIP_TRY(hipStreamSynchronize(t->st));
#ifdef PRT_HOST_PROFILE
  HP(7);
  for (int k = 1; k < 8; ++k) g_hp[k] += g_hp_t[k] - g_hp_t[0];
  g_hp_n += 1;
#endif
  return rc64;
}

extern "C" int64_t prt_trace(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                             int generation_limit, double ray_offset, double* rows_out,
                             int64_t rows_cap, int64_t* rows_per_generation, void* workspace,
                             int flags, void* stream) {
  if (!rows_per_generation) return fail(PRT_ERR_ARG, "rows_per_generation is null");
  const int rc = prt_trace_begin(s, device, 0, rays, n, ld, generation_limit, ray_offset, rows_out, rows_cap,
                                 workspace, flags, stream);
  if (rc) return rc;
  return prt_trace_end(s, device, 0, rows_per_generation);
}

// A sequence of traces of one scene, `depth` of them in flight (ticket k % depth, its workspace, its
// stream): the loop DeviceScene.trace_many runs in Python, as one call.  (Measured with a host pause between
// collecting a trace and starting the next, profiles/r3/batch_issue.txt: the pace of the host is not what
// bounds overlapped traces -- a tight Python loop reaches the same step time down to 125k rays.)
static_assert(sizeof(prt_trace_job) == 56, "prt_trace_job is part of the ABI (engine.JOB_DTYPE, INTEGRATION.md)");
extern "C" int64_t prt_trace_batch(prt_scene* s, int device, prt_trace_job* jobs, int64_t count, int generation_limit,
                                   double ray_offset, int depth, void* const* workspaces, void* const* streams,
                                   int flags) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c, true);
  if (rc) return rc;
  if (count < 0 || (count && !jobs) || depth < 1 || depth > PRT_TRACE_TICKETS || !workspaces)
    return fail(PRT_ERR_ARG, "bad job list / depth out of range (1..PRT_TRACE_TICKETS) / null workspaces");
  for (int64_t k = 0; k < count; ++k) {
    if (!jobs[k].rows_per_generation) return fail(PRT_ERR_ARG, "a job has no rows_per_generation");
    jobs[k].total = 0;
  }
  for (int k = 0; k < depth; ++k)
    if (c->ticket[k].active) return fail(PRT_ERR_ARG, "a ticket this batch needs has a trace in flight");
  // PRT_TRACE_BUSY: every job gets a pair of HIP events of its own around its launches, on its stream; behind the
  // batch the intervals are merged -- how long the device had at least one of the batch's traces in flight
  const bool busy = (flags & PRT_TRACE_BUSY) != 0 && count > 0;
  if (busy) {
    while ((int64_t)c->busy_events.size() < 2 * count) {
      hipEvent_t e = nullptr;
      HIP_TRY(hipEventCreate(&e));
      c->busy_events.push_back(e);
    }
    for (double& v : c->busy) v = 0;
    c->busy_recorded.assign((size_t)count, 0);
  }
  int64_t sum = 0, first_error = 0;
  std::string message;
  for (int64_t k = 0; k < count + depth; ++k) {
    const int lane = (int)(k % depth);
    if (k >= depth && c->ticket[lane].active) {  // the ticket about to be reused: collect its trace first
      prt_trace_job& done = jobs[k - depth];
      done.total = prt_trace_end(s, device, lane, done.rows_per_generation);
      if (done.total < 0 && !first_error) { first_error = done.total; message = g_error; }
      if (done.total > 0) sum += done.total;
    }
    if (k < count && !first_error) {  // (after an error nothing new is started; what is in flight is collected)
      const prt_trace_job& job = jobs[k];
      c->ticket[lane].busy0 = busy ? c->busy_events[2 * k] : nullptr;
      c->ticket[lane].busy1 = busy ? c->busy_events[2 * k + 1] : nullptr;
      c->ticket[lane].busy_recorded = busy ? &c->busy_recorded[(size_t)k] : nullptr;
      rc = prt_trace_begin(s, device, lane, job.rays, job.n, job.ld, generation_limit, ray_offset, job.rows_out,
                           job.rows_cap, workspaces[lane], flags, streams ? streams[lane] : nullptr);
      if (rc) { jobs[k].total = rc; first_error = rc; message = g_error; }
    }
  }
  for (int k = 0; k < depth; ++k) { c->ticket[k].busy0 = c->ticket[k].busy1 = nullptr; c->ticket[k].busy_recorded = nullptr; }
  if (first_error) return fail((int)first_error, message.c_str());
  if (busy) {
    // intervals relative to the first job's start (a float of milliseconds resolves ~2 ns over a 30 ms region)
    std::vector<std::pair<double, double>> spans;
    int64_t base = -1;  // the first job that launched anything: every interval is measured from its start
    for (int64_t k = 0; k < count; ++k) {
      if (!c->busy_recorded[(size_t)k]) continue;  // (no launch of this job was bracketed: an empty job, the three-kernel path)
      if (base < 0) base = k;
      HIP_TRY(hipEventSynchronize(c->busy_events[2 * k + 1]));
      float t0 = 0, t1 = 0;
      HIP_TRY(hipEventElapsedTime(&t0, c->busy_events[2 * base], c->busy_events[2 * k]));
      HIP_TRY(hipEventElapsedTime(&t1, c->busy_events[2 * base], c->busy_events[2 * k + 1]));
      spans.emplace_back((double)t0, (double)t1);
    }
    std::sort(spans.begin(), spans.end());
    double merged = 0, each = 0, open_from = 0, open_to = 0;
    bool open = false;
    for (const auto& sp : spans) {
      each += sp.second - sp.first;
      if (open && sp.first <= open_to) { open_to = std::max(open_to, sp.second); continue; }
      if (open) merged += open_to - open_from;
      open = true; open_from = sp.first; open_to = sp.second;
    }
    if (open) merged += open_to - open_from;
    c->busy[0] = merged;
    c->busy[1] = each;
    c->busy[2] = (double)spans.size();
    double first = 0, last = 0;
    for (size_t k = 0; k < spans.size(); ++k) {
      first = k ? std::min(first, spans[k].first) : spans[k].first;
      last = k ? std::max(last, spans[k].second) : spans[k].second;
    }
    c->busy[3] = last - first;
  }
  return sum;
}

extern "C" int prt_trace_batch_busy(const prt_scene* s, int device, double* out4) {
  if (!s || !out4) return fail(PRT_ERR_ARG, "null argument");
  if (device < 0 || device >= (int)s->per_device.size()) return fail(PRT_ERR_ARG, "no trace of this scene ran on that device");
  for (int k = 0; k < 4; ++k) out4[k] = s->per_device[device].busy[k];
  return PRT_OK;
}


extern "C" int prt_trace_telemetry(const prt_scene* s, int64_t* out12) {
  int64_t* out8 = out12;
  if (!s || !out8) return fail(PRT_ERR_ARG, "null argument");
  out12[8] = s->plan_launches;  // generation launches under a record plan (slots 8 / 9 counted the per-tile records retired in round 6)
  out12[9] = s->plan_misses;    // traces under a plan repeated because one of the plan's dense hints did not hold
  out12[10] = s->sparse_keep_launches;
  out12[11] = s->plan_dense_launches;  // generation launches under a record plan that ran dense
  out8[0] = s->lookback_fallbacks;
  out8[1] = s->speculation_misses;
  out8[2] = s->dense_launches;
  out8[3] = s->full_rows_fallbacks;
  for (int k = 0; k < 4; ++k) out8[4 + k] = s->path_counts[k];
  return PRT_OK;
}

extern "C" int prt_trace_stats(const prt_scene* s, double* out8) {
  if (!s || !out8) return fail(PRT_ERR_ARG, "null argument");
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  if (s->stats_device < 0 || s->stats_device >= (int)s->per_device.size()) return PRT_OK;  // nothing traced yet
  TraceTicket* t = &const_cast<prt_scene*>(s)->per_device[s->stats_device].ticket[s->stats_ticket];
  if (!t->active) {  // (begun again already: its events belong to the new trace; the time stays out)
    HIP_TRY(hipSetDevice(s->stats_device));
    int rc = settle_timing(t);  // the last batch's event time is collected on demand
    if (rc) return rc;
  }
  for (int k = 0; k < 8; ++k) out8[k] = t->stats[k];
  return PRT_OK;
}

#ifdef PRT_TIMING
// experiment build: the s_memtime stamps of generation 0's waves (tools/lookback_analysis.py)
extern "C" int prt_debug_wave_stamps(long long* out, int64_t count) {
  if (!out || count < 0 || count > 16384 * 4 * 8) return PRT_ERR_ARG;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)count * sizeof(long long)) != hipSuccess) return PRT_ERR_HIP;
  return PRT_OK;
}
#endif
