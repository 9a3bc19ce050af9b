// prt_abi_states.hpp -- extern "C" entry points of include/prt.h for the single states of the path: library / device
// queries, prt_intersect, prt_propagate, prt_generate_rays, prt_world_normals, prt_material_trace, prt_interact (with its
// small kernels) and, through prt_host_shade.hpp, the gather / scatter around a caller's own Material.trace().  Host
// code; included by prt_kernels.hip behind the kernels it launches.
#pragma once
// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" int prt_version(void) { return PRT_VERSION; }
extern "C" const char* prt_last_error(void) { return g_error.c_str(); }
extern "C" int prt_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + PRT_BLOCK - 1) / PRT_BLOCK); }
static inline size_t lds_bytes(int slots) { return (size_t)slots * PRT_BLOCK * 12; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// the fused kernel parks 8 float64 per lane behind the hit lists
static inline size_t lds_bytes_fused(int slots) {
  return (size_t)((3 * slots + 1) / 2 + 8) * PRT_BLOCK * sizeof(double);
}

static SceneDev trace_scene_dev(const prt_scene* s, const DeviceCopy* c) {
  SceneDev sd{c->prims, c->trace_code, (int)s->trace_program.code.size(), s->trace_program.lds_slots};
  for (size_t p = 0; p < s->dev_prims.size() && p < 64; ++p)
    if (s->dev_prims[p].mat_kind == MAT_ABSORBER) sd.absorber_mask |= 1ull << p;
  return sd;
}

// ---- nearest-hit kernel selection -----------------------------------------------------------------
// prt_scene_options.hit_lanes = 4 | 8 | 16 (surface-parallel: K lanes per ray, shuffle min-reduce)
// and / or .hit_staged (program staged in LDS); applies to prt_propagate and to the three-kernel trace
// (PRT_TRACE_UNFUSED).  Default: one ray per lane, steps through the scalar cache -- the measured
// winner.
struct HitVariant {
  int lanes = 1;
  bool staged = false;
};
static HitVariant hit_variant(const prt_scene* s) {
  HitVariant v;
  v.lanes = s->options.hit_lanes > 1 ? s->options.hit_lanes : 1;
  v.staged = s->options.hit_staged != 0;
  return v;
}

// n_bound: rays the grid must cover (the kernels read the exact count from ctrl when given);
// tile_counts: (live, carried) per PRT_BLOCK-ray tile, or null
static int launch_hit(const prt_scene* s, const DeviceCopy* c, const SceneDev& sd, hipStream_t st,
                      const double* rays, int64_t ld, const TraceCtrl* ctrl, int64_t n_fixed, int64_t n_bound,
                      double* hit_t, int32_t* hit_prim, int64_t* surf_out, int32_t* tile_counts,
                      int keep_absorbed, unsigned long long* paths = nullptr) {
  const HitVariant v = hit_variant(s);
  size_t lds = lds_bytes(sd.lds_slots);
  if (v.staged) lds = align_up(lds, 16) + (size_t)sd.n_instr * sizeof(DevInstr) + 8 * PRT_BLOCK;
  if (lds > kMaxLdsBytes) return fail(PRT_ERR_SCENE, "program too large to stage in LDS");
  if (v.lanes == 1) {
    auto kernel = v.staged ? k_hit<true, false> : (paths ? k_hit<false, true> : k_hit<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks_for(n_bound)), dim3(PRT_BLOCK), lds, st, sd, rays, ld, ctrl, n_fixed,
                       hit_t, hit_prim, surf_out, tile_counts, keep_absorbed, paths);
  } else {
    if (tile_counts) HIP_TRY(hipMemsetAsync(tile_counts, 0, (size_t)blocks_for(n_bound) * 2 * sizeof(int32_t), st));
    const int n_comp = (int)s->roots.size();
    const unsigned grid = (unsigned)((n_bound + PRT_BLOCK / v.lanes - 1) / (PRT_BLOCK / v.lanes));
#define PRT_LAUNCH_LANES(K, STAGED)                                                                          \
    hipLaunchKernelGGL((k_hit_lanes<K, STAGED>), dim3(grid), dim3(PRT_BLOCK), lds, st, sd,                    \
                       (const int32_t*)c->trace_component_first, n_comp, rays, ld, ctrl, n_fixed, hit_t,     \
                       hit_prim, surf_out, tile_counts, keep_absorbed)
    if (v.lanes == 4) { if (v.staged) PRT_LAUNCH_LANES(4, true); else PRT_LAUNCH_LANES(4, false); }
    else if (v.lanes == 8) { if (v.staged) PRT_LAUNCH_LANES(8, true); else PRT_LAUNCH_LANES(8, false); }
    else { if (v.staged) PRT_LAUNCH_LANES(16, true); else PRT_LAUNCH_LANES(16, false); }
#undef PRT_LAUNCH_LANES
  }
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_intersect(prt_scene* s, int device, int root, const double* rays, int64_t n,
                             int64_t ld, double* hits_out, int64_t* ids_out, int64_t ld_out,
                             void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (root < 0 || root >= (int)s->roots.size()) return fail(PRT_ERR_ARG, "bad component index");
  if (n < 0 || ld < n || ld_out < n || (n && (!rays || !hits_out || !ids_out)))
    return fail(PRT_ERR_ARG, "bad ray / output buffers");
  if (n == 0) return PRT_OK;
  const Program& p = s->component_programs[root];
  SceneDev sd{c->prims, c->component_code[root], (int)p.code.size(), p.lds_slots};
  hipLaunchKernelGGL(k_intersect, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds_bytes(p.lds_slots),
                     (hipStream_t)stream, sd, s->component_result[root], rays, ld, n, hits_out,
                     ids_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_propagate(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                             double* t_out, int64_t* surf_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld < n || (n && (!rays || !t_out || !surf_out)))
    return fail(PRT_ERR_ARG, "bad ray / output buffers");
  if (n == 0) return PRT_OK;
  SceneDev sd = trace_scene_dev(s, c);
  return launch_hit(s, c, sd, (hipStream_t)stream, rays, ld, (const TraceCtrl*)nullptr, n, n, t_out,
                    (int32_t*)nullptr, surf_out, (int32_t*)nullptr, 1);
}

extern "C" int prt_generate_rays(int device, const prt_source* source, int64_t n_total,
                                 int64_t first, int64_t count, int64_t id_first, double* rays_out,
                                 int64_t ld, int64_t col_offset, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  if (!source || source->kind < PRT_SRC_LINE || source->kind > PRT_SRC_LAMP)
    return fail(PRT_ERR_ARG, "unknown source kind");
  if (n_total < 0 || first < 0 || count < 0 || first + count > n_total || col_offset < 0 ||
      ld < col_offset + count || (count && !rays_out))
    return fail(PRT_ERR_ARG, "bad ray range / output buffer");
  if (count == 0) return PRT_OK;
  HIP_TRY(hipSetDevice(device));
  DevSource d;
  d.kind = source->kind;
  d.p0 = source->params[0]; d.p1 = source->params[1]; d.p2 = source->params[2];
  d.wavelength = source->wavelength;
  std::memcpy(d.world, source->world, sizeof(d.world));
  d.seed = source->seed;
  hipLaunchKernelGGL(k_source, dim3(blocks_for(count)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, d,
                     n_total, first, count, id_first, rays_out, ld, col_offset);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_world_normals(prt_scene* s, int device, int prim, const double* points,
                                 int64_t k, int64_t ld, double* normals_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (prim < 0 || prim >= (int)s->prims.size()) return fail(PRT_ERR_ARG, "bad primitive index");
  if (k < 0 || ld < k || (k && (!points || !normals_out))) return fail(PRT_ERR_ARG, "bad buffers");
  if (k == 0) return PRT_OK;
  hipLaunchKernelGGL(k_normals, dim3(blocks_for(k)), dim3(PRT_BLOCK), 0, (hipStream_t)stream,
                     c->prims + prim, points, ld, k, normals_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_material_trace(prt_scene* s, int device, int prim, double* rays, int64_t k,
                                  int64_t ld, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (prim < 0 || prim >= (int)s->prims.size()) return fail(PRT_ERR_ARG, "bad primitive index");
  if (s->dev_prims[prim].mat_kind == MAT_NONE || s->dev_prims[prim].mat_kind == MAT_HOST)
    return fail(PRT_ERR_UNTRACABLE, "surface " + std::to_string(s->prims[prim].surface_id) +
                                        " has a material without trace()");
  if (k < 0 || ld < k || (k && !rays)) return fail(PRT_ERR_ARG, "bad buffers");
  if (k == 0) return PRT_OK;
  hipLaunchKernelGGL(k_material_trace, dim3(blocks_for(k)), dim3(PRT_BLOCK), 0,
                     (hipStream_t)stream, c->prims + prim, rays, ld, k);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// workspace of prt_interact: ctrl | block counts (2 x int32) | block offsets (2 x int64) | hit_t
struct InteractLayout {
  size_t ctrl, counts, offsets, total;
};
static InteractLayout interact_layout(int64_t n) {
  const size_t nb = blocks_for(n) + 1;
  InteractLayout l;
  l.ctrl = 0;
  l.counts = align_up(sizeof(TraceCtrl), 256);
  l.offsets = l.counts + align_up(nb * 2 * sizeof(int32_t), 256);
  l.total = l.offsets + align_up(nb * 2 * sizeof(int64_t), 256);
  return l;
}

extern "C" int64_t prt_interact_workspace_bytes(int64_t n) {
  return (int64_t)interact_layout(n < 0 ? 0 : n).total;
}

// live / carried counts for prt_interact, where the hits come from the caller
__global__ void __launch_bounds__(PRT_BLOCK)
k_count(SceneDev scene, int n_prims, const double* __restrict__ rays, int64_t ld, int64_t n,
        const int64_t* __restrict__ surf, int32_t* __restrict__ block_counts) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  bool live = false;
  if (i < n) {
    const Ray8 r = load_ray8(rays, ld, i);
    live = is_live(r, prim_of_surface(scene.prims, n_prims, surf[i]));
  }
  __shared__ int s_live;
  if (threadIdx.x == 0) s_live = 0;
  __syncthreads();
  const int w = __popcll(__ballot(live));
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_live, w);
  __syncthreads();
  if (threadIdx.x == 0) {
    block_counts[2 * blockIdx.x] = s_live;
    block_counts[2 * blockIdx.x + 1] = s_live;  // the stepwise API keeps absorbed rays (Q3)
  }
}

__global__ void k_ctrl_init(TraceCtrl* ctrl, int64_t n, int64_t rows_cap) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    ctrl->n_cur = n; ctrl->n_live = 0; ctrl->n_carry = 0; ctrl->row_base = 0;
    ctrl->rows_cap = rows_cap; ctrl->error = 0; ctrl->pad = 0;
    for (int k = 0; k < 4; ++k) ctrl->paths[k] = 0;
  }
}

__global__ void k_interact_finish(const TraceCtrl* ctrl, int64_t* n_live_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *n_live_out = ctrl->error ? (int64_t)ctrl->error : ctrl->n_live;
}

extern "C" int prt_interact(prt_scene* s, int device, const double* rays_in, int64_t n,
                            int64_t ld_in, const double* t, const int64_t* surf, double* rays_out,
                            int64_t ld_out, int generation, int generation_limit, double ray_offset,
                            double* rows_out, int64_t ld_rows, int64_t* n_live_out, const double* shaded,
                            int64_t ld_shaded, void* workspace, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld_in < n || ld_out < n || ld_rows < n || !n_live_out || !workspace ||
      (n && (!rays_in || !t || !surf || !rays_out || !rows_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (shaded && ld_shaded < n) return fail(PRT_ERR_ARG, "shaded block narrower than the ray set");
  hipStream_t st = (hipStream_t)stream;
  const InteractLayout l = interact_layout(n);
  char* w = (char*)workspace;
  TraceCtrl* ctrl = (TraceCtrl*)(w + l.ctrl);
  int32_t* counts = (int32_t*)(w + l.counts);
  int64_t* offsets = (int64_t*)(w + l.offsets);
  hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, st, ctrl, n, ld_rows);
  if (n > 0) {
    SceneDev sd{c->prims, nullptr, (int)s->prims.size(), 0};
    hipLaunchKernelGGL(k_count, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, st, sd,
                       (int)s->prims.size(), rays_in, ld_in, n, surf, counts);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, counts, offsets, ctrl);
    const int relaunch = (generation + 1 != generation_limit) ? 1 : 0;
    hipLaunchKernelGGL(k_shade, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, st, sd, rays_in, ld_in,
                       (const TraceCtrl*)ctrl, n, t, (const int32_t*)nullptr, surf,
                       (const int64_t*)offsets, rays_out, ld_out, rows_out, ld_rows, (int64_t)0,
                       (double)(generation + 1), relaunch, ray_offset, 1, ctrl, shaded, ld_shaded);
  }
  hipLaunchKernelGGL(k_interact_finish, dim3(1), dim3(1), 0, st, (const TraceCtrl*)ctrl, n_live_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

#include "prt_host_shade.hpp"
