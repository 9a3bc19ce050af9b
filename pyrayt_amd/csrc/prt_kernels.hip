// prt_kernels.hip -- HIP kernels (gfx950 / CDNA4) + the C-ABI of include/prt.h.
//
// One translation unit, in this order:
//   prt_device.hpp   per-ray arithmetic: primitives, CSG nodes, normals, shading, the step interpreter
//   prt_scene.hpp    host side: scene object, scene compiler, per-device upload
//   (this file)      trace kernels: k_generation (fused PROPAGATE + INTERACT + record, decoupled
//                    look-back compaction), the three-kernel path k_hit / k_scan / k_shade /
//                    k_advance, the per-object kernels k_intersect / k_normals / k_material_trace
//   prt_sources.hpp  k_source (ray sources on the device)
//   prt_render.hpp   k_render, k_render_hits, k_gooch*, k_camera, k_edge_* (renderers)
//   (this file)      the extern "C" entry points
//
// Data layout in HBM
//   ray set      (13, n) float64 row-major = PyRayT's RaySet verbatim (pyrayt/_pyrayt.py:13-44):
//                one contiguous row per field -> lane i of a wave reads element i of every
//                row: each of the 13 loads is a fully coalesced 512 B wave transaction.
//   record rows  (15, cap) float64 row-major: one contiguous row per DataFrame column
//                (_pyrayt.py:154-165); generation g appends its rows at column row_base(g).
//   scene        DevPrim[] + DevInstr[]: wave-uniform, fetched through the scalar cache.
//   per-lane CSG hit lists live in LDS (see prt_device.hpp), never in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/prt.h"
#include "prt_device.hpp"

#include "prt_scene.hpp"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ LaneLists lane_lists(int slots) { return LaneLists{slots, 0}; }

__device__ __forceinline__ Ray8 load_ray8(const double* __restrict__ rays, int64_t ld, int64_t i) {
  Ray8 r;
  r.ox = rays[0 * ld + i]; r.oy = rays[1 * ld + i]; r.oz = rays[2 * ld + i]; r.ow = rays[3 * ld + i];
  r.dx = rays[4 * ld + i]; r.dy = rays[5 * ld + i]; r.dz = rays[6 * ld + i]; r.dw = rays[7 * ld + i];
  r.gated = false;    // (nearest_hit() opens the gate; a ray that never meets it takes no shortcut)
  r.any_w = true;
  r.lex = false;
  r.paths = nullptr;
  return r;
}

// control block of a trace, in device memory
struct TraceCtrl {
  int64_t n_cur;      // rays alive at the entry of the current generation
  int64_t n_live;     // ... of which are recorded this generation
  int64_t n_carry;    // ... of which go on to the next generation
  int64_t row_base;   // first record column of the current generation
  int64_t rows_cap;
  int32_t error;      // PRT_ERR_* raised on the device
  int32_t pad;
  unsigned long long paths[4];  // PRT_TRACE_COUNT_PATHS: [1] rays not well formed, [2] implied-box nodes with survivors, [3] ... tested exactly
};

// dead-ray rule of _pyrayt.py:415-420: absorbed (|d| ~ 0 before the interaction) or no hit;
// the intensity threshold is a no-op upstream (Q2).
// |d| <= 1e-8 is decided on |d|^2: sqrt is monotonic and correctly rounded, and 0x1.cd2b297d889bdp-54
// is the largest double whose square root does not exceed the double 1e-8 (NaN compares false both ways)
__device__ __forceinline__ bool is_live(const Ray8& r, int prim) {
  const double len2 = ((r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz) + r.dw * r.dw;
  return !(len2 <= 0x1.cd2b297d889bdp-54 || prim < 0);
}

// prt_interact receives surface ids from the caller: map one back to its primitive (-1 = none)
__device__ __forceinline__ int prim_of_surface(const DevPrim* __restrict__ prims, int n_prims,
                                               int64_t sid) {
  int prim = -1;
  for (int p = 0; p < n_prims; ++p)
    if (sid >= 0 && (int64_t)prims[p].surface_id == sid) prim = p;
  return prim;
}

// [_st_propagate] nearest hit + per-workgroup counts of live / carried rays.
// prt_propagate uses the same kernel with counts == nullptr and surf_out != nullptr.
// STAGED (experiment, A/B partner of the scalar-load step fetch): the workgroup first copies the
// program into LDS and the interpreter reads its steps from there.
// COUNT: the PRT_TRACE_COUNT_PATHS instantiation (the counters cost the kernel 40 VGPRs and a wave of
// occupancy, so the kernel that serves prt_propagate and the ordinary three-kernel trace carries none)
template <bool STAGED, bool COUNT = false>
__global__ void __launch_bounds__(PRT_BLOCK)
k_hit(SceneDev scene, const double* __restrict__ rays, int64_t ld, const TraceCtrl* __restrict__ ctrl,
      int64_t n_fixed, double* __restrict__ hit_t, int32_t* __restrict__ hit_prim,
      int64_t* __restrict__ surf_out, int32_t* __restrict__ block_counts, int keep_absorbed,
      unsigned long long* __restrict__ paths) {
  const int64_t n = ctrl ? ctrl->n_cur : n_fixed;
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const LaneLists lists = lane_lists(scene.lds_slots);
  const DevInstr* code = scene.code;
  if (STAGED) {
    double* stage = lds_dyn + ((3 * scene.lds_slots + 1) / 2) * PRT_BLOCK;
    const double* src = reinterpret_cast<const double*>(scene.code);
    const int words = scene.n_instr * (int)(sizeof(DevInstr) / sizeof(double));
    for (int k = threadIdx.x; k < words; k += PRT_BLOCK) stage[k] = src[k];
    __syncthreads();
    code = reinterpret_cast<const DevInstr*>(stage);
  }
  bool live = false, carry = false;
  if (i < n) {
    Ray8 r = load_ray8(rays, ld, i);
    r.paths = COUNT ? paths : nullptr;
    double t;
    int prim;
    nearest_hit(scene.prims, code, scene.n_instr, r, lists, t, prim);
    hit_t[i] = t;
    if (hit_prim) hit_prim[i] = prim;
    if (surf_out) surf_out[i] = prim >= 0 ? (int64_t)scene.prims[prim].surface_id : -1;
    live = is_live(r, prim);
    carry = live && (keep_absorbed || scene.prims[prim].mat_kind != MAT_ABSORBER);
  }
  if (block_counts) {
    __shared__ int s_live, s_carry;
    if (threadIdx.x == 0) { s_live = 0; s_carry = 0; }
    __syncthreads();
    const int w_live = __popcll(__ballot(live));
    const int w_carry = __popcll(__ballot(carry));
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&s_live, w_live);
      atomicAdd(&s_carry, w_carry);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      block_counts[2 * blockIdx.x + 0] = s_live;
      block_counts[2 * blockIdx.x + 1] = s_carry;
    }
  }
}

// [_st_propagate, surface-parallel form] K lanes per ray.  Lane j of a ray's group evaluates components
// j, j + K, ... (each with its own program counter), then the group reduces its candidates with
// wavefront shuffles to the lexicographic minimum of (t, component order) -- exactly the running
// strict '<' minimum of _pyrayt.py:380-386: the earliest component among those with the smallest t.
// Lane 0 of the group owns the ray's outputs.  STAGED: the program is first copied to LDS by the
// workgroup and the per-lane step records are read from there (ds_read) instead of through vector
// loads -- the north-star sketch's "LDS-staged surface transform matrices".
template <int K, bool STAGED>
__global__ void __launch_bounds__(PRT_BLOCK)
k_hit_lanes(SceneDev scene, const int32_t* __restrict__ comp_first, int n_comp,
            const double* __restrict__ rays, int64_t ld, const TraceCtrl* __restrict__ ctrl, int64_t n_fixed,
            double* __restrict__ hit_t, int32_t* __restrict__ hit_prim, int64_t* __restrict__ surf_out,
            int32_t* __restrict__ tile_counts, int keep_absorbed) {
  constexpr int RAYS = PRT_BLOCK / K;
  const int64_t n = ctrl ? ctrl->n_cur : n_fixed;
  const int sub = threadIdx.x % K;
  const int64_t i = (int64_t)blockIdx.x * RAYS + threadIdx.x / K;
  const LaneLists lists = lane_lists(scene.lds_slots);
  const DevInstr* code = scene.code;
  if (STAGED) {
    // the program behind the hit lists, 16-byte aligned; every thread copies a strided share
    double* stage = lds_dyn + ((3 * scene.lds_slots + 1) / 2) * PRT_BLOCK;
    const double* src = reinterpret_cast<const double*>(scene.code);
    const int words = scene.n_instr * (int)(sizeof(DevInstr) / sizeof(double));
    for (int k = threadIdx.x; k < words; k += PRT_BLOCK) stage[k] = src[k];
    __syncthreads();
    code = reinterpret_cast<const DevInstr*>(stage);
  }
  bool live = false, carry = false;
  if (i < n) {
    Ray8 r = load_ray8(rays, ld, i);
    r.gated = true;  // well_formed() is the one gate of every shortcut (prt_device.hpp)
    // (each lane reduces ITS components with the strict '<' in ascending list index; the group reduce below
    // is lexicographic on (t, list index) whatever order the program stores the components in)
    double best_t = PRT_INF;
    int best_prim = -1, best_comp = 0x7fffffff;
    for (int c = sub; c < n_comp; c += K) {
      double t;
      int prim;
      component_candidate(code, comp_first[2 * c], comp_first[2 * c + 1], r, lists, t, prim);
      if (t < best_t) { best_t = t; best_prim = prim; best_comp = c; }
    }
#pragma unroll
    for (int off = K / 2; off > 0; off >>= 1) {  // group of K adjacent lanes, K a power of two <= 64
      const double t2 = __shfl_xor(best_t, off);
      const int p2 = __shfl_xor(best_prim, off), c2 = __shfl_xor(best_comp, off);
      const bool take = t2 < best_t || (t2 == best_t && c2 < best_comp);
      best_t = take ? t2 : best_t;
      best_prim = take ? p2 : best_prim;
      best_comp = take ? c2 : best_comp;
    }
    if (sub == 0) {
      hit_t[i] = best_t;
      if (hit_prim) hit_prim[i] = best_prim;
      if (surf_out) surf_out[i] = best_prim >= 0 ? (int64_t)scene.prims[best_prim].surface_id : -1;
      live = is_live(r, best_prim);
      carry = live && (keep_absorbed || scene.prims[best_prim].mat_kind != MAT_ABSORBER);
    }
  }
  if (tile_counts) {  // counts per PRT_BLOCK-ray tile, the unit k_scan / k_shade work in (zeroed by the caller)
    const int w_live = __popcll(__ballot(live)), w_carry = __popcll(__ballot(carry));
    if ((threadIdx.x & 63) == 0 && (w_live | w_carry)) {
      const int64_t first_ray = (int64_t)blockIdx.x * RAYS + (threadIdx.x / K);
      atomicAdd(&tile_counts[2 * (first_ray / PRT_BLOCK)], w_live);
      atomicAdd(&tile_counts[2 * (first_ray / PRT_BLOCK) + 1], w_carry);
    }
  }
}

// exclusive scan of the (live, carry) workgroup counts; one workgroup, grid-stride chunks
__global__ void __launch_bounds__(1024)
k_scan(const int32_t* __restrict__ block_counts, int64_t* __restrict__ block_offsets,
       TraceCtrl* __restrict__ ctrl) {
  __shared__ int64_t s_part[2][1024];
  __shared__ int64_t s_run[2];
  const int64_t n_blocks = (ctrl->n_cur + PRT_BLOCK - 1) / PRT_BLOCK;
  if (threadIdx.x == 0) { s_run[0] = 0; s_run[1] = 0; }
  __syncthreads();
  for (int64_t base = 0; base < n_blocks; base += 1024) {
    const int64_t b = base + threadIdx.x;
    int64_t v0 = 0, v1 = 0;
    if (b < n_blocks) { v0 = block_counts[2 * b]; v1 = block_counts[2 * b + 1]; }
    s_part[0][threadIdx.x] = v0;
    s_part[1][threadIdx.x] = v1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      int64_t a0 = 0, a1 = 0;
      if ((int)threadIdx.x >= off) {
        a0 = s_part[0][threadIdx.x - off];
        a1 = s_part[1][threadIdx.x - off];
      }
      __syncthreads();
      s_part[0][threadIdx.x] += a0;
      s_part[1][threadIdx.x] += a1;
      __syncthreads();
    }
    if (b < n_blocks) {
      block_offsets[2 * b] = s_run[0] + s_part[0][threadIdx.x] - v0;
      block_offsets[2 * b + 1] = s_run[1] + s_part[1][threadIdx.x] - v1;
    }
    __syncthreads();
    if (threadIdx.x == 1023) {
      s_run[0] += s_part[0][1023];
      s_run[1] += s_part[1][1023];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ctrl->n_live = s_run[0];
    ctrl->n_carry = s_run[1];
    if (ctrl->row_base + s_run[0] > ctrl->rows_cap) ctrl->error = PRT_ERR_ROWS_CAP;
  }
}

// exclusive rank of this lane among the flagged lanes of its workgroup
__device__ __forceinline__ int block_rank(bool flag, int* s_wave /*[4]*/) {
  const unsigned long long mask = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int before = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[wave] = __popcll(mask);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  __syncthreads();
  return base + before;
}

// One live ray through INTERACT (_pyrayt.py:394-452) and the record writer (:168-186), in two
// halves so that a kernel can do the arithmetic before it knows where the results go:
//   interact_compute : load the 5 metadata rows, advance to the hit point, shade by the material
//                      of the primitive hit (wave "waterfall" over the distinct primitives)
//   interact_store   : record row at column `row`, next state at column `j` if the ray goes on
struct Shaded {
  double generation, intensity, wavelength, index_in, id;  // pre-hit metadata
  double px, py, pz, pw;                                    // hit point
  double dx, dy, dz, dw, index;                             // post-interaction direction / index
  double tx, ty, tz;                                        // unit tilt of the incoming ray
  double surface_id;
  int err;                                                  // 0, or the PRT_ERR_* this ray raises (see shade())
};

struct Meta5 {
  double generation, intensity, wavelength, index, id;
};
__device__ __forceinline__ Meta5 load_meta(const double* __restrict__ rays, int64_t ld, int64_t i) {
  Meta5 m;
  m.generation = rays[8 * ld + i];
  m.intensity = rays[9 * ld + i];
  m.wavelength = rays[10 * ld + i];
  m.index = rays[11 * ld + i];
  m.id = rays[12 * ld + i];
  return m;
}

__device__ __forceinline__ Shaded interact_compute(const SceneDev& scene, const Meta5& m, const Ray8& r,
                                                   double t, int prim) {
  Shaded s;
  s.generation = m.generation;
  s.intensity = m.intensity;
  s.wavelength = m.wavelength;
  s.index_in = m.index;
  s.id = m.id;
  // advance to the hit point: o += d * t, all four homogeneous components (_pyrayt.py:404-407)
  s.px = r.ox + r.dx * t; s.py = r.oy + r.dy * t; s.pz = r.oz + r.dz * t; s.pw = r.ow + r.dw * t;
  s.dx = r.dx; s.dy = r.dy; s.dz = r.dz; s.dw = r.dw;
  s.index = s.index_in;
  s.surface_id = -1.0;
  s.err = 0;
  // tilt columns: pre-hit direction over its 3-norm (_pyrayt.py:176-177)
  const double tilt = norm3(r.dx, r.dy, r.dz);
  div3(r.dx, r.dy, r.dz, tilt, s.tx, s.ty, s.tz);
  // material dispatch: the primitive table is wave-uniform data, so lanes that hit the same
  // surface shade together and the loop runs once per distinct surface hit in the wave
  // (the table reads below are per-lane vector loads of one address -- a single L1 line per field
  // group, fetched in one batch.  The scalar-load form of this loop -- v_readfirstlane of the lanes still
  // pending, table entry through the constant address space -- measured 7 % slower for the whole kernel:
  // its reads are dependent round trips to the scalar cache.)
  unsigned long long todo = __ballot(true);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int cur = __shfl(prim, leader);
    if (prim == cur) {
      const DevPrim* __restrict__ p = scene.prims + cur;
      s.err = shade(p, s.px, s.py, s.pz, s.pw, s.dx, s.dy, s.dz, s.dw, s.wavelength, s.index,
                   s.tx, s.ty, s.tz);
      s.surface_id = p->surface_id;
    }
    todo &= ~__ballot(prim == cur);
  }
  return s;
}

__device__ __forceinline__ void interact_store(const Shaded& s, const Ray8& r, bool carry, int64_t row,
                                               int64_t j, double* __restrict__ next, int64_t ld_next,
                                               double* __restrict__ rows, int64_t ld_rows,
                                               double next_generation, int relaunch, double ray_offset) {
  // record row (_pyrayt.py:168-186): pre-hit metadata, surface, start, end, unit tilt
  rows[PRT_COL_GENERATION * ld_rows + row] = s.generation;
  rows[PRT_COL_INTENSITY * ld_rows + row] = s.intensity;
  rows[PRT_COL_WAVELENGTH * ld_rows + row] = s.wavelength;
  rows[PRT_COL_INDEX * ld_rows + row] = s.index_in;
  rows[PRT_COL_ID * ld_rows + row] = s.id;
  rows[PRT_COL_SURFACE * ld_rows + row] = s.surface_id;
  rows[PRT_COL_X0 * ld_rows + row] = r.ox;
  rows[PRT_COL_Y0 * ld_rows + row] = r.oy;
  rows[PRT_COL_Z0 * ld_rows + row] = r.oz;
  rows[PRT_COL_X1 * ld_rows + row] = s.px;
  rows[PRT_COL_Y1 * ld_rows + row] = s.py;
  rows[PRT_COL_Z1 * ld_rows + row] = s.pz;
  rows[PRT_COL_XTILT * ld_rows + row] = s.tx;
  rows[PRT_COL_YTILT * ld_rows + row] = s.ty;
  rows[PRT_COL_ZTILT * ld_rows + row] = s.tz;
  if (carry) {
    // next state (_pyrayt.py:437-449): generation + 1, re-launch 1e-6 along the new direction
    double qx = s.px, qy = s.py, qz = s.pz, qw = s.pw;
    if (relaunch) {
      qx = s.px + ray_offset * s.dx; qy = s.py + ray_offset * s.dy; qz = s.pz + ray_offset * s.dz;
      qw = s.pw + ray_offset * s.dw;
    }
    next[0 * ld_next + j] = qx;
    next[1 * ld_next + j] = qy;
    next[2 * ld_next + j] = qz;
    next[3 * ld_next + j] = qw;
    next[4 * ld_next + j] = s.dx;
    next[5 * ld_next + j] = s.dy;
    next[6 * ld_next + j] = s.dz;
    next[7 * ld_next + j] = s.dw;
    next[8 * ld_next + j] = next_generation;
    next[9 * ld_next + j] = s.intensity;
    next[10 * ld_next + j] = s.wavelength;
    next[11 * ld_next + j] = s.index;
    next[12 * ld_next + j] = s.id;
  }
}

__device__ __forceinline__ int interact_lane(const SceneDev& scene, const double* __restrict__ rays,
                                              int64_t ld, int64_t i, const Ray8& r, double t, int prim,
                                              bool carry, int64_t row, int64_t j,
                                              double* __restrict__ next, int64_t ld_next,
                                              double* __restrict__ rows, int64_t ld_rows,
                                              double next_generation, int relaunch, double ray_offset) {
  const Shaded s = interact_compute(scene, load_meta(rays, ld, i), r, t, prim);
  interact_store(s, r, carry, row, j, next, ld_next, rows, ld_rows, next_generation, relaunch, ray_offset);
  return s.err;
}

// [_st_interact + _RayTraceDataframe.insert]
__global__ void __launch_bounds__(PRT_BLOCK)
k_shade(SceneDev scene, const double* __restrict__ rays, int64_t ld, const TraceCtrl* __restrict__ ctrl_in,
        int64_t n_fixed, const double* __restrict__ hit_t, const int32_t* __restrict__ hit_prim,
        const int64_t* __restrict__ hit_surf, const int64_t* __restrict__ block_offsets,
        double* __restrict__ next, int64_t ld_next, double* __restrict__ rows, int64_t ld_rows,
        int64_t row_base_fixed, double next_generation, int relaunch, double ray_offset,
        int keep_absorbed, TraceCtrl* __restrict__ ctrl, const double* __restrict__ shaded, int64_t ld_shaded) {
  __shared__ int s_wave[4];
  const int64_t n = ctrl_in ? ctrl_in->n_cur : n_fixed;
  if (ctrl_in && (ctrl_in->error != 0 || ctrl_in->n_live == 0)) return;  // uniform
  const int64_t row_base = ctrl_in ? ctrl_in->row_base : row_base_fixed;
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const bool in_range = i < n;
  Ray8 r = {0, 0, 0, 1, 0, 0, 0, 0};
  double t = PRT_INF;
  int prim = -1;
  if (in_range) {
    r = load_ray8(rays, ld, i);
    t = hit_t[i];
    if (hit_prim) {
      prim = hit_prim[i];
    } else {  // prt_interact (n_instr carries the primitive count on this path)
      prim = prim_of_surface(scene.prims, scene.n_instr, hit_surf[i]);
    }
  }
  const bool live = in_range && is_live(r, prim);
  const bool carry = live && (keep_absorbed || scene.prims[prim].mat_kind != MAT_ABSORBER);
  const int live_rank = block_rank(live, s_wave);
  const int carry_rank = block_rank(carry, s_wave);
  if (!live) return;

  const int64_t row = row_base + block_offsets[2 * blockIdx.x] + live_rank;
  const int64_t j = block_offsets[2 * blockIdx.x + 1] + carry_rank;
  if (shaded != nullptr && scene.prims[prim].mat_kind == MAT_HOST) {
    // A surface whose material.trace() is the caller's own code (_pyrayt.py:408-410): column i of `shaded` is
    // what it returned for this ray -- all 13 rows are taken over (:408 assigns the whole column), the record
    // row keeps the pre-hit metadata and ends at the origin trace() left (:172, :181), the generation is set
    // as for every ray (:437) and the re-launch runs along the direction trace() left (:449).
    const Meta5 m = load_meta(rays, ld, i);
    Shaded s;
    s.generation = m.generation; s.intensity = m.intensity; s.wavelength = m.wavelength;
    s.index_in = m.index; s.id = m.id;
    s.px = shaded[0 * ld_shaded + i]; s.py = shaded[1 * ld_shaded + i]; s.pz = shaded[2 * ld_shaded + i];
    s.pw = shaded[3 * ld_shaded + i];
    s.dx = shaded[4 * ld_shaded + i]; s.dy = shaded[5 * ld_shaded + i]; s.dz = shaded[6 * ld_shaded + i];
    s.dw = shaded[7 * ld_shaded + i];
    s.index = shaded[11 * ld_shaded + i];
    s.surface_id = scene.prims[prim].surface_id;
    s.err = 0;
    const double tilt = norm3(r.dx, r.dy, r.dz);
    div3(r.dx, r.dy, r.dz, tilt, s.tx, s.ty, s.tz);
    interact_store(s, r, carry, row, j, next, ld_next, rows, ld_rows, next_generation, relaunch, ray_offset);
    if (carry) {  // the metadata trace() may have changed as well
      next[9 * ld_next + j] = shaded[9 * ld_shaded + i];
      next[10 * ld_next + j] = shaded[10 * ld_shaded + i];
      next[12 * ld_next + j] = shaded[12 * ld_shaded + i];
    }
    return;
  }
  const int err = interact_lane(scene, rays, ld, i, r, t, prim, carry, row, j, next, ld_next, rows, ld_rows,
                                next_generation, relaunch, ray_offset);
  if (err) atomicExch(&ctrl->error, err);
}

// end of a generation: roll the control block forward
__global__ void k_advance(TraceCtrl* ctrl, int64_t* __restrict__ rows_per_generation, int generation) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const bool bad = ctrl->error != 0;
    const int64_t live = bad ? 0 : ctrl->n_live;
    rows_per_generation[generation] = live;
    ctrl->row_base += live;
    ctrl->n_cur = (live == 0) ? 0 : ctrl->n_carry;
    ctrl->n_live = 0;
    ctrl->n_carry = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// fused generation: PROPAGATE + INTERACT + record in ONE launch, one pass over HBM
// ------------------------------------------------------------------------------------------------
// per-generation control slot (device memory).  Generation g reads slot g and its last tile
// fills slot g+1, so consecutive generations chain on the stream with no host round trip.
struct GenCtrl {
  int64_t n_in;      // rays alive at entry
  int64_t row_base;  // first record column of this generation
  int64_t n_live;    // rows recorded by this generation
  int64_t n_carry;   // rays handed to the next generation
  uint32_t pad[8];
};
static_assert(sizeof(GenCtrl) == 64, "host_gen sizing");
// Per-tile record of a generation that compacted (look-back): where the tile's rows and carried rays went and how
// many it had.  The next trace of this ticket with this workspace may run that generation on the record instead
// of a look-back (assume == 3): every tile checks its own counts against it, exactly as dense mode checks "all".
// The records of the first kTileHintGenerations generations sit at a fixed distance behind the generation slots
// (no kernel argument of their own: the generation kernel has no register to spare for one).
struct TileHint { unsigned excl_live, excl_carry, live, carry; };
static const int kTileHintGenerations = 16;
// Dead lists: a generation launched dense with its absorbed rays kept (hint mode 4) notes, per tile that kept any,
// (tile << 9 | how many); the generation behind it, launched on that list (assume 5 / 6), takes its tiles' offsets
// from "tile index x tile size minus the dead rays in front" -- no look-back for a handful of dead rays.  Three
// lists in rotation: generation g writes list g % 3, g + 1 reads it, and every generation empties list (g + 1) % 3.
static const int kDeadListCap = 1020;
struct DeadList { unsigned count, pad[3], entry[kDeadListCap]; };
static_assert(sizeof(DeadList) == 4096, "dead list sizing");
static const size_t kDeadListOffset =
    ((size_t)(kMaxGenerationSlots + 2) * sizeof(GenCtrl) + 255) / 256 * 256 - sizeof(GenCtrl);  // from gen[0], see trace_layout
static const size_t kTileHintOffset = kDeadListOffset + 3 * sizeof(DeadList);
__device__ __forceinline__ DeadList* dead_list(GenCtrl* gen, int g) {
  return reinterpret_cast<DeadList*>(reinterpret_cast<char*>(gen) + kDeadListOffset) + (g % 3);
}
__device__ __forceinline__ TileHint* tile_hints(GenCtrl* gen, int g) {
  return reinterpret_cast<TileHint*>(reinterpret_cast<char*>(gen) + kTileHintOffset) + (size_t)g * gridDim.x;
}
struct FusedCtrl {
  int32_t error;
  int32_t pad;
  int64_t rows_cap;
};

// What the host needs from a batch of generations, in host-mapped (fine-grained) memory: the device
// writes it at the end of the batch and the host spins on `epoch` -- no copy engine, no interrupt.
struct HostMirror {
  unsigned long long epoch;  // written last, system scope
  int32_t error;
  int32_t pad[13];
  GenCtrl gen[kMaxBatch + 4];
};
static_assert(offsetof(HostMirror, gen) == 64, "mirror header");

#define PRT_ERR_SPECULATION (-101) /* internal: a generation launched in dense mode was not dense -> host re-runs without hints */
#define PRT_ERR_STALL (-100) /* internal: look-back gave up -> host falls back to the unfused path */
#define PRT_ERR_TILE_HINT (-103) /* internal: a generation launched on the per-tile record of its last run found other counts -> host re-runs without the records */
#define PRT_ERR_FULL_ROWS (-102) /* internal: a ray set needs the rows the compact form leaves out -> host re-runs with all 13 */

// Raising an error on the device.  The verdicts that make the host repeat the trace (SPECULATION, STALL,
// FULL_ROWS) replace whatever is there and are WAITED FOR -- the returned value is consumed, so the
// atomic has been performed at the memory side before anything this lane does next (publishing a tile
// word, checking in, telling the host): a tile that consumes this tile's word afterwards and then reads
// the error word sees the verdict.  ROWS_CAP and UNTRACABLE never replace another error: a dense-mode
// generation measures the record block against its ASSUMED offsets, so after a missed hint a tile may
// find the block too small although the real rows fit -- the miss is what the host has to hear about.
__device__ __forceinline__ void raise_verdict(int32_t* error, int code) {
  const int before = atomicExch(error, code);
  asm volatile("" ::"v"(before) : "memory");
}
__device__ __forceinline__ void raise_error(int32_t* error, int code) { atomicCAS(error, 0, code); }

// tile status word for the decoupled look-back: [63:62] status, [61:31] live, [30:0] carried.
// One naturally aligned 8-byte word written by one agent-scope store: payload and flag cannot
// be observed torn, so no fence is needed around it.
#define TILE_INVALID 0ull
#define TILE_AGGREGATE 1ull
#define TILE_PREFIX 2ull
__device__ __forceinline__ unsigned long long tile_pack(unsigned long long status, unsigned live,
                                                        unsigned carry) {
  return (status << 62) | ((unsigned long long)live << 31) | (unsigned long long)carry;
}
__device__ __forceinline__ void tile_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long tile_load(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of (live, carry) over all tiles before `tile`; called by one full wave.
// Lane l inspects tile (base - l).  The wave consumes the contiguous run of published words
// nearest to it -- up to and including the first inclusive prefix -- and moves on; if the
// nearest predecessor has not published yet it backs off with s_sleep and polls again.
// Tiles are numbered by blockIdx.x: the dispatcher starts workgroups in index order, so
// every predecessor is resident (or finished) before its successors and publishes without
// waiting on anything behind it.  That ordering is not an architectural guarantee, so the spin
// is bounded: on expiry the wave reports failure and the host re-runs the generation loop on
// the three-kernel path, which has no inter-workgroup dependency.
#ifndef PRT_LOOKBACK_WINDOWS
#define PRT_LOOKBACK_WINDOWS 1
#endif
#ifndef PRT_LOOKBACK_LANES
#define PRT_LOOKBACK_LANES 32  // words inspected per poll: polls are fabric traffic; 16-32 measured best (64: +1-2 %, 8: +7 %)
#endif
__device__ __forceinline__ bool lookback(unsigned long long* state, int tile, unsigned agg_live,
                                         unsigned agg_carry, unsigned& excl_live,
                                         unsigned& excl_carry, int32_t* error) {
  // One poll fetches PRT_LOOKBACK_WINDOWS x 64 predecessor words (lane l: tiles base - l, base - 64 - l,
  // ...), all loads in flight together: a poll is a round trip to the fabric (~1 us under load, the
  // words are device-scope), and the nearest inclusive prefix is typically 40-130 tiles back
  // (tools/lookback_analysis.py), so one window per poll meant two or three dependent round trips.
  const int lane = threadIdx.x & 63;
  unsigned sum_live = 0, sum_carry = 0;
  int base = tile - 1;
  int idle = 0;
  bool ok = true;
  while (base >= 0) {
    unsigned long long w[PRT_LOOKBACK_WINDOWS];
#pragma unroll
    for (int k = 0; k < PRT_LOOKBACK_WINDOWS; ++k) {
      const int idx = base - 64 * k - lane;
      w[k] = tile_pack(TILE_PREFIX, 0, 0);  // before tile 0: empty prefix
      if (idx >= 0) w[k] = lane < PRT_LOOKBACK_LANES ? tile_load(state + idx) : TILE_INVALID;
    }
    bool closed = false, stalled = false;
    int consumed = 0;
#pragma unroll
    for (int k = 0; k < PRT_LOOKBACK_WINDOWS; ++k) {
      if (closed || stalled) break;
      const unsigned long long status = w[k] >> 62;
      const unsigned long long pending = __ballot(status == TILE_INVALID);
      const unsigned long long prefix = __ballot(status == TILE_PREFIX);
      // lanes [0, run) have published; stop after the first inclusive prefix among them
      int run = pending ? (__ffsll((long long)pending) - 1) : 64;
      if (prefix) {
        const int first = __ffsll((long long)prefix) - 1;
        if (first < run) { run = first + 1; closed = true; }
      }
      unsigned l = (lane < run) ? (unsigned)((w[k] >> 31) & 0x7fffffffull) : 0u;
      unsigned c = (lane < run) ? (unsigned)(w[k] & 0x7fffffffull) : 0u;
      for (int off = 32; off > 0; off >>= 1) {
        l += __shfl_xor(l, off);
        c += __shfl_xor(c, off);
      }
      sum_live += l;
      sum_carry += c;
      consumed += run;
      stalled = run < 64 && !closed;
    }
    if (closed) break;
    if (consumed == 0) {
      // (an idle poll is a fabric round trip plus s_sleep 8, ~1 us: 2^17 of them are ~0.15 s -- three orders
      // of magnitude beyond the longest a live predecessor has been seen to take, and short enough that a
      // box whose dispatcher does not start workgroups in index order falls back without a visible stall)
      if (++idle > (1 << 17)) { ok = false; break; }
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    base -= consumed;
  }
  excl_live = sum_live;
  excl_carry = sum_carry;
  if (lane == 0) {
    // (an expired spin leaves a partial prefix: the verdict goes out first, so that whoever consumes
    // this word -- the last tile, which tells the host, included -- finds the error word set)
    if (!ok) raise_verdict(error, PRT_ERR_STALL);
    tile_store(state + tile, tile_pack(TILE_PREFIX, sum_live + agg_live, sum_carry + agg_carry));
  }
  return ok;
}

#ifdef PRT_TIMING
// experiment build: s_memtime stamps of every wave of one generation (PRT_TIMING_GEN, default 0) at 8 points of k_generation
__device__ long long g_stamps[16384 * 4 * 8];
#ifndef PRT_TIMING_GEN
#define PRT_TIMING_GEN 0
#endif
#define STAMP(k) do { if (g == PRT_TIMING_GEN && (threadIdx.x & 63) == 0 && blockIdx.x < 16384) g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// parking rows sit behind the hit lists (f64 rows, then int32 rows = half an f64 row each)
#define PARK(k) lds_dyn[(park_base + (k)) * PRT_BLOCK + threadIdx.x]
#define PRT_PARK_ROWS 8

// ---- row-major HBM access through buffer descriptors ----------------------------------------------
// Every access of the generation kernel is "element (uniform column + lane offset) of row k".  A
// flat global_load/store needs the 64-bit address per lane (one VALU add per row plus the scalar
// 64-bit row multiply); a buffer access takes the row's base from four SGPRs and one 32-bit lane
// offset shared by all rows, so a row costs two scalar adds and the memory instruction itself.
typedef unsigned int prt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff8, 0x00020000);
}
#ifndef PRT_LOAD_AUX
#define PRT_LOAD_AUX 2
#endif
__device__ __forceinline__ double row_load(const double* base, unsigned lane_bytes) {
  const prt_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(row_rsrc(base), lane_bytes, 0, PRT_LOAD_AUX);
  return __hiloint2double((int)v.y, (int)v.x);
}
// Cache policy of the three streams (aux bit 1 = nt, "non-temporal").  The record rows are written
// once and never read by the GPU again; the ray state a generation reads is dead once read; the next
// state, on the other hand, is what the following launch reads -- 104 MB per 1M rays, which the
// 256 MB Infinity Cache can hold if the other two streams do not sweep it out.  Measured (config 2,
// interleaved A/B): nt on the record stores -6...9 %, plus nt on the state loads another -3 %;
// nt on the next-state stores as well gives most of it back (+6 %).
#ifndef PRT_STORE_AUX_REC
#define PRT_STORE_AUX_REC 2
#endif
#ifndef PRT_STORE_AUX_NEXT
#define PRT_STORE_AUX_NEXT 0
#endif
template <int AUX = 0>
__device__ __forceinline__ void row_store(double* base, unsigned lane_bytes, double value) {
  prt_u32x2 v;
  v.x = (unsigned)__double2loint(value);
  v.y = (unsigned)__double2hiint(value);
  __builtin_amdgcn_raw_buffer_store_b64(v, row_rsrc(base), lane_bytes, 0, AUX);
}
// a value every lane holds identically, moved to SGPRs so that addresses built on it are scalar
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffll));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// interact_store through buffer descriptors: `rec` / `nxt` already point at this workgroup's first
// column of row 0, lane offsets are rank * 8 bytes
// homogeneous coordinates of a well-formed ray: origin w exactly 1, direction w exactly +0 (bit for bit:
// a -0 is kept apart because it can decide the sign of a zero sum in the object-space transform)
__device__ __forceinline__ bool w_is_trivial(double ow, double dw) {
  return ow == 1.0 && __double_as_longlong(dw) == 0ll;
}

// compact: the next state goes without its rows 3, 7 and 8 (see k_generation); returns false if a ray
// that goes on does not have the values the reader will assume for them
template <bool COMPACT>
__device__ __forceinline__ bool interact_store_rows(const Shaded& s, const Ray8& r, bool carry, unsigned row_bytes,
                                                    unsigned next_bytes, double* __restrict__ nxt, int64_t ld_next,
                                                    double* __restrict__ rec, int64_t ld_rows,
                                                    double next_generation, int relaunch, double ray_offset) {
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_GENERATION * ld_rows, row_bytes, s.generation);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_INTENSITY * ld_rows, row_bytes, s.intensity);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_WAVELENGTH * ld_rows, row_bytes, s.wavelength);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_INDEX * ld_rows, row_bytes, s.index_in);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_ID * ld_rows, row_bytes, s.id);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_SURFACE * ld_rows, row_bytes, s.surface_id);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_X0 * ld_rows, row_bytes, r.ox);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_Y0 * ld_rows, row_bytes, r.oy);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_Z0 * ld_rows, row_bytes, r.oz);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_X1 * ld_rows, row_bytes, s.px);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_Y1 * ld_rows, row_bytes, s.py);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_Z1 * ld_rows, row_bytes, s.pz);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_XTILT * ld_rows, row_bytes, s.tx);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_YTILT * ld_rows, row_bytes, s.ty);
  row_store<PRT_STORE_AUX_REC>(rec + PRT_COL_ZTILT * ld_rows, row_bytes, s.tz);
  if (carry) {
    double qx = s.px, qy = s.py, qz = s.pz, qw = s.pw;
    if (relaunch) {
      qx = s.px + ray_offset * s.dx; qy = s.py + ray_offset * s.dy; qz = s.pz + ray_offset * s.dz;
      qw = s.pw + ray_offset * s.dw;
    }
    row_store<PRT_STORE_AUX_NEXT>(nxt + 0 * ld_next, next_bytes, qx);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 1 * ld_next, next_bytes, qy);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 2 * ld_next, next_bytes, qz);
    if (!COMPACT) row_store<PRT_STORE_AUX_NEXT>(nxt + 3 * ld_next, next_bytes, qw);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 4 * ld_next, next_bytes, s.dx);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 5 * ld_next, next_bytes, s.dy);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 6 * ld_next, next_bytes, s.dz);
    if (!COMPACT) {
      row_store<PRT_STORE_AUX_NEXT>(nxt + 7 * ld_next, next_bytes, s.dw);
      row_store<PRT_STORE_AUX_NEXT>(nxt + 8 * ld_next, next_bytes, next_generation);
    }
    row_store<PRT_STORE_AUX_NEXT>(nxt + 9 * ld_next, next_bytes, s.intensity);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 10 * ld_next, next_bytes, s.wavelength);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 11 * ld_next, next_bytes, s.index);
    row_store<PRT_STORE_AUX_NEXT>(nxt + 12 * ld_next, next_bytes, s.id);
    if (COMPACT && relaunch) return w_is_trivial(qw, s.dw);  // (!relaunch: nobody reads this state)
  }
  return true;
}

// Register-allocated for 5 waves per SIMD (96 VGPRs, no spills -- possible because the record
// columns known before the shading wait in LDS, see PARK).  Measured on MI355X, same box,
// interleaved: occupancy matters (identical code held to 3 workgroups/CU by LDS padding is 30 %
// slower than at 4), 5 waves beat 4 by 7 %, and 6 (80 VGPRs, 56 B/lane of scratch) lose 16 %:
// spills in the fp64 hot path cost more than the extra wave buys.
#ifndef PRT_GEN_WAVES
#define PRT_GEN_WAVES 5
#endif
// CULL = the trace program carries component cull steps (scenes of three or more components);
// the instantiation without them is the one the register budget above was tuned for.
template <bool CULL, bool COMPACT>
__global__ void __launch_bounds__(PRT_BLOCK, PRT_GEN_WAVES)
k_generation(SceneDev scene, const double* __restrict__ rays, int64_t ld, double* __restrict__ next,
             int64_t ld_next, double* __restrict__ rows, int64_t ld_rows, FusedCtrl* __restrict__ ctrl,
             GenCtrl* __restrict__ gen, int g, unsigned long long* __restrict__ tiles_cur,
             unsigned long long* __restrict__ tiles_next, double next_generation, int generation_limit,
             double ray_offset, int keep_absorbed, HostMirror* mirror, unsigned long long epoch,
             int mirror_slot, int batch_last, int assume) {
  const int relaunch = (g + 1 != generation_limit) ? 1 : 0;  // the state written here is traced further
  __shared__ int s_wave_live[4], s_wave_carry[4];
  __shared__ unsigned s_excl[3];
  // A ticket from one atomic word would also give start-ordered tile numbers, but a single
  // word hands out only ~80 tickets/us chip-wide: 4k tiles would cost ~50 us per generation.
  const int tile = blockIdx.x;
  const int64_t n = gen[g].n_in;
  if ((int64_t)tile * PRT_BLOCK >= n) {  // uniform per workgroup; never a predecessor
    // The grid always covers the ray count the trace started with, so that every launch recycles the
    // whole of the other status buffer (below) whatever is left of the rays: the launch behind this one
    // -- of this trace or of the next -- then finds its buffer clean without a kernel in between.
    if (threadIdx.x == 0) tiles_next[tile] = TILE_INVALID;
    return;
  }
  // A tile that sees an error raised earlier does no work but MUST still publish its (empty)
  // aggregate: tiles behind it may already be waiting on it.
  const bool failed = ctrl->error != 0;
  const int64_t row_base = gen[g].row_base;
  if (threadIdx.x == 0) {
    tiles_next[tile] = TILE_INVALID;  // recycle the other buffer's word
    if (tile == 0) dead_list(gen, g + 1)->count = 0;  // ... and the dead list the next generation may write
  }

  const int64_t i = (int64_t)tile * PRT_BLOCK + threadIdx.x;
  const LaneLists lists = lane_lists(scene.lds_slots);
  const int park_base = (3 * scene.lds_slots + 1) / 2;
  Ray8 r = {0, 0, 0, 1, 0, 0, 0, 0};
  double t = PRT_INF;
  int prim = -1;
  bool live = false, carry = false, absorbs = false;
  STAMP(0);
  if (i < n && !failed) {
    // all 13 rows in one burst; the five metadata rows go straight to the lane's parking slots in
    // LDS (nothing needs them before the shading), so no second trip to HBM after the hit phase
    const double* tile_rays = rays + (int64_t)tile * PRT_BLOCK;
    const unsigned lane_bytes = threadIdx.x * 8u;
    // Compact state (the default; a scene falls back to all 13 rows the first time a ray set needs
    // them): between the generations of a trace the homogeneous w rows (3, 7) and the generation row
    // (8) hold what every well-formed ray set holds there -- 1, +0 and the generation's number -- so
    // the generations neither write nor read them: 24 B of the 104 B state, each way.  Generation 0
    // reads the caller's 13 rows and checks; every generation checks the rays it hands on; a ray
    // that differs raises PRT_ERR_FULL_ROWS and the host repeats the trace with all rows in use.
    // (COMPACT is a template parameter: the form with all 13 rows is the kernel it was before.)
    r.ox = row_load(tile_rays + 0 * ld, lane_bytes); r.oy = row_load(tile_rays + 1 * ld, lane_bytes);
    r.oz = row_load(tile_rays + 2 * ld, lane_bytes);
    if (!COMPACT) r.ow = row_load(tile_rays + 3 * ld, lane_bytes);
    r.dx = row_load(tile_rays + 4 * ld, lane_bytes); r.dy = row_load(tile_rays + 5 * ld, lane_bytes);
    r.dz = row_load(tile_rays + 6 * ld, lane_bytes);
    if (!COMPACT) {
      r.dw = row_load(tile_rays + 7 * ld, lane_bytes);
      PARK(0) = row_load(tile_rays + 8 * ld, lane_bytes);
    }
    PARK(1) = row_load(tile_rays + 9 * ld, lane_bytes);
    PARK(2) = row_load(tile_rays + 10 * ld, lane_bytes); PARK(3) = row_load(tile_rays + 11 * ld, lane_bytes);
    PARK(4) = row_load(tile_rays + 12 * ld, lane_bytes);
    if (COMPACT) {
      if (g > 0) {  // uniform
        r.ow = 1.0;
        r.dw = 0.0;
        PARK(0) = next_generation - 1.0;  // (small integers: exact)
      } else {
        r.ow = row_load(tile_rays + 3 * ld, lane_bytes);
        r.dw = row_load(tile_rays + 7 * ld, lane_bytes);
        const double generation = row_load(tile_rays + 8 * ld, lane_bytes);
        PARK(0) = generation;
        // (raised with a plain exchange: waiting for it here costs the kernel its register allocation -- 100 B
        // of scratch per lane; the tile publishes long after this point, behind the whole hit phase)
        if (!(w_is_trivial(r.ow, r.dw) && generation == 0.0)) atomicExch(&ctrl->error, PRT_ERR_FULL_ROWS);
      }
    }
#ifdef PRT_TIMING
    if (r.ox + r.oy + r.oz + r.ow + r.dx + r.dy + r.dz + r.dw == 1.2345e300) t = 0;  // force the wait here
    STAMP(1);
#endif
    // (COMPACT: every origin w is 1 -- assumed from generation 1 on, checked above in generation 0)
    nearest_hit<false, CULL, !COMPACT>(scene.prims, scene.code, scene.n_instr, r, lists, t, prim);
    live = is_live(r, prim);
    // absorbed rays are recorded and dropped; which primitives absorb is a bit mask for the first 64
    // (one scalar test instead of a per-lane table lookup in HBM)
    if (live) absorbs = prim < 64 ? ((scene.absorber_mask >> prim) & 1ull) != 0 : scene.prims[prim].mat_kind == MAT_ABSORBER;
    carry = live && (keep_absorbed || !absorbs);
  }
  // workgroup aggregate and ranks: wave ballots + popcounts, four waves combined through LDS
  STAMP(2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m_live = __ballot(live), m_carry = __ballot(carry);
  // (absorbed rays that go on, dead, because this launch keeps them: counted per tile for the dead list)
  const unsigned long long m_kept = keep_absorbed ? __ballot(absorbs) : 0ull;
  const unsigned long long below = (1ull << lane) - 1ull;
  int live_rank = __popcll(m_live & below), carry_rank = __popcll(m_carry & below);
  if (lane == 0) {
    s_wave_live[wave] = __popcll(m_live);
    s_wave_carry[wave] = __popcll(m_carry) | (__popcll(m_kept) << 16);  // (two counts of at most 64 in one word)
  }
  __syncthreads();
  unsigned agg_live = 0, agg_carry = 0;
  for (int w = 0; w < PRT_BLOCK / 64; ++w) {
    if (w < wave) { live_rank += s_wave_live[w]; carry_rank += s_wave_carry[w] & 0xffff; }
    agg_live += s_wave_live[w];
    agg_carry += s_wave_carry[w];
  }
  const unsigned agg_kept = agg_carry >> 16;
  agg_carry &= 0xffffu;
  // Launched on the dead list of the generation before (assume 5 / 6): the dead rays in front of this tile, in it,
  // and in all; one wave reads the list (a handful of entries as a rule: that is when the host offers this mode)
  if (assume >= 5 && wave == 0) {
    const DeadList* dead = dead_list(gen, g - 1);
    const unsigned entries = dead->count;
    unsigned before = 0, here = 0, all = 0;
    if (entries > kDeadListCap) here = 0xffffu;  // (more tiles kept rays than the list holds: no tile's counts agree)
    else if (entries) {
      for (unsigned k = lane; k < entries; k += 64) {
        const unsigned e = dead->entry[k], at = e >> 9, count = e & 511u;
        all += count;
        before += at < (unsigned)tile ? count : 0u;
        here += at == (unsigned)tile ? count : 0u;
      }
      for (int off = 32; off > 0; off >>= 1) {
        all += __shfl_xor(all, off);
        before += __shfl_xor(before, off);
        here += __shfl_xor(here, off);
      }
    }
    if (lane == 0) { s_excl[0] = before; s_excl[1] = here; s_excl[2] = all; }
  }
  STAMP(3);
  // Dense mode (assume != 0): the previous trace of this scene and ray count recorded every ray of this
  // generation and carried all (1) or none (2) of them on, so the host launched it on the assumption
  // that it will again: every tile's prefix is then its index times the tile size -- no status words,
  // no look-back, no second barrier.  Each tile checks the assumption on its own counts; a tile that
  // finds it wrong raises PRT_ERR_SPECULATION and the host repeats the trace without assumptions.
  // The totals of a dense generation are the assumption itself, so tile 0 hands over to g + 1.
  bool finisher = false;
  // the generation that ends the batch (known at launch in this mode) also tells the host: its tiles
  // check in on counters kept in the otherwise unused status buffer (cleared by the next generation like
  // any status word), and the last one to arrive publishes, with every tile's verdict visible to it
  // (assume == 3: whether the generation ends the trace is in the previous trace's totals, still in its slot)
  const bool publish_here = assume && mirror != nullptr &&
                            (assume == 2 || assume == 6 || batch_last ||
                             (assume == 3 && (gen[g].n_live == 0 || gen[g].n_carry == 0)));
  if (assume) {
    if (threadIdx.x == 0) {
      const int64_t mine = (n - (int64_t)tile * PRT_BLOCK) < PRT_BLOCK ? (n - (int64_t)tile * PRT_BLOCK) : PRT_BLOCK;
      bool holds = (int64_t)agg_live == mine && (int64_t)agg_carry == (assume == 1 ? mine : 0);
      if (assume >= 5) {  // (thread 0 wrote s_excl itself, above)
        const unsigned before = s_excl[0];
        const int64_t alive = mine - (int64_t)s_excl[1];
        holds = (int64_t)agg_live == alive && (int64_t)agg_carry == (assume == 5 ? alive : 0);
        s_excl[0] = (unsigned)tile * PRT_BLOCK - before;
        s_excl[1] = assume == 5 ? (unsigned)tile * PRT_BLOCK - before : 0u;
      }
      // a tile of a launch that keeps its absorbed rays notes how many it kept (generation 0 has nobody to empty
      // its list for it: it keeps rays without noting them, and the generation behind it compacts by look-back)
      if (agg_kept && g > 0) {
        DeadList* dead = dead_list(gen, g);
        const unsigned at = atomicAdd(&dead->count, 1u);
        if (at < (unsigned)kDeadListCap) dead->entry[at] = ((unsigned)tile << 9) | agg_kept;
      }
      if (assume == 3) {  // the previous trace's record of this tile
        const TileHint h = tile_hints(gen, g)[tile];
        holds = agg_live == h.live && agg_carry == h.carry && n == *reinterpret_cast<const int64_t*>(gen[g].pad);
        s_excl[0] = h.excl_live;
        s_excl[1] = h.excl_carry;
      }
      if (!holds && !failed)  // in place before this tile checks in below
        raise_verdict(&ctrl->error, assume == 3 ? PRT_ERR_TILE_HINT : PRT_ERR_SPECULATION);
      finisher = tile == 0;
      if (publish_here) {
        // Two levels (64 tiles to a counter, the last of each on to the root): thousands of increments of
        // one address would serialise for longer than the generation runs.  Relaxed on purpose: an
        // agent-scope release / acquire writes back and invalidates this XCD's L2 -- per tile, that tripled
        // the generation's run time -- and the words involved (counters, error) are only ever touched
        // by agent-scope atomics, which meet at the memory side like the look-back's status words.
        const unsigned long long last = (unsigned long long)((n - 1) / PRT_BLOCK), group = (unsigned long long)tile >> 6;
        const unsigned long long members = group == (last >> 6) ? (last & 63) + 1 : 64;
        bool done = __hip_atomic_fetch_add(tiles_cur + (group << 6), 1ull, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT) == members - 1;
        if (done && (last >> 6) > 0)
          done = __hip_atomic_fetch_add(tiles_cur + 32, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                 (last >> 6);
        finisher = done;
      }
    }
  } else if (threadIdx.x == 0 && tile > 0) {
    tile_store(tiles_cur + tile, tile_pack(TILE_AGGREGATE, agg_live, agg_carry));
  }
  // Shade BEFORE asking where the results go: the predecessors get this long to publish their
  // aggregates, so the look-back below mostly finds them ready instead of waiting.
  // Record columns that are known before the shading (metadata, segment start) are parked in
  // the lane's own LDS slots -- the hit lists are dead by now -- instead of being held in
  // registers across the fp64-heavy shading.
  Shaded sh;
  sh.err = 0;
  STAMP(4);
  if (live) {
    Meta5 meta = {0, 0, 0, 0, 0};  // the shading reads wavelength and index only; the rest stays parked
    meta.wavelength = PARK(2);
    meta.index = PARK(3);
    PARK(5) = r.ox; PARK(6) = r.oy; PARK(7) = r.oz;
    sh = interact_compute(scene, meta, r, t, prim);
  }
  STAMP(5);
  const int64_t last_tile = (n - 1) / PRT_BLOCK;
  int64_t excl_live, excl_carry;
  if (assume == 3 || assume >= 5) {
    __syncthreads();
    excl_live = s_excl[0];
    excl_carry = s_excl[1];
  } else if (assume) {
    excl_live = (int64_t)tile * PRT_BLOCK;
    excl_carry = assume == 1 ? (int64_t)tile * PRT_BLOCK : 0;
  } else {
  if (wave == 0) {
    unsigned e_live, e_carry;
    bool ok = lookback(tiles_cur, tile, agg_live, agg_carry, e_live, e_carry, &ctrl->error);
    // test hook: pretend the spin expired (such traces publish through k_fused_reinit behind the batch)
    if (ctrl->pad == 1 && tile == 3 && lane == 0) raise_verdict(&ctrl->error, PRT_ERR_STALL);
    (void)ok;
    if (lane == 0) {
      s_excl[0] = e_live; s_excl[1] = e_carry;
      if (g < kTileHintGenerations) tile_hints(gen, g)[tile] = TileHint{e_live, e_carry, agg_live, agg_carry};
    }
  }
  __syncthreads();
  excl_live = s_excl[0];
  excl_carry = s_excl[1];
  finisher = tile == last_tile;
  if (finisher && threadIdx.x == 0) *reinterpret_cast<int64_t*>(gen[g].pad) = n;  // the ray count the tile records belong to
  }
  STAMP(6);

  if (finisher && threadIdx.x == 0) {  // totals are known here: hand over to g + 1
    // (dense mode: the totals are the assumption itself; if it failed the error word says so)
    // (assume == 3: the totals of the previous trace's generation g are still in its slot)
    // (assume 5 / 6: every ray but the dead ones of the list)
    const int64_t total_live = assume == 3 ? gen[g].n_live : assume >= 5 ? n - (int64_t)s_excl[2] : assume ? n : excl_live + agg_live;
    const int64_t total_carry = assume == 3 ? gen[g].n_carry
                                : assume ? (assume == 1 ? n : assume == 5 ? n - (int64_t)s_excl[2] : 0) : excl_carry + agg_carry;
    const int64_t next_in = (total_live == 0) ? 0 : total_carry;
    gen[g].n_live = total_live;
    gen[g].n_carry = total_carry;
    gen[g + 1].n_in = next_in;
    gen[g + 1].row_base = row_base + total_live;
    // (the end of the trace: launches queued blind behind it must find empty generations, whatever an
    // earlier trace left in their slots)
    if (next_in == 0)
      for (int k = g + 2; k <= generation_limit; ++k) gen[k].n_in = 0;
    if (mirror) {
      // The host is told from here, not by a copy after the launch: the counts of this generation go to
      // host-mapped memory, and the generation that ends the trace (or the batch) raises the epoch word
      // the host spins on.  Errors that can still be raised after this point are ruled out by the
      // caller (scenes with untracable materials are published by k_fused_reinit behind the batch).
      mirror->gen[mirror_slot].n_in = n;
      mirror->gen[mirror_slot].row_base = row_base;
      mirror->gen[mirror_slot].n_live = total_live;
      mirror->gen[mirror_slot].n_carry = total_carry;
      if (next_in == 0 || batch_last) {
        mirror->gen[mirror_slot + 1].n_in = next_in;
        int err = __hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err == 0 && row_base + total_live > ld_rows) err = PRT_ERR_ROWS_CAP;
        mirror->error = err;
        __threadfence_system();
        __hip_atomic_store(&mirror->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (row_base + excl_live + agg_live > ld_rows) {  // uniform per workgroup (ld_rows == rows_cap)
    if (threadIdx.x == 0 && !failed) raise_error(&ctrl->error, PRT_ERR_ROWS_CAP);
    return;
  }
  if (!live) return;
  // un-park the early columns
  sh.generation = PARK(0); sh.intensity = PARK(1); sh.wavelength = PARK(2);
  sh.index_in = PARK(3); sh.id = PARK(4);
  r.ox = PARK(5); r.oy = PARK(6); r.oz = PARK(7);
  if (!interact_store_rows<COMPACT>(sh, r, carry, (unsigned)live_rank * 8u, (unsigned)carry_rank * 8u,
                                    next + uniform64(excl_carry), ld_next, rows + uniform64(row_base + excl_live),
                                    ld_rows, next_generation, relaunch, ray_offset))
    atomicExch(&ctrl->error, PRT_ERR_FULL_ROWS);
  if (sh.err) raise_error(&ctrl->error, sh.err);
  STAMP(7);
}


// start of a fused trace: clear the control slots and tile buffer 0
__global__ void k_fused_init(FusedCtrl* ctrl, GenCtrl* gen, int n_gen_slots,
                             unsigned long long* tiles0, int64_t n_tiles, int64_t n, int test_stall) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = i; k < n_tiles; k += stride) tiles0[k] = TILE_INVALID;
  for (int64_t k = i; k < n_gen_slots; k += stride) {
    GenCtrl z;
    memset(&z, 0, sizeof(z));
    if (k == 0) z.n_in = n;
    gen[k] = z;
  }
  if (i == 0) { ctrl->error = 0; ctrl->pad = test_stall; ctrl->rows_cap = 0; }
}

// Behind a batch whose generation kernels could not tell the host themselves (scenes that can raise
// PRT_ERR_UNTRACABLE at store time, the look-back test hook): publish the counts of slots [first_slot,
// first_slot + count) and the error word, and if the batch ended the trace (the slot behind it holds no
// rays, the generation limit is reached, or an error is set) clear the control words for the next trace
// of the same shape.  Traces that publish from their generation kernels never run this kernel.
__global__ void k_fused_reinit(FusedCtrl* ctrl, GenCtrl* gen, int n_gen_slots, int end_slot, int limit,
                               unsigned long long* tiles0, int64_t n_tiles, int64_t n, int test_stall,
                               HostMirror* mirror, unsigned long long epoch, int first_slot, int count) {
  const bool over = end_slot >= limit || gen[end_slot].n_in == 0 || ctrl->error != 0;
  if (mirror) {
    const int k = threadIdx.x;
    if (k < count) {
      mirror->gen[k].n_in = gen[first_slot + k].n_in;
      mirror->gen[k].row_base = gen[first_slot + k].row_base;
      mirror->gen[k].n_live = gen[first_slot + k].n_live;
      mirror->gen[k].n_carry = gen[first_slot + k].n_carry;
    }
    if (k == 0) mirror->error = ctrl->error;
    __threadfence_system();
    __syncthreads();
    if (k == 0) __hip_atomic_store(&mirror->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!over) return;
  // every block must see the same `over`: the slot is rewritten below, so sync through a second launch
  // is avoided by letting block 0 alone touch the slots and only after all blocks have read them --
  // simplest: one block does everything
  for (int64_t k = threadIdx.x; k < n_tiles; k += blockDim.x) tiles0[k] = TILE_INVALID;
  __syncthreads();
  for (int k = threadIdx.x; k < n_gen_slots; k += blockDim.x) {
    GenCtrl z;
    memset(&z, 0, sizeof(z));
    if (k == 0) z.n_in = n;
    gen[k] = z;
  }
  if (threadIdx.x == 0) { ctrl->error = 0; ctrl->pad = test_stall; ctrl->rows_cap = 0; }
}

// component.intersect(): run one component's program and spill its list
__global__ void __launch_bounds__(PRT_BLOCK)
k_intersect(SceneDev scene, Operand result, const double* __restrict__ rays, int64_t ld, int64_t n,
            double* __restrict__ hits, int64_t* __restrict__ ids, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  const LaneLists lists = lane_lists(scene.lds_slots);
  const Ray8 r = load_ray8(rays, ld, i);
  // reuse the interpreter: without an I_ROOT it only builds the lists
  Pair ra = {PRT_INF, PRT_INF, -1}, rb = {PRT_INF, PRT_INF, -1};
  for (int pc = 0; pc < scene.n_instr; ++pc) {
    const DevInstr step = scene.code[pc];
    const DevInstr* in = &step;
    if (in->kind == I_LEAF) {
      double t0, t1;
      surface_pair(in->type, in->data, in->data + 6, r, t0, t1);
      if (in->a1 == OPER_REGA) { ra.t0 = t0; ra.t1 = t1; ra.prim = in->a0; }
      else if (in->a1 == OPER_REGB) { rb.t0 = t0; rb.t1 = t1; rb.prim = in->a0; }
      else { lists.put(in->a2, t0, in->a0); lists.put(in->a2 + 1, t1, in->a0); }
    } else if (in->kind == I_CSG) {
      bool is_root;
      double t_unused;
      int prim_unused;
      csg_step(in, r, lists, ra, rb, is_root, t_unused, prim_unused);
    }
  }
  for (int k = 0; k < result.len; ++k) {
    const double v = operand_t(result, lists, ra, rb, k);
    const int p = operand_id(result, lists, ra, rb, k);
    hits[k * ld_out + i] = v;
    ids[k * ld_out + i] = (is_finite(v) && p >= 0) ? (int64_t)scene.prims[p].surface_id : -1;
  }
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_normals(const DevPrim* __restrict__ prim, const double* __restrict__ pts, int64_t ld, int64_t k,
          double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= k) return;
  double nx, ny, nz;
  world_normal(prim, pts[i], pts[ld + i], pts[2 * ld + i], pts[3 * ld + i], nx, ny, nz);
  out[i] = nx;
  out[ld + i] = ny;
  out[2 * ld + i] = nz;
  out[3 * ld + i] = 0.0 * (double)prim->normal_scale;
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_material_trace(const DevPrim* __restrict__ prim, double* __restrict__ rays, int64_t ld, int64_t k) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= k) return;
  const Ray8 r = load_ray8(rays, ld, i);
  double dx = r.dx, dy = r.dy, dz = r.dz, dw = r.dw, index = rays[11 * ld + i];
  const double len3 = norm3(dx, dy, dz);
  shade(prim, r.ox, r.oy, r.oz, r.ow, dx, dy, dz, dw, rays[10 * ld + i], index, dx / len3, dy / len3,
        dz / len3);
  rays[4 * ld + i] = dx;
  rays[5 * ld + i] = dy;
  rays[6 * ld + i] = dz;
  rays[7 * ld + i] = dw;
  rays[11 * ld + i] = index;
}

#include "prt_sources.hpp"
#include "prt_render.hpp"
#include "prt_ops.hpp"

// deep CSG trees may need more than the default 64 KiB of dynamic LDS per workgroup
static int raise_lds_limits() {
  HIP_TRY(hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_render_hits, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_hit_lanes<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_generation<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  HIP_TRY(hipFuncSetAttribute((const void*)k_intersect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  return PRT_OK;
}

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

// workspace of prt_trace:
//   ctrl | fused ctrl | generation slots | rows_per_generation (device) | tile words A | B
//   | block counts | block offsets | hit_t (n f64) | hit_prim (n i32)      [unfused path only]
//   | ray buffer A (13 n) | ray buffer B (13 n)
struct TraceLayout {
  size_t ctrl, fctrl, gen, dead_lists, tile_hints, gen_rows, tiles_a, tiles_b, counts, offsets, hit_t, hit_prim, rays_a,
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
  l.tile_hints = take((size_t)kTileHintGenerations * nb * sizeof(TileHint));  // (at kTileHintOffset from gen[0]: tile_hints())
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
  if (l.tile_hints - l.gen != kTileHintOffset || l.dead_lists - l.gen != kDeadListOffset) abort();  // (the kernel finds them by these constants)
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
  HP(3);
  const bool culls = s->has_cull_steps;
  auto kernel = t->compact ? (culls ? k_generation<true, true> : k_generation<false, true>)
                           : (culls ? k_generation<true, false> : k_generation<false, false>);
  for (int b = 0; b < t->batch; ++b) {
    const int gg = t->g + b;
    const double* src = (gg == 0) ? t->rays : buf[(gg - 1) & 1];
    const int64_t src_ld = (gg == 0) ? t->ld : n;
    // dense-mode hint of the previous trace for this generation (the kernel reads the generation's
    // ray count on the device and checks the assumption tile by tile)
    int assume = 0;
    if (t->use_hints && gg < (int)s->hint_mode.size()) assume = s->hint_mode[gg];
    // (a generation whose dense hint was refuted lately keeps compacting for a while: see hint_rest)
    if (assume && gg < (int)s->hint_rest.size() && s->hint_rest[gg] > 0) assume = 0;
    // a generation that compacts on the per-tile record its last run left in this workspace (exact, and it leaves
    // nothing behind: preferred to mode 4 below when the trace is of the same ray buffer again)
    const bool on_record = (assume == 0 || assume == 4) && t->use_tile_records && !t->records_off &&
                           gg < kTileHintGenerations && t->tile_record[gg];
    const bool kept_then = t->launch_mode[gg] == 4;  // in the ticket's last trace (meaningful where it left records)
    // Mode 4 (sparse loss): last time every ray of this generation was recorded and all but a few of them -- absorbed
    // ones -- were carried on.  Compacting those few away costs every tile a look-back, and the tiles that hold such a
    // ray are as a rule the slow ones (a ray that misses the part it was expected to hit visits the parts behind it):
    // with 1 tile in 60 slow and 1280 in flight, every tile waits for a straggler (profiles/r4/lookback_stragglers.txt).
    // Such a generation runs dense instead, keeping its absorbed rays the way upstream does (PRT_TRACE_KEEP_ABSORBED
    // for this launch only: _pyrayt.py:415-428 carries them, direction zeroed, and drops them a generation later);
    // the next generation finds them dead on arrival, records nothing for them and drops them when it compacts.
    int keep = keep_absorbed;
    if (assume == 4) {
      // (a trace of the ticket's last ray buffer without a record of this generation -- it ran dense last time --
      // compacts once to leave one: from then on such traces run on records, which leave the next generation alone)
      const bool wants_record = t->use_tile_records && gg < kTileHintGenerations;
      if (on_record || wants_record || (t->flags & PRT_TRACE_NO_SPARSE_KEEP)) assume = 0;
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
    // (a generation that keeps its absorbed rays this time and did not last time, or the other way round, hands the
    // generations behind it other rays than their records were taken on)
    if ((t->launch_mode[gg] == 4) != kept_then) t->records_off = true;
    s->dense_launches += assume ? 1 : 0;
    s->sparse_keep_launches += keep && !keep_absorbed ? 1 : 0;
    if (on_record && assume == 0) {
      assume = 3;
      t->launch_mode[gg] = 3;
      t->used_tile_hints = true;
      s->tile_hint_launches += 1;
    }
    // launches, not generations, alternate between the two status buffers, across traces too: every
    // launch works on the one the launch before it left clean and cleans the other (k_generation)
    hipLaunchKernelGGL(kernel, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds, st, sd, src, src_ld, buf[gg & 1], n,
                       t->rows_out, t->rows_cap, ctrl, gen, gg, tiles[t->flip], tiles[t->flip ^ 1], (double)(gg + 1),
                       t->limit, t->ray_offset, keep,
                       t->publish_in_kernel ? t->mirror_dev : (HostMirror*)nullptr, epoch, b,
                       b + 1 == t->batch ? 1 : 0, assume);
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
    for (bool& kept : t->tile_record) kept = false;  // (the slots the records are checked against are cleared)
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
  // The per-tile records of this ticket's last trace (TileHint) serve the generations that compact: offered with
  // the other hints, to a trace that publishes from its kernels (the slots the records lean on are then never
  // cleared between traces), and rested after a miss like them.
  // ... and only to a trace of the very buffer the records were taken from: another ray set loses its rays in other
  // tiles, every offer would be a miss and a repeat (a caller that refills one buffer with new rays is still offered
  // them -- and every tile checks)
  bool allow_tiles = t->allow_tile_hints && t->use_hints && t->publish_in_kernel && t->record_rays == t->rays;
  if (allow_tiles && s->tile_hint_holdoff > 0) {
    s->tile_hint_holdoff -= 1;
    allow_tiles = false;
  }
  t->use_tile_records = allow_tiles;
  t->records_off = false;
  t->used_tile_hints = false;
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
    }
    if (!done && host_gen[t->batch].n_in == 0) done = true;
    t->g += t->batch;
    if (done || t->g >= t->limit) break;
    t->batch = std::min(kGenerationBatch, t->limit - t->g);
    rc = fused_launch_batch(s, c, t);
    if (rc) return rc;
  }
  t->launched = false;
  if (error) for (bool& kept : t->tile_record) kept = false;  // (whatever the attempt overwrote before it failed)
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
  if (error == PRT_ERR_SPECULATION || error == PRT_ERR_FULL_ROWS || error == PRT_ERR_TILE_HINT) return error;
  // a record block that looked too small to a generation launched on a hint may only have been too small
  // for the hint: the caller repeats without hints before it reports it
  if (error == PRT_ERR_ROWS_CAP && t->used_tile_hints) return PRT_ERR_TILE_HINT;
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
  if (t->used_tile_hints) s->tile_hint_misses_in_a_row = 0;
  // which generations left (or confirmed) a per-tile record in this workspace: those that compacted
  // (a trace that published through k_fused_reinit had its generation slots -- the ray count and totals the records
  // are checked against -- cleared behind it: its records are not offered)
  for (int g = 0; g < kTileHintGenerations; ++g)
    t->tile_record[g] = t->publish_in_kernel && g < t->n_seen && (t->launch_mode[g] == 0 || t->launch_mode[g] == 3) &&
                        (t->seen_mode[g] == 0 || t->seen_mode[g] == 4);
  t->record_rays = t->rays;
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
  int rc = on_device(s, device, &c);
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
  rc = settle_timing(t);  // events of the ticket's previous trace, before they are recorded again
  if (rc) return rc;
  t->rays = rays; t->n = n; t->ld = ld; t->limit = generation_limit; t->ray_offset = ray_offset;
  t->rows_out = rows_out; t->rows_cap = rows_cap; t->w = (char*)workspace; t->flags = flags;
  t->st = (hipStream_t)stream;
  t->launched = false;
  t->allow_hints = !(flags & PRT_TRACE_NO_HINTS);
  t->allow_tile_hints = t->allow_hints && !(flags & PRT_TRACE_NO_TILE_RECORDS);
  t->compact = !s->full_rows && !(flags & PRT_TRACE_FULL_ROWS);
  t->active = true;
  if (n == 0 || generation_limit == 0) { reset_stats(s, t, PRT_VARIANT_FUSED); return PRT_OK; }
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
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (ticket < 0 || ticket >= PRT_TRACE_TICKETS || !rows_per_generation)
    return fail(PRT_ERR_ARG, "bad ticket / null rows_per_generation");
  TraceTicket* t = &c->ticket[ticket];
  if (!t->active) return fail(PRT_ERR_ARG, "no trace in flight on this ticket");
  t->active = false;
  s->stats_device = device;
  s->stats_ticket = ticket;
  for (int g = 0; g < t->limit; ++g) rows_per_generation[g] = 0;
  if (t->n == 0 || t->limit == 0) return 0;
  int64_t rc64 = 0;
  if (t->flags & PRT_TRACE_UNFUSED) {
    rc64 = trace_unfused(s, c, t, rows_per_generation);
  } else {
    for (int attempt = 0;; ++attempt) {
      rc64 = fused_finish(s, c, t, rows_per_generation);
      if ((rc64 != PRT_ERR_SPECULATION && rc64 != PRT_ERR_FULL_ROWS && rc64 != PRT_ERR_TILE_HINT) || attempt == 4) break;
      if (rc64 == PRT_ERR_TILE_HINT) {
        // a generation found other counts in a tile than its last run left on record (other rays, or a scene
        // that treats them differently): again without the records -- the dense hints stay, they were not refuted
        s->tile_hint_misses += 1;
        s->tile_hint_misses_in_a_row = std::min(s->tile_hint_misses_in_a_row + 1, 6);
        s->tile_hint_holdoff = 1 << s->tile_hint_misses_in_a_row;
        t->allow_tile_hints = false;
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
    if (rc64 == PRT_ERR_SPECULATION || rc64 == PRT_ERR_FULL_ROWS || rc64 == PRT_ERR_TILE_HINT)
      rc64 = fail(PRT_ERR_HIP, "trace kept failing its own assumptions");
    if (rc64 == PRT_ERR_STALL) {  // never observed outside the test hook; see lookback()
      for (int g = 0; g < t->limit; ++g) rows_per_generation[g] = 0;
      rc = settle_timing(t);
      if (rc) return rc;
      s->lookback_fallbacks += 1;  // telemetry: a box that falls back silently would just look 2x slow
      reset_stats(s, t, PRT_VARIANT_UNFUSED);
      rc64 = trace_unfused(s, c, t, rows_per_generation);
    }
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
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (count < 0 || (count && !jobs) || depth < 1 || depth > PRT_TRACE_TICKETS || !workspaces)
    return fail(PRT_ERR_ARG, "bad job list / depth out of range (1..PRT_TRACE_TICKETS) / null workspaces");
  for (int64_t k = 0; k < count; ++k) {
    if (!jobs[k].rows_per_generation) return fail(PRT_ERR_ARG, "a job has no rows_per_generation");
    jobs[k].total = 0;
  }
  for (int k = 0; k < depth; ++k)
    if (c->ticket[k].active) return fail(PRT_ERR_ARG, "a ticket this batch needs has a trace in flight");
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
      rc = prt_trace_begin(s, device, lane, job.rays, job.n, job.ld, generation_limit, ray_offset, job.rows_out,
                           job.rows_cap, workspaces[lane], flags, streams ? streams[lane] : nullptr);
      if (rc) { jobs[k].total = rc; first_error = rc; message = g_error; }
    }
  }
  if (first_error) return fail((int)first_error, message.c_str());
  return sum;
}


extern "C" int prt_trace_telemetry(const prt_scene* s, int64_t* out12) {
  int64_t* out8 = out12;
  if (!s || !out8) return fail(PRT_ERR_ARG, "null argument");
  out12[8] = s->tile_hint_launches;
  out12[9] = s->tile_hint_misses;
  out12[10] = s->sparse_keep_launches;
  out12[11] = 0;
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

// ------------------------------------------------------------------------------------------------
// renderers
// ------------------------------------------------------------------------------------------------
static int camera_of(const prt_camera* cam, DevCamera* out) {
  if (!cam) return fail(PRT_ERR_ARG, "camera is null");
  if (cam->h_pixels < 0 || cam->v_pixels < 0) return fail(PRT_ERR_ARG, "negative camera resolution");
  std::memcpy(out->world, cam->world, sizeof(out->world));
  out->h_pixels = cam->h_pixels; out->v_pixels = cam->v_pixels;
  out->h_width = cam->h_width; out->v_width = cam->v_width;
  return PRT_OK;
}

extern "C" int prt_camera_rays(int device, const prt_camera* camera, int64_t first, int64_t count,
                               double* rays_out, int64_t ld, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  DevCamera cam;
  int rc = camera_of(camera, &cam);
  if (rc) return rc;
  if (first < 0 || count < 0 || first + count > cam.h_pixels * cam.v_pixels || ld < count ||
      (count && !rays_out))
    return fail(PRT_ERR_ARG, "bad pixel range / output buffer");
  if (count == 0) return PRT_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_camera, dim3(blocks_for(count)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, cam,
                     first, count, rays_out, ld);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

static SceneDev render_scene_dev(const prt_scene* s, const DeviceCopy* c) {
  return SceneDev{c->prims, c->render_code, (int)s->render_program.code.size(), s->render_program.lds_slots};
}

extern "C" int prt_render_hits(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                               double* t_out, int64_t* surf_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld < n || (n && (!rays || !t_out || !surf_out)))
    return fail(PRT_ERR_ARG, "bad ray / output buffers");
  if (n == 0) return PRT_OK;
  SceneDev sd = render_scene_dev(s, c);
  hipLaunchKernelGGL(k_render_hits, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds_bytes(sd.lds_slots),
                     (hipStream_t)stream, sd, rays, ld, n, t_out, surf_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_gooch_shade(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                               const double* t, const int64_t* surf, const double* gooch,
                               const double* light, double* rgba_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld < n || !light || (n && (!rays || !t || !surf || !gooch || !rgba_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (n == 0) return PRT_OK;
  hipLaunchKernelGGL(k_gooch, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, c->prims,
                     (int)s->prims.size(), rays, ld, n, t, surf, gooch, light[0], light[1], light[2],
                     rgba_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_gooch_mix(int device, const double* points, const double* normals, int64_t n,
                             int64_t ld, const double* shade, const double* light, double* rgba_out,
                             int64_t ld_out, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  if (n < 0 || ld < n || ld_out < n || !shade || !light || (n && (!points || !normals || !rgba_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (n == 0) return PRT_OK;
  HIP_TRY(hipSetDevice(device));
  GoochShade g;
  std::memcpy(g.warm, shade, sizeof(g.warm));
  std::memcpy(g.cool, shade + 4, sizeof(g.cool));
  hipLaunchKernelGGL(k_gooch_mix, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, points,
                     normals, ld, n, g, light[0], light[1], light[2], rgba_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_render(prt_scene* s, int device, const prt_camera* camera, int64_t first,
                          int64_t count, const double* gooch, const double* light, double* rgba_out,
                          double* t_out, int64_t* surf_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  DevCamera cam;
  rc = camera_of(camera, &cam);
  if (rc) return rc;
  if (first < 0 || count < 0 || first + count > cam.h_pixels * cam.v_pixels)
    return fail(PRT_ERR_ARG, "bad pixel range");
  if (rgba_out && (!gooch || !light)) return fail(PRT_ERR_ARG, "shading needs the gooch table and a light");
  if (!rgba_out && !t_out && !surf_out) return fail(PRT_ERR_ARG, "no output requested");
  if (count == 0) return PRT_OK;
  SceneDev sd = render_scene_dev(s, c);
  const double lx = light ? light[0] : 0.0, ly = light ? light[1] : 0.0, lz = light ? light[2] : 0.0;
  hipLaunchKernelGGL(k_render, dim3(blocks_for(count)), dim3(PRT_BLOCK), lds_bytes(sd.lds_slots),
                     (hipStream_t)stream, sd, cam, first, count, gooch, lx, ly, lz, rgba_out, t_out,
                     surf_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int64_t prt_edge_workspace_bytes(int64_t h_pixels, int64_t v_pixels) {
  if (h_pixels < 0 || v_pixels < 0) return PRT_ERR_ARG;
  return (int64_t)align_up((size_t)(h_pixels * v_pixels) + 1, 256);
}

extern "C" int prt_edge_canvas(int device, const int64_t* surf, int64_t h_pixels, int64_t v_pixels,
                               int rings, double* rgba_out, void* workspace, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  if (h_pixels < 0 || v_pixels < 0 || rings < 0) return fail(PRT_ERR_ARG, "bad picture size");
  const int64_t n = h_pixels * v_pixels;
  if (n == 0) return PRT_OK;
  if (!surf || !rgba_out || !workspace) return fail(PRT_ERR_ARG, "null buffer");
  HIP_TRY(hipSetDevice(device));
  unsigned char* seed = (unsigned char*)workspace;
  hipLaunchKernelGGL(k_edge_seed, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, surf,
                     h_pixels, v_pixels, seed);
  hipLaunchKernelGGL(k_edge_canvas, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream,
                     (const unsigned char*)seed, h_pixels, v_pixels, rings, rgba_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ------------------------------------------------------------------------------------------------
// tinygfx/g3d/operations.py as entry points
// ------------------------------------------------------------------------------------------------
static int ops_device(int device) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  return PRT_OK;
}

extern "C" int prt_reflect(int device, const double* vectors, const double* normals, int rows, int64_t n,
                           int64_t ld, double* out, int64_t ld_out, void* stream) {
  if (rows < 1 || rows > 4 || n < 0 || ld < n || ld_out < n || (n && (!vectors || !normals || !out)))
    return fail(PRT_ERR_ARG, "bad buffers (vectors of 1..4 components)");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_reflect, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, vectors, normals,
                     rows, ld, n, out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_refract(int device, double* vectors, const double* normals, const double* n1,
                           const double* n2, double n_global, int rows, int64_t n, int64_t ld, double* out,
                           int64_t ld_out, double* index_out, void* stream) {
  if (rows < 1 || rows > 4 || n < 0 || ld < n || ld_out < n ||
      (n && (!vectors || !normals || !n1 || !n2 || !out || !index_out)))
    return fail(PRT_ERR_ARG, "bad buffers (vectors of 1..4 components)");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_refract, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, vectors, normals,
                     n1, n2, n_global, rows, ld, n, out, ld_out, index_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_binomial_root(int device, const double* a, const double* b, const double* c, int64_t n,
                                 double* roots_out, int64_t ld_out, void* stream) {
  if (n < 0 || ld_out < n || (n && (!a || !b || !c || !roots_out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_binomial_root, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, a, b, c, n,
                     roots_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_smallest_positive_root(int device, const double* a, const double* b, const double* c,
                                          int64_t n, double* out, void* stream) {
  if (n < 0 || (n && (!a || !b || !c || !out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_smallest_positive_root, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, a,
                     b, c, n, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_dot(int device, const double* m1, const double* m2, int64_t reduce_len,
                       int64_t reduce_stride, int64_t out_len, int64_t out_stride, double* out, void* stream) {
  if (reduce_len < 0 || out_len < 0 || (out_len && (!m1 || !m2 || !out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || out_len == 0) return rc;
  hipLaunchKernelGGL(k_dot, dim3(blocks_for(out_len)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, m1, m2,
                     reduce_len, reduce_stride, out_len, out_stride, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_array_csg(int device, const double* left, int m_left, const double* right, int m_right,
                             int64_t n, int64_t ld, int op, int sort_output, double* out, int64_t ld_out,
                             void* stream) {
  if (m_left < 0 || m_right < 0 || (m_left & 1) || (m_right & 1))
    return fail(PRT_ERR_ARG, "hit lists hold enter/exit pairs: an even number of rows each");
  if (op < PRT_NODE_UNION || op > PRT_NODE_DIFFERENCE) return fail(PRT_ERR_ARG, "operation is invalid");
  if (n < 0 || ld < n || ld_out < n || (n && ((m_left && !left) || (m_right && !right) || !out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0 || m_left + m_right == 0) return rc;
  hipLaunchKernelGGL(k_array_csg, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, left, m_left,
                     right, m_right, ld, n, op, sort_output, out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

static int primitive_args(int type, const double* params, PrimParams* out) {
  if (type < PRT_PRIM_SPHERE || type > PRT_PRIM_PARABOLOID) return fail(PRT_ERR_ARG, "unknown primitive type");
  if (!params) return fail(PRT_ERR_ARG, "params is null");
  std::memcpy(out->q, params, sizeof(out->q));
  return PRT_OK;
}

extern "C" int prt_primitive_intersect(int device, int type, const double* params, const double* rays,
                                       int64_t n, int64_t ld, double* hits_out, int64_t ld_out, void* stream) {
  PrimParams q;
  int rc = primitive_args(type, params, &q);
  if (rc) return rc;
  if (n < 0 || ld < n || ld_out < n || (n && (!rays || !hits_out))) return fail(PRT_ERR_ARG, "bad buffers");
  rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_primitive_intersect, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, type, q,
                     rays, ld, n, hits_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_primitive_normal(int device, int type, const double* params, const double* points,
                                    int64_t n, int64_t ld, double* normals_out, int64_t ld_out, void* stream) {
  PrimParams q;
  int rc = primitive_args(type, params, &q);
  if (rc) return rc;
  if (n < 0 || ld < n || ld_out < n || (n && (!points || !normals_out))) return fail(PRT_ERR_ARG, "bad buffers");
  rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_primitive_normal, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, type, q,
                     points, ld, n, normals_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ------------------------------------------------------------------------------------------------
// frame re-assembly across ranks (RCCL all-gather + placement kernel)
// ------------------------------------------------------------------------------------------------
#include "prt_gather.hpp"

// ------------------------------------------------------------------------------------------------
// reductions over the result frame
// ------------------------------------------------------------------------------------------------
#include "prt_frame.hpp"
