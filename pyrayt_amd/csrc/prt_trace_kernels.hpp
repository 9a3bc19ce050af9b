// prt_trace_kernels.hpp -- the device side of the trace path: k_generation (fused PROPAGATE + INTERACT + record in one
// launch, with its look-back / dense / per-tile-record / dead-list compaction), the three-kernel path k_hit / k_scan /
// k_shade / k_advance (fallback, A/B partner and the halves behind prt_propagate / prt_interact), the k-lanes-per-ray
// nearest-hit kernels, and the per-object kernels k_intersect / k_normals / k_material_trace.  Included by
// prt_kernels.hip (one translation unit); the per-ray arithmetic these kernels call is prt_device.hpp.
#pragma once
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
// Dead lists: a generation launched dense with its absorbed rays kept (hint mode 4) notes, per tile that kept any,
// (tile << 9 | how many); the generation behind it, launched on that list (assume 5 / 6), takes its tiles' offsets
// from "tile index x tile size minus the dead rays in front" -- no look-back for a handful of dead rays.  Three
// lists in rotation: generation g writes list g % 3, g + 1 reads it, and every generation empties list (g + 1) % 3.
static const int kDeadListCap = 1020;
struct DeadList { unsigned count, pad[3], entry[kDeadListCap]; };
static_assert(sizeof(DeadList) == 4096, "dead list sizing");
static const size_t kDeadListOffset =
    ((size_t)(kMaxGenerationSlots + 2) * sizeof(GenCtrl) + 255) / 256 * 256 - sizeof(GenCtrl);  // from gen[0], see trace_layout
__device__ __forceinline__ DeadList* dead_list(GenCtrl* gen, int g) {
  return reinterpret_cast<DeadList*>(reinterpret_cast<char*>(gen) + kDeadListOffset) + (g % 3);
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

// ---- record plans (round 6; include/prt.h prt_record_plan) ------------------------------------------------------------
// What a trace records, when the caller does not want every row.  The reference appends one row per live ray and
// generation (pyrayt/_pyrayt.py:168-186) and its users drop most of them on the next line -- `results.loc[results[
// 'surface'] == imager.get_id()]` (examples/lens_design.ipynb cells 11, 19, 38), the rows of the last generation (cells
// 12, 15, 20) -- to look at a spot size or a focus.  A plan lets the generation kernel do that while the row is still in
// registers: rows of other surfaces than the listed ones are not stored (the 15 stores are 52 % of what a config-2
// trace moves), and / or the per-group sums k_frame_reduce / k_frame_mean_square would compute from the stored frame
// are accumulated right here, per generation (the caller picks the generation -- "the last one" -- afterwards).
// One of these per ticket, in device memory the library owns; the PLAN instantiations of k_generation read it through
// the constant address space (scalar loads).  The other instantiations never look at it.
enum { SINK_STATS = 12 };  // [0..8] as k_frame_reduce (prt_frame.hpp); [9] rows with a finite v, [10] sum v, [11] sum v^2 (k_frame_mean_square)
struct PlanDev {
  int32_t n_rec;          // 0: rows of every surface pass; else only those of rec_prims[0 .. n_rec)
  int32_t store_rows;     // the rows that pass are stored in rows_out (0: nothing is stored -- sums only)
  int32_t n_groups;       // > 0: the rows that pass go to the sink
  int32_t slots;          // copies of the sums the waves spread their atomics over (a power of two)
  int32_t rec_prims[8];
  double rays_per_source; // group = floor(id / rays_per_source) (_pyrayt.py:349-354); <= 0: one group
  double* sums;           // (slots, generation_limit, n_groups, SINK_STATS), zeroed by the library before the trace
  const double* pivots;   // (n_groups, 3) subtracted from y1, z1 and the axis intercept before accumulating, or null
  int32_t ms_quantity;    // a frame column 0..14, 15 = the axis intercept x0 - x_tilt y0 / y_tilt, < 0: no mean-square sums
  int32_t ms_transform;   // 0 none, 1 sin
  double ms_about;
  int32_t limit;          // generation_limit the sums were sized for
  int32_t columns;        // which of the 15 record columns a stored row writes (bit k = PRT_COL_k); 0x7fff: all of them
};
typedef const __attribute__((address_space(4))) PlanDev* ConstPlan;

#define PRT_ERR_SPECULATION (-101) /* internal: a generation launched in dense mode was not dense -> host re-runs without hints */
#define PRT_ERR_STALL (-100) /* internal: look-back gave up -> host falls back to the unfused path */
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

// ---- lean segments of the compact state (round 5) ---------------------------------------------------------------
// Three of the ten rows the generations hand on never change on the fused path -- intensity (9), wavelength (10), ray
// id (12): no built-in material touches them -- and in a ray set as sources emit it they are redundant within a wave:
// one intensity, one wavelength, ids that count up by one.  A wave whose 64 outputs land, in lane order, on the 64
// columns that ONE wave of the next generation will read (its tile's first column is tile x 256 and the waves in front
// of it carried every ray: dense launches, and compacting ones as long as nothing has died yet) checks exactly that --
// bit-equal intensity and wavelength, id == id0 + lane with id0 an integer in [0, 2^48), at least three rays -- and if it
// holds, lane 0 alone writes, into the first three entries of the segment's ID row: id0 boxed into a NaN (tag in the
// top 16 bits), the intensity, the wavelength; rows 9 and 10 of the segment are not written at all.  The reader
// fetches those three entries with one scalar load (one line: spread over the three rows they cost config 4 its gain,
// 125 000 x 3 lines per generation through the scalar cache that holds the scene program): tagged -> `id0 + lane` and
// the two values, no vector loads of those rows; anything else -> the rows, as before.  24 B per ray less each way.
// A genuine id that carries the tag would be misread: a ray with such an id makes the trace repeat with all 13 rows
// (PRT_ERR_FULL_ROWS, like a non-trivial w), whose kernels know nothing of this.
#define PRT_LEAN_TAG 0x7ffbll
__device__ __forceinline__ bool lean_tagged(double v) { return (__double_as_longlong(v) >> 48) == PRT_LEAN_TAG; }

// compact: the next state goes without its rows 3, 7 and 8 (see k_generation); returns false if a ray
// that goes on does not have the values the reader will assume for them
// lean: this wave's rows 9, 10 and 12 are written by its first lane alone (see above; `id0` = that lane's id)
template <bool COMPACT>
__device__ __forceinline__ bool interact_store_rows(const Shaded& s, const Ray8& r, bool carry, unsigned row_bytes,
                                                    unsigned next_bytes, double* __restrict__ nxt, int64_t ld_next,
                                                    double* __restrict__ rec, int64_t ld_rows,
                                                    double next_generation, int relaunch, double ray_offset,
                                                    bool lean = false, double id0 = 0.0, bool record = true,
                                                    unsigned columns = 0x7fffu) {
  // (record == false: a PLAN launch whose record plan drops this ray's row; columns: the columns such a plan wants of
  // the rows it keeps (uniform) -- see PlanDev.  The default kernels pass the constants `true` and "all fifteen".)
  if (record) {
#define PRT_STORE_COLUMN(col, value) \
  if (columns & (1u << (col))) row_store<PRT_STORE_AUX_REC>(rec + (col) * ld_rows, row_bytes, (value))
  PRT_STORE_COLUMN(PRT_COL_GENERATION, s.generation);
  PRT_STORE_COLUMN(PRT_COL_INTENSITY, s.intensity);
  PRT_STORE_COLUMN(PRT_COL_WAVELENGTH, s.wavelength);
  PRT_STORE_COLUMN(PRT_COL_INDEX, s.index_in);
  PRT_STORE_COLUMN(PRT_COL_ID, s.id);
  PRT_STORE_COLUMN(PRT_COL_SURFACE, s.surface_id);
  PRT_STORE_COLUMN(PRT_COL_X0, r.ox);
  PRT_STORE_COLUMN(PRT_COL_Y0, r.oy);
  PRT_STORE_COLUMN(PRT_COL_Z0, r.oz);
  PRT_STORE_COLUMN(PRT_COL_X1, s.px);
  PRT_STORE_COLUMN(PRT_COL_Y1, s.py);
  PRT_STORE_COLUMN(PRT_COL_Z1, s.pz);
  PRT_STORE_COLUMN(PRT_COL_XTILT, s.tx);
  PRT_STORE_COLUMN(PRT_COL_YTILT, s.ty);
  PRT_STORE_COLUMN(PRT_COL_ZTILT, s.tz);
#undef PRT_STORE_COLUMN
  }
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
    row_store<PRT_STORE_AUX_NEXT>(nxt + 11 * ld_next, next_bytes, s.index);
    if (COMPACT && lean) {  // (wave-uniform)
      if ((threadIdx.x & 63) == 0) {
        row_store<PRT_STORE_AUX_NEXT>(nxt + 12 * ld_next, next_bytes,
                                      __longlong_as_double((PRT_LEAN_TAG << 48) | (long long)id0));
        row_store<PRT_STORE_AUX_NEXT>(nxt + 12 * ld_next, next_bytes + 8u, s.intensity);
        row_store<PRT_STORE_AUX_NEXT>(nxt + 12 * ld_next, next_bytes + 16u, s.wavelength);
      }
    } else {
      row_store<PRT_STORE_AUX_NEXT>(nxt + 9 * ld_next, next_bytes, s.intensity);
      row_store<PRT_STORE_AUX_NEXT>(nxt + 10 * ld_next, next_bytes, s.wavelength);
      row_store<PRT_STORE_AUX_NEXT>(nxt + 12 * ld_next, next_bytes, s.id);
    }
    // (!relaunch: nobody reads this state)
    if (COMPACT && relaunch) return w_is_trivial(qw, s.dw) && !lean_tagged(s.id);
  }
  return true;
}

// The sink of a record plan: what k_frame_reduce and k_frame_mean_square (prt_frame.hpp) would read back out of the
// stored frame, accumulated while the row is in registers.  Same per-row values (the operands are the very numbers the
// record stores would have written: x0, y0 = the pre-hit origin, x1 .. z1 = the hit point, the unit tilts), summed per
// wave with xor shuffles and added with one atomic per statistic to this generation's block of the caller's sums --
// spread over `slots` copies (prt_sink_fold adds them up): a generation of a million rays is 16 000 waves, and that
// many atomics on twelve words would take longer than the generation.  A wave's rows almost always fall into one
// group (ids ascend along the tile); a wave that straddles groups works them off one after the other.
__device__ __forceinline__ double sink_quantity(int quantity, const Shaded& s, double ox, double oy, double oz) {
  switch (quantity) {  // (uniform)
    case PRT_COL_GENERATION: return s.generation;
    case PRT_COL_INTENSITY: return s.intensity;
    case PRT_COL_WAVELENGTH: return s.wavelength;
    case PRT_COL_INDEX: return s.index_in;
    case PRT_COL_ID: return s.id;
    case PRT_COL_SURFACE: return s.surface_id;
    case PRT_COL_X0: return ox;
    case PRT_COL_Y0: return oy;
    case PRT_COL_Z0: return oz;
    case PRT_COL_X1: return s.px;
    case PRT_COL_Y1: return s.py;
    case PRT_COL_Z1: return s.pz;
    case PRT_COL_XTILT: return s.tx;
    case PRT_COL_YTILT: return s.ty;
    case PRT_COL_ZTILT: return s.tz;
    default: return ox - s.tx * oy / s.ty;  // the axis intercept (k_frame_mean_square's FRAME_AXIS_INTERCEPT)
  }
}
// (called by EVERY lane of the wave -- the xor shuffles need their partners -- with the operands of the lanes that
// have a row to add: its end point y1, z1, its axis intercept, the mean-square quantity already transformed and
// shifted, and the three metadata values)
__device__ __forceinline__ void sink_accumulate(ConstPlan plan, int g, bool sunk, double py, double pz, double focus,
                                                double w, double wavelength, double intensity, double id) {
  if (__ballot(sunk) == 0ull) return;  // (uniform)
  const int n_groups = plan->n_groups, lane = threadIdx.x & 63;
  const double rays_per_source = plan->rays_per_source;
  int group = -1;
  if (sunk) {
    group = 0;
    if (rays_per_source > 0) {
      const double q = floor(id / rays_per_source);  // _pyrayt.py:352
      group = (q >= 0 && q < (double)n_groups) ? (int)q : -1;
    }
  }
  double v[SINK_STATS];
#pragma unroll
  for (int k = 0; k < SINK_STATS; ++k) v[k] = 0.0;
  if (group >= 0) {
    double pivot_y = 0.0, pivot_z = 0.0, pivot_focus = 0.0;
    const double* __restrict__ pivots = plan->pivots;
    if (pivots) { pivot_y = pivots[3 * group]; pivot_z = pivots[3 * group + 1]; pivot_focus = pivots[3 * group + 2]; }
    const double y = py - pivot_y, z = pz - pivot_z;
    const double f = focus - pivot_focus;
    const bool f_ok = f == f && fabs(f) < PRT_INF;  // a ray parallel to the axis has no intercept
    v[0] = 1.0; v[1] = y; v[2] = z; v[3] = y * y + z * z;
    v[4] = f_ok ? f : 0.0; v[5] = f_ok ? f * f : 0.0;
    v[6] = wavelength; v[7] = intensity;
    v[8] = f_ok ? 1.0 : 0.0;
    if (plan->ms_quantity >= 0) {  // (uniform)
      const bool w_ok = w == w && fabs(w) < PRT_INF;
      v[9] = w_ok ? 1.0 : 0.0; v[10] = w_ok ? w : 0.0; v[11] = w_ok ? w * w : 0.0;
    }
  }
  const int64_t wave_id = (int64_t)blockIdx.x * (PRT_BLOCK / 64) + (threadIdx.x >> 6);
  double* const mine = plan->sums + ((size_t)(wave_id & (plan->slots - 1)) * plan->limit + g) * n_groups * SINK_STATS;
  const bool with_ms = plan->ms_quantity >= 0;  // (uniform)
  unsigned long long pending = __ballot(group >= 0);
  while (pending) {  // one turn per group present in the wave: almost always exactly one
    const int leader = __ffsll((long long)pending) - 1;
    const int cur = __shfl(group, leader);
    const bool take = group == cur;
    double* const out = mine + (size_t)cur * SINK_STATS;
    // the three counts are popcounts of ballots (scalar unit); the sums that can be non-zero go through the shuffles
    const unsigned long long m_take = __ballot(take);
    const double n_rows = (double)__popcll(m_take), n_focus = (double)__popcll(__ballot(take && v[8] != 0.0));
    if (lane == 0) {
      atomicAdd(out + 0, n_rows);
      if (n_focus != 0.0) atomicAdd(out + 8, n_focus);
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      double t = take ? v[k] : 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
      if (lane == 0 && t != 0.0) atomicAdd(out + k, t);
    }
    if (with_ms) {
      const double n_ms = (double)__popcll(__ballot(take && v[9] != 0.0));
      if (lane == 0 && n_ms != 0.0) atomicAdd(out + 9, n_ms);
#pragma unroll
      for (int k = 10; k < 12; ++k) {
        double t = take ? v[k] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0 && t != 0.0) atomicAdd(out + k, t);
      }
    }
    pending &= ~m_take;
  }
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
// PLAN = the trace runs under a record plan (PlanDev): which rows are stored is decided per ray, the sums of the rows
// that pass may be accumulated in the kernel, and `assume` carries two fields -- bits 0..3 how the carried rays are
// placed (0 look-back, 1 every ray goes on, 2 none does), bit 4 "no ray of this generation stores a row" (else, in a
// dense form: every ray does).  The instantiations without a plan are the product kernels of round 5, instruction for
// instruction (tools/kernel_isa.py): everything a plan adds sits behind `if (PLAN)`.
template <bool CULL, bool COMPACT, bool PLAN = false>
__global__ void __launch_bounds__(PRT_BLOCK, PRT_GEN_WAVES)
k_generation(SceneDev scene, const double* __restrict__ rays, int64_t ld, double* __restrict__ next,
             int64_t ld_next, double* __restrict__ rows, int64_t ld_rows, FusedCtrl* __restrict__ ctrl,
             GenCtrl* __restrict__ gen, int g, unsigned long long* __restrict__ tiles_cur,
             unsigned long long* __restrict__ tiles_next, double next_generation, int generation_limit,
             double ray_offset, int keep_absorbed, HostMirror* mirror, unsigned long long epoch,
             int mirror_slot, int batch_last, int assume, const PlanDev* plan_dev) {
  const ConstPlan plan = (ConstPlan)(unsigned long long)plan_dev;
  const int relaunch = (g + 1 != generation_limit) ? 1 : 0;  // the state written here is traced further
  __shared__ int s_wave_live[4], s_wave_carry[4];
  __shared__ unsigned s_excl[3];
  // A ticket from one atomic word would also give start-ordered tile numbers, but a single
  // word hands out only ~80 tickets/us chip-wide: 4k tiles would cost ~50 us per generation.
  const int tile = blockIdx.x;
  // The head of this wave's segment of the state (lean segments, interact_store_rows), asked for before anything else:
  // its address needs nothing that is loaded, and what it says decides which rows the burst below asks for.  (Asked
  // for behind the control words, with rows 9 / 10 / 12 fetched once it had answered, a ray set that does NOT qualify
  // paid a second round trip in front of its hit phase: +6.5 % per trace; in this order +3 %, profiles/r5/ab_round5.txt.)
  // (A state buffer's leading dimension is its ray count: three entries from col0 on are inside it or not asked for.)
  typedef const __attribute__((address_space(4))) double* ConstRow;  // (the state is not written by this launch: scalar loads)
  const int64_t col0 = (int64_t)tile * PRT_BLOCK + (int64_t)(__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u);
  double head_id = 0.0, head_intensity = 0.0, head_wavelength = 0.0;
  if (COMPACT && g > 0 && col0 + 2 < ld) {  // (uniform)
    const ConstRow head = (ConstRow)(unsigned long long)(rays + 12 * ld + col0);
    head_id = head[0]; head_intensity = head[1]; head_wavelength = head[2];
  }
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
  bool stored = false, sunk = false;  // (PLAN: this ray's row is stored / goes to the sink)
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
    // rows 9, 10, 12: a lean segment (written by interact_store_rows of the generation before) or the rows
    // themselves -- asked for FIRST, so that a ray set that does not qualify still fetches its ten rows in one burst
    // (behind the other seven they were a second round trip in front of the hit phase: +7 % for such a ray set)
    const bool lean_in = COMPACT && g > 0 && col0 + 2 < n && lean_tagged(head_id);  // (uniform; fewer than three rays: never lean)
    double intensity_in = head_intensity, wavelength_in = head_wavelength;
    double id_in = (double)(__double_as_longlong(head_id) & 0xffffffffffffll) + (double)(threadIdx.x & 63u);
    if (!lean_in) {
      intensity_in = row_load(tile_rays + 9 * ld, lane_bytes);
      wavelength_in = row_load(tile_rays + 10 * ld, lane_bytes);
      id_in = row_load(tile_rays + 12 * ld, lane_bytes);
    }
    r.ox = row_load(tile_rays + 0 * ld, lane_bytes); r.oy = row_load(tile_rays + 1 * ld, lane_bytes);
    r.oz = row_load(tile_rays + 2 * ld, lane_bytes);
    if (!COMPACT) r.ow = row_load(tile_rays + 3 * ld, lane_bytes);
    r.dx = row_load(tile_rays + 4 * ld, lane_bytes); r.dy = row_load(tile_rays + 5 * ld, lane_bytes);
    r.dz = row_load(tile_rays + 6 * ld, lane_bytes);
    if (!COMPACT) {
      r.dw = row_load(tile_rays + 7 * ld, lane_bytes);
      PARK(0) = row_load(tile_rays + 8 * ld, lane_bytes);
    }
    PARK(3) = row_load(tile_rays + 11 * ld, lane_bytes);
    PARK(1) = intensity_in;
    PARK(2) = wavelength_in;
    PARK(4) = id_in;
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
    if (PLAN) {
      bool wanted = live;
      const int n_rec = plan->n_rec;
      if (n_rec > 0) {  // (uniform)
        bool listed = false;
        for (int k = 0; k < n_rec; ++k) listed = listed || prim == plan->rec_prims[k];
        wanted = live && listed;
      }
      stored = wanted && plan->store_rows != 0;
      sunk = wanted && plan->n_groups > 0;
    }
  }
  // workgroup aggregate and ranks: wave ballots + popcounts, four waves combined through LDS
  // (PLAN: "live" in everything that places or counts rows below means "stores a row")
  STAMP(2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m_live = __ballot(PLAN ? stored : live), m_carry = __ballot(carry);
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
  int wave_carry_base = 0;  // rays the waves in front of this one (in this tile) carry on
  for (int w = 0; w < PRT_BLOCK / 64; ++w) {
    if (w < wave) { live_rank += s_wave_live[w]; wave_carry_base += s_wave_carry[w] & 0xffff; }
    agg_live += s_wave_live[w];
    agg_carry += s_wave_carry[w];
  }
  carry_rank += wave_carry_base;
  // (lanes of this wave that hold a ray at all: a lean segment needs every one of them carried on)
  const unsigned long long m_held = __ballot(i < n && !failed);
  const unsigned agg_kept = agg_carry >> 16;
  agg_carry &= 0xffffu;
  // Launched on the dead list of the generation before (assume 5 / 6): the dead rays in front of this tile, in it,
  // and in all; one wave reads the list (a handful of entries as a rule: that is when the host offers this mode)
  const int carry_form = PLAN ? (assume & 15) : assume;  // (a plan's launches: 0, 1 or 2)
  const bool rec_none = PLAN && (assume & 16) != 0;
  if (!PLAN && assume >= 5 && wave == 0) {
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
  const bool publish_here = assume && mirror != nullptr &&
                            (PLAN ? (carry_form == 2 || batch_last)
                                  : (assume == 2 || assume == 6 || batch_last));
  if (assume) {
    if (threadIdx.x == 0) {
      const int64_t mine = (n - (int64_t)tile * PRT_BLOCK) < PRT_BLOCK ? (n - (int64_t)tile * PRT_BLOCK) : PRT_BLOCK;
      bool holds = (int64_t)agg_live == (rec_none ? 0 : mine) && (int64_t)agg_carry == (carry_form == 1 ? mine : 0);
      if (!PLAN && assume >= 5) {  // (thread 0 wrote s_excl itself, above)
        const unsigned before = s_excl[0];
        const int64_t alive = mine - (int64_t)s_excl[1];
        holds = (int64_t)agg_live == alive && (int64_t)agg_carry == (assume == 5 ? alive : 0);
        s_excl[0] = (unsigned)tile * PRT_BLOCK - before;
        s_excl[1] = assume == 5 ? (unsigned)tile * PRT_BLOCK - before : 0u;
      }
      // a tile of a launch that keeps its absorbed rays notes how many it kept (generation 0 has nobody to empty
      // its list for it: it keeps rays without noting them, and the generation behind it compacts by look-back)
      if (!PLAN && agg_kept && g > 0) {
        DeadList* dead = dead_list(gen, g);
        const unsigned at = atomicAdd(&dead->count, 1u);
        if (at < (unsigned)kDeadListCap) dead->entry[at] = ((unsigned)tile << 9) | agg_kept;
      }
      if (!holds && !failed)  // in place before this tile checks in below
        raise_verdict(&ctrl->error, PRT_ERR_SPECULATION);
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
  if (!PLAN && assume >= 5) {
    __syncthreads();
    excl_live = s_excl[0];
    excl_carry = s_excl[1];
  } else if (assume) {
    excl_live = rec_none ? 0 : (int64_t)tile * PRT_BLOCK;
    excl_carry = carry_form == 1 ? (int64_t)tile * PRT_BLOCK : 0;
  } else {
  if (wave == 0) {
    unsigned e_live, e_carry;
    bool ok = lookback(tiles_cur, tile, agg_live, agg_carry, e_live, e_carry, &ctrl->error);
    // test hook: pretend the spin expired (such traces publish through k_fused_reinit behind the batch)
    if (ctrl->pad == 1 && tile == 3 && lane == 0) raise_verdict(&ctrl->error, PRT_ERR_STALL);
    (void)ok;
    if (lane == 0) { s_excl[0] = e_live; s_excl[1] = e_carry; }
  }
  __syncthreads();
  excl_live = s_excl[0];
  excl_carry = s_excl[1];
  finisher = tile == last_tile;
  }
  STAMP(6);

  if (finisher && threadIdx.x == 0) {  // totals are known here: hand over to g + 1
    // (dense mode: the totals are the assumption itself; if it failed the error word says so)
    // (assume 5 / 6: every ray but the dead ones of the list)
    const int64_t total_live = PLAN ? (assume ? (rec_none ? 0 : n) : excl_live + agg_live)
                               : assume >= 5 ? n - (int64_t)s_excl[2] : assume ? n : excl_live + agg_live;
    const int64_t total_carry = PLAN ? (assume ? (carry_form == 1 ? n : 0) : excl_carry + agg_carry)
                                : assume ? (assume == 1 ? n : assume == 5 ? n - (int64_t)s_excl[2] : 0) : excl_carry + agg_carry;
    // (the rays that go on are among those that were alive: no row recorded means no ray carried -- unless a plan
    // stores no row for rays that live on)
    const int64_t next_in = PLAN ? total_carry : ((total_live == 0) ? 0 : total_carry);
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
  if (!PLAN && !live) return;
  if (live) {  // (PLAN: the lanes without a live ray stay for the wave reductions of the sink below)
  // un-park the early columns
  sh.generation = PARK(0); sh.intensity = PARK(1); sh.wavelength = PARK(2);
  sh.index_in = PARK(3); sh.id = PARK(4);
  r.ox = PARK(5); r.oy = PARK(6); r.oz = PARK(7);
  // a lean segment (see interact_store_rows): this wave's outputs are the 64 columns one wave of the next generation
  // reads, in lane order, and their three constant rows are redundant
  bool lean_out = false;
  double id0 = 0.0;
  if (COMPACT && relaunch && m_carry == m_held && __popcll(m_held) >= 3 && excl_carry == (int64_t)tile * PRT_BLOCK &&
      wave_carry_base == 64 * wave) {  // (uniform)
    // (every lane still here is live and carried; the wave's lanes are 0 .. k - 1)
    const double intensity0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sh.intensity)),
                                               __builtin_amdgcn_readfirstlane(__double2loint(sh.intensity)));
    const double wavelength0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sh.wavelength)),
                                                __builtin_amdgcn_readfirstlane(__double2loint(sh.wavelength)));
    id0 = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(sh.id)),
                           __builtin_amdgcn_readfirstlane(__double2loint(sh.id)));
    const bool redundant = __double_as_longlong(sh.intensity) == __double_as_longlong(intensity0) &&
                           __double_as_longlong(sh.wavelength) == __double_as_longlong(wavelength0) &&
                           sh.id == id0 + (double)lane && id0 >= 0.0 && id0 < 0x1p48 && id0 == (double)(long long)id0;
    lean_out = __ballot(!redundant) == 0ull;
  }
  if (!interact_store_rows<COMPACT>(sh, r, carry, (unsigned)live_rank * 8u, (unsigned)carry_rank * 8u,
                                    next + uniform64(excl_carry), ld_next, rows + uniform64(row_base + excl_live),
                                    ld_rows, next_generation, relaunch, ray_offset, lean_out, id0, PLAN ? stored : true,
                                    PLAN ? (unsigned)plan->columns : 0x7fffu))
    atomicExch(&ctrl->error, PRT_ERR_FULL_ROWS);
  if (sh.err) raise_error(&ctrl->error, sh.err);
  if (PLAN) {
    // the sink's operands wait in the lane's parking slots (free again: the record columns are stored) for the
    // wave to come together below -- held in registers across the join they cost the kernel 24-32 B of scratch
    if (plan->n_groups > 0 && sunk) {  // (first clause uniform)
      PARK(0) = sh.py; PARK(3) = sh.pz;
      PARK(5) = r.ox - sh.tx * r.oy / sh.ty;  // the axis intercept, as k_frame_reduce computes it from the stored columns
      const int quantity = plan->ms_quantity;
      if (quantity >= 0) {
        double q = sink_quantity(quantity, sh, r.ox, r.oy, r.oz);
        if (plan->ms_transform == 1) q = sin(q);
        PARK(6) = q - plan->ms_about;
      }
    }
  }
  }
  if (PLAN) {
    if (plan->n_groups > 0)  // (uniform; every lane of the wave: the reductions need their partners)
      sink_accumulate(plan, g, sunk, PARK(0), PARK(3), PARK(5), PARK(6), PARK(2), PARK(1), PARK(4));
  }
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
