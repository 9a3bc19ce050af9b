// prt_frame.hpp -- reductions over the result frame, on the device.
//
// The record block of a trace stays in HBM as (15, R) rows (pyrayt/_pyrayt.py:147-186 is the frame
// it becomes).  What the reference's users do with that frame (examples/lens_design.ipynb cells
// 11-16, 19-20, 38) is always the same shape of work: select the rows of one surface and/or one
// generation, group them by source (ray id // rays_per_source, _pyrayt.py:349-354 -- which is also
// "by wavelength" when every source has its own), and look at per-group spot positions and at the
// x-axis intercept of each ray, `x0 - x_tilt * y0 / y_tilt` (cells 12 and 15).  k_frame_reduce does
// all of it in one pass over the eleven columns involved: per group
//     [0] count  [1] sum (y1 - py)  [2] sum (z1 - pz)  [3] sum ((y1 - py)^2 + (z1 - pz)^2)
//     [4] sum (focus - pf)  [5] sum (focus - pf)^2  [6] sum wavelength  [7] sum intensity
//     [8] number of rows with a finite axis intercept (a ray parallel to the axis -- y_tilt = 0, e.g. the
//         axial ray of a cone -- has none: 0 / 0; sums 4 and 5 run over the others, like pandas' mean / std
//         skip NaN, so they are divided by this count and not by [0])
// accumulated in registers per wave, then added to the output with one atomic per touched entry.
// The pivots (py, pz, pf) make the second moments well conditioned: the host wrapper runs the pass
// twice, the second time about the first pass's means.
#pragma once

enum { FRAME_STATS = 9, FRAME_OUT = 8 };  // sums per group in a reduction pass / statistics per group of prt_frame_stats
static const int kFrameRowsPerWave = 16 * 64;  // a wave's share of the block: 16 consecutive slices of 64 rows
                                               // (3M rows = 2 930 waves: three per SIMD)

// Rows are generation-major and, inside a generation, in ascending id order (pyrayt/_pyrayt.py:168-186),
// so the group index id // rays_per_source changes only a handful of times along the block.  Every wave
// therefore walks a contiguous run of rows with its eight sums in registers for the group it is in
// ("current"), and only when a row of another group turns up does it fold the 64 lanes together (xor
// shuffles) and add the eight totals to the output -- one atomic per statistic, group and wave-run
// instead of one per row.  A slice that straddles groups is worked off group by group.
__device__ __forceinline__ void frame_flush(double (&acc)[FRAME_STATS], int group, double* __restrict__ out) {
#pragma unroll
  for (int k = 0; k < FRAME_STATS; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(out + (size_t)group * FRAME_STATS + k, v);
    acc[k] = 0.0;
  }
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_reduce(const double* __restrict__ rows, int64_t ld, int64_t n_rows, double surface, double generation,
               double rays_per_source, int n_groups, const double* __restrict__ pivots,
               double* __restrict__ out, int slots) {
  const bool any_surface = surface != surface, any_generation = generation != generation;  // NaN = no filter
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (PRT_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t first = wave * kFrameRowsPerWave;
  const int64_t last = first + kFrameRowsPerWave < n_rows ? first + kFrameRowsPerWave : n_rows;
  double acc[FRAME_STATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int current = -1;  // wave-uniform: the group the sums in `acc` belong to
  // waves spread their totals over `slots` copies of the output (k_frame_fold adds them up): with a
  // handful of groups every wave would otherwise queue up on the same eight words
  double* const mine = out + (size_t)(wave & (slots - 1)) * n_groups * FRAME_STATS;
  for (int64_t base = first; base < last; base += 64) {
    const int64_t j = base + lane;
    int group = -1;  // -1: this lane has nothing to add
    if (j < last && (any_surface || rows[PRT_COL_SURFACE * ld + j] == surface) &&
        (any_generation || rows[PRT_COL_GENERATION * ld + j] == generation)) {
      group = 0;
      if (rays_per_source > 0) {
        const double g = floor(rows[PRT_COL_ID * ld + j] / rays_per_source);  // _pyrayt.py:352
        group = (g >= 0 && g < (double)n_groups) ? (int)g : -1;
      }
    }
    double v[FRAME_STATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (group >= 0) {
      double pivot_y = 0.0, pivot_z = 0.0, pivot_focus = 0.0;
      if (pivots) { pivot_y = pivots[3 * group]; pivot_z = pivots[3 * group + 1]; pivot_focus = pivots[3 * group + 2]; }
      const double y = rows[PRT_COL_Y1 * ld + j] - pivot_y, z = rows[PRT_COL_Z1 * ld + j] - pivot_z;
      const double focus = rows[PRT_COL_X0 * ld + j] -
                           rows[PRT_COL_XTILT * ld + j] * rows[PRT_COL_Y0 * ld + j] / rows[PRT_COL_YTILT * ld + j];
      const double f = focus - pivot_focus;
      const bool f_ok = f == f && fabs(f) < PRT_INF;  // a ray parallel to the axis has no intercept
      v[0] = 1.0; v[1] = y; v[2] = z; v[3] = y * y + z * z;
      v[4] = f_ok ? f : 0.0; v[5] = f_ok ? f * f : 0.0;
      v[6] = rows[PRT_COL_WAVELENGTH * ld + j]; v[7] = rows[PRT_COL_INTENSITY * ld + j];
      v[8] = f_ok ? 1.0 : 0.0;
    }
    unsigned long long pending = __ballot(group >= 0);
    while (pending) {  // one turn per group present in the slice: almost always exactly one
      const int leader = __ffsll((long long)pending) - 1;
      const int g = __shfl(group, leader);
      if (g != current) {
        if (current >= 0) frame_flush(acc, current, mine);
        current = g;
      }
      const bool take = group == g;
      if (take) {
#pragma unroll
        for (int k = 0; k < FRAME_STATS; ++k) acc[k] += v[k];
      }
      pending &= ~__ballot(take);
    }
  }
  if (current >= 0) frame_flush(acc, current, mine);
}

// out[e] = sum over the slots of partial[slot][e]
__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_fold(const double* __restrict__ partial, int slots, int n, double* __restrict__ out) {
  const int e = blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (e >= n) return;
  double v = 0.0;
  for (int k = 0; k < slots; ++k) v += partial[(size_t)k * n + e];
  out[e] = v;
}

// out: (n_groups, 9) float64 on the device, overwritten.  surface / generation: NaN = every row.
// rays_per_source <= 0: one group.  pivots: DEVICE (n_groups, 3) float64 -- per group the (y, z, focus)
// subtracted before accumulating -- or null for zeros.
static const int kFrameSlots = 64;           // copies of the output the waves spread their atomics over ...
static const int kFrameSlotGroups = 2048;    // ... when there are at most this many groups (64 x 2048 x 64 B = 8 MiB)
extern "C" int prt_frame_reduce(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                                double generation, double rays_per_source, int n_groups, const double* pivots,
                                double* out, void* stream) {
  if (n_rows < 0 || ld < n_rows || n_groups < 1 || !out || (n_rows && !rows)) return fail(PRT_ERR_ARG, "bad buffers");
  if (!(rays_per_source > 0) && n_groups != 1) return fail(PRT_ERR_ARG, "one group without rays_per_source");
  int rc = ops_device(device);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t out_bytes = (size_t)n_groups * FRAME_STATS * sizeof(double);
  if (n_rows == 0) {
    HIP_TRY(hipMemsetAsync(out, 0, out_bytes, st));
    return PRT_OK;
  }
  const int64_t waves = (n_rows + kFrameRowsPerWave - 1) / kFrameRowsPerWave;
  const unsigned grid = (unsigned)((waves + PRT_BLOCK / 64 - 1) / (PRT_BLOCK / 64));
  // few groups and many waves: partial sums in a stream-ordered scratch block, folded by a second kernel
  // (measured, 3M rows: one group 305 us with direct atomics, 55 us through the slots; 100 groups 31 / 54 us)
  const int slots = (n_groups <= kFrameSlotGroups && waves / n_groups >= 256) ? kFrameSlots : 1;
  double* partial = out;
  if (slots > 1) HIP_TRY(hipMallocAsync((void**)&partial, out_bytes * slots, st));
  HIP_TRY(hipMemsetAsync(partial, 0, out_bytes * slots, st));
  hipLaunchKernelGGL(k_frame_reduce, dim3(grid), dim3(PRT_BLOCK), 0, st, rows, ld, n_rows, surface, generation,
                     rays_per_source, n_groups, pivots, partial, slots);
  if (slots > 1) {
    const int n = n_groups * FRAME_STATS;
    hipLaunchKernelGGL(k_frame_fold, dim3((n + PRT_BLOCK - 1) / PRT_BLOCK), dim3(PRT_BLOCK), 0, st, partial, slots, n, out);
    HIP_TRY(hipFreeAsync(partial, st));
  }
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ---- the statistics themselves, in one call ---------------------------------------------------------
// Two reduction passes without a trip to the host in between: the first pass's sums become per-group
// pivots (means) on the device, the second pass accumulates about them, and a last small kernel turns
// the sums into what the notebook looks at.  Per group: [0] count  [1] mean y1  [2] mean z1
// [3] rms spot radius about that centroid  [4] mean axis intercept  [5] its standard deviation (both over
// the rows that have one, as pandas would; NaN if none has)  [6] mean wavelength  [7] mean intensity;
// NaN in [1..7] for a group without rows.
__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_pivots(const double* __restrict__ sums, int n_groups, double* __restrict__ pivots) {
  const int g = blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (g >= n_groups) return;
  const double count = sums[g * FRAME_STATS], with_focus = sums[g * FRAME_STATS + 8];
  const double safe = count > 0 ? count : 1.0;
  pivots[3 * g + 0] = sums[g * FRAME_STATS + 1] / safe;
  pivots[3 * g + 1] = sums[g * FRAME_STATS + 2] / safe;
  pivots[3 * g + 2] = sums[g * FRAME_STATS + 4] / (with_focus > 0 ? with_focus : 1.0);
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_finish(const double* __restrict__ sums, const double* __restrict__ pivots, int n_groups,
               double* __restrict__ out) {
  const int g = blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (g >= n_groups) return;
  const double* s = sums + (size_t)g * FRAME_STATS;
  const double count = s[0], with_focus = s[8];  // (rows without an axis intercept are not in sums 4 and 5)
  const double safe = count > 0 ? count : 1.0, safe_f = with_focus > 0 ? with_focus : 1.0;
  const double dy = s[1] / safe, dz = s[2] / safe, df = s[4] / safe_f;
  const double var_r = fmax(s[3] / safe - dy * dy - dz * dz, 0.0), var_f = fmax(s[5] / safe_f - df * df, 0.0);
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  double* o = out + (size_t)g * FRAME_OUT;
  o[0] = count;
  o[1] = count > 0 ? pivots[3 * g + 0] + dy : nan;
  o[2] = count > 0 ? pivots[3 * g + 1] + dz : nan;
  o[3] = count > 0 ? sqrt(var_r) : nan;
  o[4] = with_focus > 0 ? pivots[3 * g + 2] + df : nan;
  o[5] = with_focus > 0 ? sqrt(var_f) : nan;
  o[6] = count > 0 ? s[6] / safe : nan;
  o[7] = count > 0 ? s[7] / safe : nan;
}

// out: (n_groups, 8) float64 on the device; workspace: prt_frame_stats_workspace_bytes(n_groups) bytes
extern "C" int64_t prt_frame_stats_workspace_bytes(int n_groups) {
  return n_groups < 1 ? 0 : (int64_t)n_groups * (FRAME_STATS + 3) * (int64_t)sizeof(double);
}

// The statistics of a frame whose rows are spread over the ranks of a communicator (a trace with gather "none":
// every rank holds the rows of its own id range).  The sums of a pass are additive over any partition of the
// rows, so every rank reduces its own rows and the (n_groups, 9) sums are added across ranks -- one small
// all-reduce per pass, the second about the pivots of the WHOLE frame's first pass, which every rank then holds.
// Nothing of the frame itself moves (the alternative, re-assembling it, brings 315 MB into every GPU for the
// north-star job).  comm == nullptr: the single-rank statistics.
static int frame_stats_impl(prt_comm* comm, int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                            double generation, double rays_per_source, int n_groups, double* out, void* workspace,
                            hipStream_t st) {
  if (!workspace || !out || n_groups < 1) return fail(PRT_ERR_ARG, "bad buffers");
  double* sums = (double*)workspace;
  double* pivots = sums + (size_t)n_groups * FRAME_STATS;
  const size_t n_sums = (size_t)n_groups * FRAME_STATS;
  int rc = prt_frame_reduce(device, rows, ld, n_rows, surface, generation, rays_per_source, n_groups, nullptr, sums, st);
  if (rc) return rc;
  if (comm) RCCL_TRY(g_rccl.AllReduce(sums, sums, n_sums, ncclDouble, ncclSum, comm->comm, st));
  const dim3 grid((n_groups + PRT_BLOCK - 1) / PRT_BLOCK);
  hipLaunchKernelGGL(k_frame_pivots, grid, dim3(PRT_BLOCK), 0, st, sums, n_groups, pivots);
  rc = prt_frame_reduce(device, rows, ld, n_rows, surface, generation, rays_per_source, n_groups, pivots, sums, st);
  if (rc) return rc;
  if (comm) RCCL_TRY(g_rccl.AllReduce(sums, sums, n_sums, ncclDouble, ncclSum, comm->comm, st));
  hipLaunchKernelGGL(k_frame_finish, grid, dim3(PRT_BLOCK), 0, st, sums, pivots, n_groups, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_frame_stats(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                               double generation, double rays_per_source, int n_groups, double* out,
                               void* workspace, void* stream) {
  return frame_stats_impl(nullptr, device, rows, ld, n_rows, surface, generation, rays_per_source, n_groups, out,
                          workspace, (hipStream_t)stream);
}

extern "C" int prt_frame_stats_sharded(prt_comm* comm, const double* rows, int64_t ld, int64_t n_rows, double surface,
                                       double generation, double rays_per_source, int n_groups, double* out,
                                       void* workspace, void* stream) {
  if (!comm) return fail(PRT_ERR_ARG, "communicator is null");
  return frame_stats_impl(comm, comm->device, rows, ld, n_rows, surface, generation, rays_per_source, n_groups, out,
                          workspace, (hipStream_t)stream);
}

// The two small steps between and behind the passes on their own, for sums that are added across ranks by another
// transport (torch.distributed over gloo in the tests; MPI): prt_frame_reduce -> add -> prt_frame_pivots ->
// prt_frame_reduce about them -> add -> prt_frame_finish.  sums (n_groups, 9), pivots (n_groups, 3), out
// (n_groups, 8): device.
extern "C" int prt_frame_pivots(int device, const double* sums, int n_groups, double* pivots_out, void* stream) {
  if (!sums || !pivots_out || n_groups < 1) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_frame_pivots, dim3((n_groups + PRT_BLOCK - 1) / PRT_BLOCK), dim3(PRT_BLOCK), 0,
                     (hipStream_t)stream, sums, n_groups, pivots_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_frame_finish(int device, const double* sums, const double* pivots, int n_groups, double* out,
                                void* stream) {
  if (!sums || !pivots || !out || n_groups < 1) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_frame_finish, dim3((n_groups + PRT_BLOCK - 1) / PRT_BLOCK), dim3(PRT_BLOCK), 0,
                     (hipStream_t)stream, sums, pivots, n_groups, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ---- mean squares of one quantity of the frame (examples/lens_design.ipynb cells 20, 28, 32) ------------------
// The notebook's merit functions are all `np.mean(np.square(f(rows) - c))` over the rows of the last generation
// (cell 20: f = sin(y_tilt), c = sin(angle) -- the coma metric; cells 28 / 32: f = the axis intercept, c = the design
// focus).  One pass: per group [0] rows counted  [1] sum of v  [2] sum of v^2, v = f(quantity) - about, over the rows
// that pass the surface / generation filter and whose v is finite (a ray parallel to the axis has no intercept:
// pandas' mean skips the NaN).  quantity: a frame column (PRT_COL_*, 0..14) or PRT_FRAME_AXIS_INTERCEPT (15) =
// x0 - x_tilt y0 / y_tilt; transform: 0 none, 1 sin (numpy's, libm-accurate: the device's sin()).
enum { FRAME_AXIS_INTERCEPT = 15, FRAME_MS_STATS = 3 };
__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_mean_square(const double* __restrict__ rows, int64_t ld, int64_t n_rows, double surface, double generation,
                    double rays_per_source, int n_groups, int quantity, int transform, double about,
                    double* __restrict__ out) {
  const bool any_surface = surface != surface, any_generation = generation != generation;  // NaN = no filter
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (PRT_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t first = wave * kFrameRowsPerWave;
  const int64_t last = first + kFrameRowsPerWave < n_rows ? first + kFrameRowsPerWave : n_rows;
  double acc[FRAME_MS_STATS] = {0, 0, 0};
  int current = -1;
  auto flush = [&](int group) {
#pragma unroll
    for (int k = 0; k < FRAME_MS_STATS; ++k) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && v != 0.0) atomicAdd(out + (size_t)group * FRAME_MS_STATS + k, v);
      acc[k] = 0.0;
    }
  };
  for (int64_t base = first; base < last; base += 64) {
    const int64_t j = base + lane;
    int group = -1;
    double v = 0.0;
    if (j < last && (any_surface || rows[PRT_COL_SURFACE * ld + j] == surface) &&
        (any_generation || rows[PRT_COL_GENERATION * ld + j] == generation)) {
      group = 0;
      if (rays_per_source > 0) {
        const double g = floor(rows[PRT_COL_ID * ld + j] / rays_per_source);  // _pyrayt.py:352
        group = (g >= 0 && g < (double)n_groups) ? (int)g : -1;
      }
      double q = quantity == FRAME_AXIS_INTERCEPT
                     ? rows[PRT_COL_X0 * ld + j] - rows[PRT_COL_XTILT * ld + j] * rows[PRT_COL_Y0 * ld + j] / rows[PRT_COL_YTILT * ld + j]
                     : rows[(int64_t)quantity * ld + j];
      if (transform == 1) q = sin(q);
      v = q - about;
      if (!(v == v && fabs(v) < PRT_INF)) group = -1;
    }
    unsigned long long pending = __ballot(group >= 0);
    while (pending) {  // one turn per group present in the slice: almost always exactly one
      const int leader = __ffsll((long long)pending) - 1;
      const int g = __shfl(group, leader);
      if (g != current) {
        if (current >= 0) flush(current);
        current = g;
      }
      const bool take = group == g;
      if (take) { acc[0] += 1.0; acc[1] += v; acc[2] += v * v; }
      pending &= ~__ballot(take);
    }
  }
  if (current >= 0) flush(current);
}

// out: (n_groups, 3) float64 on the device, overwritten: rows counted, sum v, sum v^2 (additive over any partition
// of the rows: a sharded frame adds them across ranks and divides afterwards).
extern "C" int prt_frame_mean_square(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                                     double generation, double rays_per_source, int n_groups, int quantity,
                                     int transform, double about, double* out, void* stream) {
  if (n_rows < 0 || ld < n_rows || n_groups < 1 || !out || (n_rows && !rows)) return fail(PRT_ERR_ARG, "bad buffers");
  if (!(rays_per_source > 0) && n_groups != 1) return fail(PRT_ERR_ARG, "one group without rays_per_source");
  if (quantity < 0 || quantity > FRAME_AXIS_INTERCEPT || transform < 0 || transform > 1)
    return fail(PRT_ERR_ARG, "quantity: a frame column 0..14 or 15 (axis intercept); transform: 0 none, 1 sin");
  int rc = ops_device(device);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_groups * FRAME_MS_STATS * sizeof(double), st));
  if (n_rows == 0) return PRT_OK;
  const int64_t waves = (n_rows + kFrameRowsPerWave - 1) / kFrameRowsPerWave;
  const unsigned grid = (unsigned)((waves + PRT_BLOCK / 64 - 1) / (PRT_BLOCK / 64));
  hipLaunchKernelGGL(k_frame_mean_square, dim3(grid), dim3(PRT_BLOCK), 0, st, rows, ld, n_rows, surface, generation,
                     rays_per_source, n_groups, quantity, transform, about, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}
