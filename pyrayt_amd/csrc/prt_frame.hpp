// prt_frame.hpp -- reductions over the result frame, on the device.
//
// The record block of a trace stays in HBM as (15, R) rows (pyrayt/_pyrayt.py:147-186 is the frame
// it becomes).  What the reference's users do with that frame (examples/lens_design.ipynb cells
// 11-16, 19-20, 38) is always the same shape of work: select the rows of one surface and/or one
// generation, group them by source (ray id // rays_per_source, _pyrayt.py:349-354 -- which is also
// "by wavelength" when every source has its own), and look at per-group spot positions and at the
// x-axis intercept of each ray, `x0 - x_tilt * y0 / y_tilt` (cells 12 and 15).  k_frame_reduce does
// all of it in one pass over the five columns involved: per group
//     [0] count  [1] sum (y1 - py)  [2] sum (z1 - pz)  [3] sum ((y1 - py)^2 + (z1 - pz)^2)
//     [4] sum (focus - pf)  [5] sum (focus - pf)^2  [6] sum wavelength  [7] sum intensity
// accumulated in LDS per workgroup, then added to the output with one atomic per touched entry.
// The pivots (py, pz, pf) make the second moments well conditioned: the host wrapper runs the pass
// twice, the second time about the first pass's means.
#pragma once

enum { FRAME_STATS = 8 };
static const int kFrameLdsGroups = 2048;  // groups accumulated in LDS (2048 x 8 doubles = 128 KiB would be
                                          // too much: see the launch, which sizes the LDS to the group count)

__global__ void __launch_bounds__(PRT_BLOCK)
k_frame_reduce(const double* __restrict__ rows, int64_t ld, int64_t n_rows, double surface, double generation,
               double rays_per_source, int n_groups, double pivot_y, double pivot_z, double pivot_focus,
               double* __restrict__ out, int use_lds) {
  extern __shared__ double acc[];  // [n_groups][FRAME_STATS] when use_lds
  if (use_lds) {
    for (int k = threadIdx.x; k < n_groups * FRAME_STATS; k += PRT_BLOCK) acc[k] = 0.0;
    __syncthreads();
  }
  const bool any_surface = surface != surface, any_generation = generation != generation;  // NaN = no filter
  for (int64_t j = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x; j < n_rows; j += (int64_t)gridDim.x * PRT_BLOCK) {
    if (!any_surface && rows[PRT_COL_SURFACE * ld + j] != surface) continue;
    if (!any_generation && rows[PRT_COL_GENERATION * ld + j] != generation) continue;
    int group = 0;
    if (rays_per_source > 0) {
      const double g = floor(rows[PRT_COL_ID * ld + j] / rays_per_source);  // _pyrayt.py:352
      if (!(g >= 0 && g < (double)n_groups)) continue;
      group = (int)g;
    }
    const double y = rows[PRT_COL_Y1 * ld + j] - pivot_y, z = rows[PRT_COL_Z1 * ld + j] - pivot_z;
    const double focus = rows[PRT_COL_X0 * ld + j] -
                         rows[PRT_COL_XTILT * ld + j] * rows[PRT_COL_Y0 * ld + j] / rows[PRT_COL_YTILT * ld + j];
    const double f = focus - pivot_focus;
    const bool f_ok = f == f && fabs(f) < PRT_INF;  // a ray parallel to the axis has no intercept
    double* slot = (use_lds ? acc : out) + (size_t)group * FRAME_STATS;
    const double v[FRAME_STATS] = {1.0, y, z, y * y + z * z, f_ok ? f : 0.0, f_ok ? f * f : 0.0,
                                   rows[PRT_COL_WAVELENGTH * ld + j], rows[PRT_COL_INTENSITY * ld + j]};
#pragma unroll
    for (int k = 0; k < FRAME_STATS; ++k) atomicAdd(slot + k, v[k]);
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < n_groups * FRAME_STATS; k += PRT_BLOCK)
      if (acc[k] != 0.0) atomicAdd(out + k, acc[k]);
  }
}

// out: (n_groups, 8) float64 on the device, overwritten.  surface / generation: NaN = every row.
// rays_per_source <= 0: one group.  pivots: (y, z, focus) subtracted before accumulating, or null.
extern "C" int prt_frame_reduce(int device, const double* rows, int64_t ld, int64_t n_rows, double surface,
                                double generation, double rays_per_source, int n_groups, const double* pivots,
                                double* out, void* stream) {
  if (n_rows < 0 || ld < n_rows || n_groups < 1 || !out || (n_rows && !rows)) return fail(PRT_ERR_ARG, "bad buffers");
  if (!(rays_per_source > 0) && n_groups != 1) return fail(PRT_ERR_ARG, "one group without rays_per_source");
  int rc = ops_device(device);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_groups * FRAME_STATS * sizeof(double), st));
  if (n_rows == 0) return PRT_OK;
  const int use_lds = n_groups <= 512 ? 1 : 0;  // 512 groups = 32 KiB of LDS per workgroup
  const size_t lds = use_lds ? (size_t)n_groups * FRAME_STATS * sizeof(double) : 0;
  const unsigned grid = (unsigned)std::min<int64_t>(blocks_for(n_rows), 256 * 8);
  hipLaunchKernelGGL(k_frame_reduce, dim3(grid), dim3(PRT_BLOCK), lds, st, rows, ld, n_rows, surface, generation,
                     rays_per_source, n_groups, pivots ? pivots[0] : 0.0, pivots ? pivots[1] : 0.0,
                     pivots ? pivots[2] : 0.0, out, use_lds);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}
