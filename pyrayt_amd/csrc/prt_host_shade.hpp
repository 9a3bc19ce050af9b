// prt_host_shade.hpp -- the two halves of a host-shaded interaction and the distinct-value scan.
//
// The reference's extension points are Python: a user's TracableMaterial.trace (pyrayt/materials.py:26-37,
// docs/source/reference/materials.rst:17-19) and a user's Glass.index_at (:88-99).  Neither can run on the
// device, so the library hands the caller exactly what upstream hands them:
//   prt_gather_hits / prt_scatter_shaded  `next_ray_set[..., surface_mask]` advanced to the hit point
//                                         (pyrayt/_pyrayt.py:401-410) out to the caller and its answer back in;
//   prt_unique_values                     the distinct wavelengths `index_at(ray_set.wavelength)` (:72-73) has to
//                                         be evaluated on, so that the kernels can look the results up.
// Included by prt_kernels.hip behind prt_interact (k_scan, block_rank, InteractLayout are defined there).
#pragma once

// flag = "nearest hit is surface `sid`"; counts per workgroup -> k_scan -> ordered gather
__global__ void __launch_bounds__(PRT_BLOCK)
k_select_count(const int64_t* __restrict__ surf, int64_t n, int64_t sid, int32_t* __restrict__ block_counts) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const bool flag = i < n && surf[i] == sid;
  __shared__ int s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int w = __popcll(__ballot(flag));
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(&s_count, w);
  __syncthreads();
  if (threadIdx.x == 0) {
    block_counts[2 * blockIdx.x] = s_count;
    block_counts[2 * blockIdx.x + 1] = s_count;
  }
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_select_gather(const double* __restrict__ rays, int64_t n, int64_t ld, const double* __restrict__ t,
                const int64_t* __restrict__ surf, int64_t sid, const int64_t* __restrict__ block_offsets,
                double* __restrict__ subset, int64_t ld_subset, int64_t* __restrict__ index) {
  __shared__ int s_wave[4];
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const bool flag = i < n && surf[i] == sid;
  const int rank = block_rank(flag, s_wave);
  if (!flag) return;
  const int64_t j = block_offsets[2 * blockIdx.x] + rank;
  const double ti = t[i];
  // o += d t on all four homogeneous components (_pyrayt.py:404-407), the other rows as they are
#pragma unroll
  for (int k = 0; k < 4; ++k) subset[k * ld_subset + j] = rays[k * ld + i] + rays[(4 + k) * ld + i] * ti;
#pragma unroll
  for (int k = 4; k < PRT_RAY_ROWS; ++k) subset[k * ld_subset + j] = rays[k * ld + i];
  index[j] = i;
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_scatter_shaded(const double* __restrict__ subset, int64_t k, int64_t ld_subset, const int64_t* __restrict__ index,
                 double* __restrict__ shaded, int64_t ld_shaded) {
  const int64_t j = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (j >= k) return;
  const int64_t i = index[j];
  if (i < 0 || i >= ld_shaded) return;
#pragma unroll
  for (int r = 0; r < PRT_RAY_ROWS; ++r) shaded[r * ld_shaded + i] = subset[r * ld_subset + j];
}

static int plain_device(int device) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  return PRT_OK;
}

extern "C" int prt_gather_hits(int device, const double* rays, int64_t n, int64_t ld, const double* t,
                               const int64_t* surf, int64_t surface_id, double* subset_out, int64_t ld_subset,
                               int64_t* index_out, int64_t* count_out, void* workspace, void* stream) {
  int rc = plain_device(device);
  if (rc) return rc;
  if (n < 0 || ld < n || !count_out || !workspace || (n && (!rays || !t || !surf || !subset_out || !index_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  *count_out = 0;
  if (n == 0) return PRT_OK;
  hipStream_t st = (hipStream_t)stream;
  const InteractLayout l = interact_layout(n);
  char* w = (char*)workspace;
  TraceCtrl* ctrl = (TraceCtrl*)(w + l.ctrl);
  int32_t* counts = (int32_t*)(w + l.counts);
  int64_t* offsets = (int64_t*)(w + l.offsets);
  hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, st, ctrl, n, n);
  hipLaunchKernelGGL(k_select_count, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, st, surf, n, surface_id, counts);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, (const int32_t*)counts, offsets, ctrl);
  TraceCtrl host;
  HIP_TRY(hipMemcpyAsync(&host, ctrl, sizeof(TraceCtrl), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int64_t count = host.n_live;
  if (count > ld_subset) return fail(PRT_ERR_ARG, "subset_out is narrower than the number of rays that hit the surface");
  if (count > 0)
    hipLaunchKernelGGL(k_select_gather, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, st, rays, n, ld, t, surf, surface_id,
                       (const int64_t*)offsets, subset_out, ld_subset, index_out);
  HIP_TRY(hipGetLastError());
  *count_out = count;
  return PRT_OK;
}

extern "C" int prt_scatter_shaded(int device, const double* subset, int64_t k, int64_t ld_subset, const int64_t* index,
                                  double* shaded, int64_t ld_shaded, void* stream) {
  int rc = plain_device(device);
  if (rc) return rc;
  if (k < 0 || ld_subset < k || ld_shaded < 0 || (k && (!subset || !index || !shaded)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (k == 0) return PRT_OK;
  hipLaunchKernelGGL(k_scatter_shaded, dim3(blocks_for(k)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, subset, k,
                     ld_subset, index, shaded, ld_shaded);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ---- distinct values of a device array -------------------------------------------------------------------------
// An open-addressing set of the values' bit patterns.  A wave first reduces its 64 values to the distinct ones
// among them (one pass per distinct value: a ray set of one source is one pass); the leader of each inserts.
// Word 0 of the header counts the distinct values found; the first `cap` are also listed in `out`.
static inline int64_t unique_slots(int64_t cap) {
  int64_t slots = 64;
  while (slots < 4 * cap) slots <<= 1;  // load factor <= 1/4 while the list is not full
  return slots;
}
extern "C" int64_t prt_unique_workspace_bytes(int64_t cap) {
  return 256 + unique_slots(cap < 1 ? 1 : cap) * (int64_t)sizeof(unsigned long long);
}
static const unsigned long long kUniqueEmpty = 0x7ff8dead00000001ull;  // a NaN payload no arithmetic produces

__global__ void __launch_bounds__(PRT_BLOCK)
k_unique(const double* __restrict__ values, int64_t n, unsigned long long* __restrict__ header,
         unsigned long long* __restrict__ table, unsigned long long mask, double* __restrict__ out, int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const bool in_range = i < n;
  unsigned long long bits = in_range ? (unsigned long long)__double_as_longlong(values[i]) : 0ull;
  if (bits == kUniqueEmpty) bits ^= 2ull;  // (another NaN: NaNs are told apart by payload anyway)
  unsigned long long todo = __ballot(in_range);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned lo = __shfl((unsigned)(bits & 0xffffffffull), leader), hi = __shfl((unsigned)(bits >> 32), leader);
    const unsigned long long cur = ((unsigned long long)hi << 32) | lo;
    todo &= ~__ballot(in_range && bits == cur);
    if (lane != leader) continue;
    // the value the set took in last: a ray set of one source costs one insert per launch, not one per wave
    if (__hip_atomic_load(&header[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cur) continue;
    unsigned long long h = cur * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
    for (unsigned long long probe = 0; probe <= mask; ++probe) {
      unsigned long long* slot = table + ((h + probe) & mask);
      const unsigned long long seen = atomicCAS(slot, kUniqueEmpty, cur);
      if (seen == cur) break;
      if (seen == kUniqueEmpty) {
        const unsigned long long at = atomicAdd(&header[0], 1ull);
        if ((int64_t)at < cap) out[at] = __longlong_as_double((long long)cur);
        __hip_atomic_store(&header[1], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      // (the list is full and so, soon, is the set: the caller falls back to the host once count > cap)
      if (__hip_atomic_load(&header[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned long long)cap) break;
    }
  }
}

__global__ void k_unique_init(unsigned long long* header, unsigned long long* table, int64_t slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots) table[i] = kUniqueEmpty;
  if (i == 0) { header[0] = 0; header[1] = kUniqueEmpty; }
}

extern "C" int prt_unique_values(int device, const double* values, int64_t n, double* out, int64_t cap,
                                 int64_t* count_out, void* workspace, void* stream) {
  int rc = plain_device(device);
  if (rc) return rc;
  if (n < 0 || cap < 1 || !count_out || !workspace || !out || (n && !values)) return fail(PRT_ERR_ARG, "bad buffers");
  *count_out = 0;
  if (n == 0) return PRT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t slots = unique_slots(cap);
  unsigned long long* header = (unsigned long long*)workspace;
  unsigned long long* table = (unsigned long long*)((char*)workspace + 256);
  hipLaunchKernelGGL(k_unique_init, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, header, table, slots);
  hipLaunchKernelGGL(k_unique, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, st, values, n, header, table,
                     (unsigned long long)(slots - 1), out, cap);
  unsigned long long found = 0;
  HIP_TRY(hipMemcpyAsync(&found, header, sizeof(found), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  *count_out = (int64_t)found;
  return PRT_OK;
}
