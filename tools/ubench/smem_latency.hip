// micro-benchmark: latency of fetching a wave-uniform 192-byte record through the scalar cache
// (s_load_dwordx16 x3) vs through LDS (ds_read_b128 broadcast + readfirstlane), under load.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Rec { int k[12]; double d[18]; };  // 192 bytes

__global__ void __launch_bounds__(256) k_smem(const Rec* __restrict__ recs, int n_recs, int iters, double* out,
                                               long long* cycles) {
  double acc = 0;
  int pc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const Rec r = recs[pc];                       // uniform address -> scalar loads
    acc += r.d[0] * threadIdx.x + r.d[5] + r.d[17] + r.k[3];
    pc = (pc + 1 + (r.k[0] & 1)) % n_recs;        // next record depends on this one (like an interpreter)
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_lds(const Rec* __restrict__ recs, int n_recs, int iters, double* out,
                                              long long* cycles) {
  __shared__ Rec s_recs[16];
  for (int i = threadIdx.x; i < n_recs * (int)(sizeof(Rec) / 8); i += 256)
    reinterpret_cast<double*>(s_recs)[i] = reinterpret_cast<const double*>(recs)[i];
  __syncthreads();
  double acc = 0;
  int pc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const Rec& r = s_recs[pc];                    // uniform LDS address: broadcast reads
    acc += r.d[0] * threadIdx.x + r.d[5] + r.d[17] + r.k[3];
    pc = (pc + 1 + (r.k[0] & 1)) % n_recs;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int n_recs = 7, iters = 2000, blocks = 256 * 4;
  Rec h[16] = {};
  for (int i = 0; i < 16; ++i) { h[i].k[0] = 2 * i; h[i].d[0] = i; }
  Rec* d; double* out; long long* cyc;
  hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 8);
  long long hc[blocks];
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      if (which == 0) k_smem<<<blocks, 256>>>(d, n_recs, iters, out, cyc);
      else k_lds<<<blocks, 256>>>(d, n_recs, iters, out, cyc);
      hipDeviceSynchronize();
    }
    hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double mean = 0; for (int b = 0; b < blocks; ++b) mean += hc[b];
    printf("%s: %.0f cycles per dependent record fetch (4 workgroups/CU resident)\n", which == 0 ? "scalar cache (s_load)" : "LDS broadcast", mean / blocks / iters);
  }
  return 0;
}
