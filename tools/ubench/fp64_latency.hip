// micro-benchmark: dependent-issue latency of fp64 VALU ops with ONE wave per SIMD (no other
// wave to hide anything): cycles per op for a chain of dependent ops vs 4 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, int OP>
__global__ void __launch_bounds__(64) k(double* out, int iters, double seed, long long* cyc) {
  double a[CHAINS];
  const double b = seed + threadIdx.x * 1e-9;
  for (int j = 0; j < CHAINS; ++j) a[j] = 1.0 + j * 0.25 + threadIdx.x * 1e-7;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) {
      if (OP == 0) a[j] = fma(a[j], b, 0.5);
      if (OP == 1) a[j] = (a[j] < b) ? a[j] + 1.0 : a[j] - 0.25;
      if (OP == 2) a[j] = b / a[j] + 1.0;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int j = 0; j < CHAINS; ++j) s += a[j];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, int OP>
void run(const char* name, int blocks) {
  const int iters = 4000;
  double* out; long long* cyc; static long long h[8192];
  hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8);
  k<CHAINS, OP><<<blocks, 64>>>(out, iters, 1.000001, cyc);
  hipDeviceSynchronize();
  k<CHAINS, OP><<<blocks, 64>>>(out, iters, 1.000001, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int b = 0; b < blocks; ++b) m += h[b];
  printf("%-34s %6.1f cycles per op per chain-step (%d chains, %d waves/CU)\n", name, m / blocks / iters / CHAINS, CHAINS, blocks / 256);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("fma f64 dependent, 1 wave/CU", 256);
  run<4, 0>("fma f64 4 chains, 1 wave/CU", 256);
  run<1, 0>("fma f64 dependent, 16 waves/CU", 256 * 16);
  run<1, 1>("cmp+sel+add dependent, 1 wave/CU", 256);
  run<1, 2>("div+add dependent, 1 wave/CU", 256);
  run<1, 2>("div+add dependent, 16 waves/CU", 256 * 16);
  run<1, 2>("div+add dependent, 20 waves/CU", 256 * 20);
  return 0;
}
