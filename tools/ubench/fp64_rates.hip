// micro-benchmark: sustained cost of fp64 fma / divide / sqrt / compare-select on gfx950.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_rates.hip -o fp64_rates && ./fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
  double a[4], b = seed + threadIdx.x * 1e-9;
  for (int j = 0; j < 4; ++j) a[j] = 1.0 + j * 0.25 + threadIdx.x * 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (OP == 0) a[j] = fma(a[j], b, 0.5);
      if (OP == 1) a[j] = b / a[j] + 1.0;
      if (OP == 2) a[j] = sqrt(a[j] + b);
      if (OP == 3) a[j] = (a[j] < b) ? a[j] + 1.0 : a[j] - 0.25;
      if (OP == 4) a[j] = a[j] * b + 0.5;  // separate mul + add (contract off)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}

template <int OP>
int run(const char* name, int ops_per_iter) {
  const int blocks = 256 * 8, iters = 2000;
  double* out;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, 10, 1.000001);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<OP><<<blocks, 256>>>(out, iters, 1.000001);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // waves per SIMD = blocks*4/1024 ; wave-ops per SIMD = that * iters * 4 * ops
  const double wave_ops_per_simd = blocks * 4.0 / 1024.0 * iters * 4.0 * ops_per_iter;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.3f ms  %6.1f cycles per wave-op (at 2.4 GHz, 8 waves/SIMD)\n", name, ms, cycles / wave_ops_per_simd);
  hipFree(out);
  return 0;
}

int main() {
  run<0>("fma f64", 1);
  run<4>("mul+add f64 (2 ops)", 2);
  run<1>("divide f64 (+1 add)", 1);
  run<2>("sqrt f64 (+1 add)", 1);
  run<3>("cmp+select+add f64", 1);
  return 0;
}
