// micro-benchmark + exhaustive-ish check of csrc/prt_math.hpp: divisions that share a denominator.
//   1. bit-equality of prt_div2 / prt_div3 with the compiler's `/` on random operands of every
//      class (moderate, huge, tiny, denormal, zero, inf, NaN; per-lane mixes so that both the
//      shared path and the fallback run)
//   2. sustained cost per wave of: 3 x `/` by one denominator, prt_div3, 2 x `/`, prt_div2
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off div_shared.hip -o div_shared && ./div_shared
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../pyrayt_amd/csrc/prt_math.hpp"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// a double of class `cls` from random bits
__device__ __forceinline__ double make(uint64_t bits, int cls) {
  const uint64_t sign = bits & 0x8000000000000000ull, frac = bits & 0x000fffffffffffffull;
  uint64_t e;
  switch (cls) {
    case 0: e = 1023 - 40 + (bits >> 52) % 80; break;      // moderate
    case 1: e = 643 - 3 + (bits >> 52) % 8; break;         // around the lower window edge
    case 2: e = 1403 - 3 + (bits >> 52) % 8; break;        // around the upper window edge
    case 3: e = 1 + (bits >> 52) % 2046; break;            // any normal
    case 4: e = 0; break;                                  // denormal
    case 5: return __longlong_as_double((long long)sign);  // +-0
    case 6: return __longlong_as_double((long long)(sign | 0x7ff0000000000000ull));  // +-inf
    default: return __longlong_as_double((long long)(sign | 0x7ff8000000000000ull | frac));  // NaN
  }
  return __longlong_as_double((long long)(sign | (e << 52) | frac));
}
__device__ __forceinline__ bool same(double a, double b) {
  const bool nan_a = a != a, nan_b = b != b;
  if (nan_a || nan_b) return nan_a && nan_b;
  return __double_as_longlong(a) == __double_as_longlong(b);
}

// mode 0: every lane of a wave draws moderate operands (shared path); 1: per-lane random classes
__global__ void __launch_bounds__(256) k_check(uint64_t seed, int mode, int rounds, unsigned long long* bad,
                                               unsigned long long* shared_taken) {
  uint64_t s = mix(seed ^ ((uint64_t)blockIdx.x * 256 + threadIdx.x));
  unsigned long long local_bad = 0, local_shared = 0;
  for (int it = 0; it < rounds; ++it) {
    s = mix(s);
    const uint64_t wave_bits = __shfl(s, 0);
    int cd, c0, c1, c2;
    if (mode == 0) { cd = c0 = c1 = c2 = 0; if ((wave_bits & 7) == 0) c1 = 5; }
    else if (mode == 1) { cd = (wave_bits >> 3) % 4; c0 = (wave_bits >> 5) % 4; c1 = (wave_bits >> 7) % 6; c2 = (wave_bits >> 10) % 8; }
    else { cd = (s >> 3) % 8; c0 = (s >> 6) % 8; c1 = (s >> 9) % 8; c2 = (s >> 12) % 8; }
    const double d = make(mix(s + 1), cd), n0 = make(mix(s + 2), c0), n1 = make(mix(s + 3), c1),
                 n2 = make(mix(s + 4), c2);
    double a0, a1, a2, b0, b1;
    prt_div3(n0, n1, n2, d, a0, a1, a2);
    prt_div2(n2, n0, d, b0, b1);
    const bool ok = prt_exp_window(d) && prt_num_ok(n0) && prt_num_ok(n1) && prt_num_ok(n2);
    if (__ballot(!ok) == 0ull) ++local_shared;
    const double r0 = n0 / d, r1 = n1 / d, r2 = n2 / d;
    if (!same(a0, r0) || !same(a1, r1) || !same(a2, r2) || !same(b0, r2) || !same(b1, r0)) ++local_bad;
  }
  if (local_bad) atomicAdd(bad, local_bad);
  if ((threadIdx.x & 63) == 0 && local_shared) atomicAdd(shared_taken, local_shared);
}

template <int OP>
__global__ void __launch_bounds__(256) k_time(double* out, int iters, double seed) {
  double x = 1.0 + threadIdx.x * 1e-7, y = 0.75 + threadIdx.x * 1e-7, z = 0.5 + threadIdx.x * 1e-7;
  double d = seed + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
    double q0, q1, q2 = z;
    if (OP == 0) { q0 = x / d; q1 = y / d; q2 = z / d; }
    if (OP == 1) prt_div3(x, y, z, d, q0, q1, q2);
    if (OP == 2) { q0 = x / d; q1 = y / d; }
    if (OP == 3) prt_div2(x, y, d, q0, q1);
    x = q0 + 1.0; y = q1 + 0.75; z = q2 + 0.5;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x + y + z;
}

template <int OP>
int run(const char* name) {
  const int blocks = 256 * 8, iters = 4000;
  double* out;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_time<OP><<<blocks, 256>>>(out, 10, 1.000001);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k_time<OP><<<blocks, 256>>>(out, iters, 1.000001);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_iters_per_simd = blocks * 4.0 / 1024.0 * iters;
  printf("%-34s %8.3f ms  %7.1f cycles per wave-iteration (2.4 GHz, 8 waves/SIMD)\n", name, ms,
         ms * 1e-3 * 2.4e9 / wave_iters_per_simd);
  hipFree(out);
  return 0;
}

int main() {
  unsigned long long *bad, *taken;
  CHECK(hipMalloc(&bad, 8)); CHECK(hipMalloc(&taken, 8));
  for (int mode = 0; mode < 3; ++mode) {
    CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(taken, 0, 8));
    const int blocks = 4096, rounds = 2048;
    k_check<<<blocks, 256>>>(0x1234567ull + mode, mode, rounds, bad, taken);
    CHECK(hipDeviceSynchronize());
    unsigned long long h_bad = 0, h_taken = 0;
    CHECK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&h_taken, taken, 8, hipMemcpyDeviceToHost));
    printf("check mode %d: %llu operand sets, %llu mismatches, shared path in %.1f %% of wave-rounds\n", mode,
           (unsigned long long)blocks * 256 * rounds, h_bad, 100.0 * h_taken / ((double)blocks * 4 * rounds));
    if (h_bad) return 2;
  }
  run<0>("3 x (n / d), one d");
  run<1>("prt_div3");
  run<2>("2 x (n / d), one d");
  run<3>("prt_div2");
  return 0;
}
