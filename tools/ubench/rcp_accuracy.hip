// rcp_accuracy.hip -- how accurate are v_rcp_f64 / v_rsq_f64 on gfx950, and what do one Newton step and the
// quotient correction leave?  (max relative error over 2^26 operands spread over [2^-20, 2^20))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
__device__ __forceinline__ double u2d(uint64_t s) {
  // splitmix64 -> mantissa + exponent in [-20, 20)
  s += 0x9e3779b97f4a7c15ull; s = (s ^ (s >> 30)) * 0xbf58476d1ce4e5b9ull; s = (s ^ (s >> 27)) * 0x94d049bb133111ebull; s ^= s >> 31;
  const double m = 1.0 + (double)(s >> 12) * 0x1p-52;
  return ldexp(m, (int)(s & 0xfff) % 40 - 20);
}
__global__ void k(double* out) {
  double e_rcp = 0, e_rcp1 = 0, e_q = 0, e_q2 = 0, e_rsq = 0, e_sqrt = 0;
  const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 1024;
  for (int k = 0; k < 1024; ++k) {
    const double d = u2d(base + k), n = u2d((base + k) * 7919 + 13);
    const double exact = 1.0 / d;
    double r = __builtin_amdgcn_rcp(d);
    e_rcp = fmax(e_rcp, fabs(r - exact) / exact);
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    e_rcp1 = fmax(e_rcp1, fabs(r - exact) / exact);
    const double qe = n / d;
    double q = n * r;
    e_q2 = fmax(e_q2, fabs(q - qe) / qe);
    const double c = fma(-d, q, n); q = fma(c, r, q);
    e_q = fmax(e_q, fabs(q - qe) / qe);
    const double se = sqrt(d);
    const double y = __builtin_amdgcn_rsq(d);
    e_rsq = fmax(e_rsq, fabs(y - 1.0 / se) * se);
    double g = d * y, h = y * 0.5; const double rr = fma(-h, g, 0.5); g = fma(g, rr, g); h = fma(h, rr, h);
    const double dd = fma(-g, g, d); g = fma(dd, h, g);
    e_sqrt = fmax(e_sqrt, fabs(g - se) / se);
  }
  const double v[6] = {e_rcp, e_rcp1, e_q2, e_q, e_rsq, e_sqrt};
  for (int j = 0; j < 6; ++j) {
    double x = v[j];
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&out[j], (unsigned long long)__double_as_longlong(x));
  }
}
int main() {
  double* d; hipMalloc(&d, 48); hipMemset(d, 0, 48);
  k<<<256, 256>>>(d);
  double h[6]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
  const char* names[6] = {"v_rcp_f64", "rcp + 1 Newton", "n * rcp1 (no correction)", "n * rcp1 + correction", "v_rsq_f64", "sqrt: rsq + Goldschmidt + 1 correction"};
  for (int j = 0; j < 6; ++j) printf("%-42s max rel err %.3e (2^%.1f)\n", names[j], h[j], log2(h[j] > 0 ? h[j] : 1e-300));
  return 0;
}
