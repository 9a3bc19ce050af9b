// prt_math.hpp -- IEEE-exact float64 division of several numerators by one denominator (gfx950).
//
// hipcc expands every `n / d` on doubles into the same eleven-instruction sequence
//     ds = v_div_scale(d, d, n)        ns = v_div_scale(n, d, n)  (raises VCC)
//     r  = v_rcp(ds); two Newton steps on r (4 fma)
//     q  = ns * r;  e = fma(-ds, q, ns);  q = v_div_fmas(e, r, q);  v_div_fixup(q, d, n)
// and repeats all of it for every numerator of a normalisation x/len, y/len, z/len or of a root
// pair (-b+s)/den, (-b-s)/den, because the scaled denominator formally depends on the numerator.
// v_div_scale only moves an operand when an exponent is extreme (|d| or |n| far outside
// 2^+-380, a quotient that over- or underflows); otherwise ds == d, ns == n, VCC == 0 and
// v_div_fmas is a plain fma.  For such operands the refined reciprocal is a function of d
// alone, so it is computed once and every further numerator costs mul + fma + fma + fixup:
// the same instructions on the same values, hence bit-identical quotients by construction
// (checked against `/` on 2^31 random operand pairs per launch by tools/ubench/div_shared.hip).
// A wave in which any lane holds an operand outside the window takes the ordinary `/`.
// Zero, infinite and NaN numerators need no window: v_div_fixup decides those cases from the
// operands' classes alone.
#pragma once
#include <hip/hip_runtime.h>

// biased exponent in [643, 1403]  <=>  2^-380 <= |x| < 2^381
__device__ __forceinline__ bool prt_exp_window(double x) {
  const unsigned e = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
  return (e - 643u) <= 760u;
}
// numerators: inside the window, or a value v_div_fixup overrides anyway (0, inf, NaN)
__device__ __forceinline__ bool prt_num_ok(double n) {
  const unsigned e = ((unsigned)__double2hiint(n) >> 20) & 0x7ffu;
  return (e - 643u) <= 760u || n == 0.0 || e == 0x7ffu;
}

// the reciprocal of d after the compiler's two Newton steps (valid when prt_exp_window(d))
__device__ __forceinline__ double prt_refined_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
// n / d given r = prt_refined_rcp(d)
__device__ __forceinline__ double prt_div_by(double n, double d, double r) {
  const double q = n * r;
  const double e = __builtin_fma(-d, q, n);
  const double q1 = __builtin_fma(e, r, q);
  return __builtin_amdgcn_div_fixup(q1, d, n);
}

// (n0 / d, n1 / d, n2 / d)
__device__ __forceinline__ void prt_div3(double n0, double n1, double n2, double d, double& q0,
                                         double& q1, double& q2) {
  const bool ok = prt_exp_window(d) && prt_num_ok(n0) && prt_num_ok(n1) && prt_num_ok(n2);
  if (__ballot(!ok) == 0ull) {
    const double r = prt_refined_rcp(d);
    q0 = prt_div_by(n0, d, r);
    q1 = prt_div_by(n1, d, r);
    q2 = prt_div_by(n2, d, r);
  } else {
    q0 = n0 / d;
    q1 = n1 / d;
    q2 = n2 / d;
  }
}

// ---- square root ---------------------------------------------------------------------------------
// The compiler's sqrt(double): correctly rounded, twenty instructions of which ten are the iteration
// (scaling wrapper for x < 2^-767, v_rsq, one Goldschmidt step, two corrections, class test for +-0 / +inf).
// Shorter forms that give the same bits for operands in range exist (tools/ubench/sqrt_fast.hip,
// tools/experiments/) and were measured: the wave-uniform branch that guards them costs what they save.
__device__ __forceinline__ double prt_sqrt(double x) { return sqrt(x); }
// sqrt(max(0, x)) for the lanes with x >= 0; a lane with x < 0 (or NaN) gets some finite value -- every
// caller overrides what it derives from it (the `!(disc >= 0)` rule of the quadratics, the total-reflection
// branch of refract).  max(0, x) is never -0 here: x is a difference of products, b*b - 4ac resp. 1 - ...
__device__ __forceinline__ double prt_sqrt_clamped(double x) { return sqrt(0.0 > x ? 0.0 : x); }
