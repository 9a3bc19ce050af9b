// prt_ops.hpp -- tinygfx/g3d/operations.py as stand-alone device entry points (rows a5 / a12 of
// SURVEY.md section 8a): the same device functions the trace uses (binomial_root, reflect4,
// refract4), one element per lane.  Included by prt_kernels.hip.
#pragma once

// operations.reflect :86-107 on (rows <= 4, n) column vectors
__global__ void __launch_bounds__(PRT_BLOCK)
k_reflect(const double* __restrict__ v, const double* __restrict__ nrm, int rows, int64_t ld, int64_t n,
          double* __restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double d[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
  for (int r = 0; r < rows; ++r) { d[r] = v[r * ld + i]; m[r] = nrm[r * ld + i]; }
  reflect4(d[0], d[1], d[2], d[3], m[0], m[1], m[2], m[3]);
  for (int r = 0; r < rows; ++r) out[r * ld_out + i] = d[r];
}

// operations.refract :110-162: normalises `v` in place like upstream (:125), then Snell / TIR
__global__ void __launch_bounds__(PRT_BLOCK)
k_refract(double* __restrict__ v, const double* __restrict__ nrm, const double* __restrict__ n1,
          const double* __restrict__ n2, double n_global, int rows, int64_t ld, int64_t n,
          double* __restrict__ out, int64_t ld_out, double* __restrict__ index_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double d[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, o[4];
  for (int r = 0; r < rows; ++r) { d[r] = v[r * ld + i]; m[r] = nrm[r * ld + i]; }
  const double len = norm4(d[0], d[1], d[2], d[3]);
  for (int r = 0; r < rows; ++r) { d[r] = d[r] / len; v[r * ld + i] = d[r]; }
  double index;
  refract4(d[0], d[1], d[2], d[3], m[0], m[1], m[2], m[3], n1[i], n2[i], n_global, o[0], o[1], o[2], o[3], index);
  for (int r = 0; r < rows; ++r) out[r * ld_out + i] = o[r];
  index_out[i] = index;
}

// operations.binomial_root :28-63 -> (2, n)
__global__ void __launch_bounds__(PRT_BLOCK)
k_binomial_root(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                int64_t n, double* __restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double p0, p1;
  binomial_root(a[i], b[i], c[i], p0, p1);
  out[i] = p0;
  out[ld_out + i] = p1;
}

// operations.smallest_positive_root :4-25
__global__ void __launch_bounds__(PRT_BLOCK)
k_smallest_positive_root(const double* __restrict__ a, const double* __restrict__ b,
                         const double* __restrict__ c, int64_t n, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double disc = b[i] * b[i] - 4 * a[i] * c[i];
  const double s = sqrt(dmax(0.0, disc));
  const double den = 2 * a[i] + (near0(a[i]) ? 1.0 : 0.0);
  const double r0 = (-b[i] + s) / den, r1 = (-b[i] - s) / den;
  // np.amin propagates NaN; only reached with NaN when an input is NaN
  const double low = (r0 != r0 || r1 != r1) ? (r0 + r1) : dmin(r0, r1);
  const double pick = (r1 >= 0) ? low : r0;
  out[i] = (disc >= 0 && pick >= 0) ? pick : PRT_INF;
}

// operations.element_wise_dot :66-83: out[o] = sum_r m1[o*out_stride + r*red_stride] * m2[same]
__global__ void __launch_bounds__(PRT_BLOCK)
k_dot(const double* __restrict__ m1, const double* __restrict__ m2, int64_t red_len, int64_t red_stride,
      int64_t out_len, int64_t out_stride, double* __restrict__ out) {
  const int64_t o = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (o >= out_len) return;
  double acc = 0.0;
  for (int64_t r = 0; r < red_len; ++r) {
    const int64_t at = o * out_stride + r * red_stride;
    acc = acc + m1[at] * m2[at];
  }
  out[o] = acc;
}

// csg.array_csg (csg.py:13-61) on two blocks of ascending hit lists, one column per ray: stable
// merge (ties: left first), +-1 by list-position parity (right list negated and depth starting at
// 1 for DIFFERENCE), keep rule with the np.roll wrap-around -- the walk csg_merge does for the
// trace, here straight from global memory.  sort_output: survivors first, ascending, then +inf;
// otherwise every merged position keeps its value or +inf.
__global__ void __launch_bounds__(PRT_BLOCK)
k_array_csg(const double* __restrict__ left, int m_left, const double* __restrict__ right, int m_right,
            int64_t ld, int64_t n, int op, int sort_output, double* __restrict__ out, int64_t ld_out) {
  const int64_t col = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (col >= n) return;
  const int total = m_left + m_right;
  int i = 0, j = 0, kept = 0;
  int depth = (op == CSG_DIFFERENCE) ? 1 : 0;
  int prev = depth;  // depth of the last merged entry: the +-1 of complete lists cancel
  double a = m_left ? left[col] : PRT_INF, b = m_right ? right[col] : PRT_INF;
  for (int s = 0; s < total; ++s) {
    const bool take_left = (i < m_left) && ((j >= m_right) || (a <= b));
    const double v = take_left ? a : b;
    const int pos = take_left ? i : j;
    int step = (pos & 1) ? -1 : 1;
    if (op == CSG_DIFFERENCE && !take_left) step = -step;
    depth += step;
    const bool keep = (op == CSG_UNION) ? ((depth != 0) != (prev != 0)) : (depth == 2 || prev == 2);
    prev = depth;
    if (take_left) { ++i; a = (i < m_left) ? left[(int64_t)i * ld + col] : PRT_INF; }
    else { ++j; b = (j < m_right) ? right[(int64_t)j * ld + col] : PRT_INF; }
    if (sort_output) {
      if (keep && v < PRT_INF) { out[(int64_t)kept * ld_out + col] = v; ++kept; }
    } else {
      out[(int64_t)s * ld_out + col] = keep ? v : PRT_INF;
    }
  }
  if (sort_output)
    for (int k = kept; k < total; ++k) out[(int64_t)k * ld_out + col] = PRT_INF;
}

// primitive.intersect / primitive.normal in object space (primitives.py): the raw pair in
// upstream's order (no sort, NaN kept), the unit object-space normal
struct PrimParams { double q[6]; };

__global__ void __launch_bounds__(PRT_BLOCK)
k_primitive_intersect(int type, PrimParams params, const double* __restrict__ rays, int64_t ld, int64_t n,
                      double* __restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double h0, h1;
  primitive_pair(type, params.q, rays[i], rays[ld + i], rays[2 * ld + i], rays[4 * ld + i], rays[5 * ld + i],
                 rays[6 * ld + i], h0, h1);
  out[i] = h0;
  out[ld_out + i] = h1;
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_primitive_normal(int type, PrimParams params, const double* __restrict__ pts, int64_t ld, int64_t n,
                   double* __restrict__ out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double ax, ay, az;
  object_normal(type, params.q, pts[i], pts[ld + i], pts[2 * ld + i], ax, ay, az);
  out[i] = ax;
  out[ld_out + i] = ay;
  out[2 * ld_out + i] = az;
  out[3 * ld_out + i] = 0.0;
}
