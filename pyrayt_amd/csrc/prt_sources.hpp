// prt_sources.hpp -- ray sources on the device: the patterns of
// pyrayt/components.py:481-654 emitted straight into an HBM ray set.  Included by prt_kernels.hip.
#pragma once
// ------------------------------------------------------------------------------------------------
// sources: components.py:481-654 on the device
// ------------------------------------------------------------------------------------------------
struct DevSource {
  int kind;
  double p0, p1, p2, wavelength;
  double world[16];
  unsigned long long seed;
};

// np.linspace(start, stop, num)[i] (endpoint=True): i*step + start, the last element forced to
// stop; (stop-start)/(num-1) == 0 falls back to (i/div)*delta like numpy does
__device__ __forceinline__ double linspace_at(double start, double stop, int64_t num, int64_t i) {
  if (num <= 1) return start;
  if (i == num - 1) return stop;
  const double div = (double)(num - 1), delta = stop - start;
  const double step = delta / div;
  if (step == 0) return ((double)i / div) * delta + start;
  return (double)i * step + start;
}

// counter-based uniform in [0,1): SplitMix64 finaliser over (seed, ray, stream)
__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned long long i, unsigned k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (4ull * i + k + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_source(DevSource src, int64_t n_total, int64_t first, int64_t count, int64_t id_first,
         double* __restrict__ rays, int64_t ld, int64_t col_offset) {
  const int64_t k = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (k >= count) return;
  const int64_t i = first + k;  // ray number within the source
  const double two_pi = 2 * 3.141592653589793;
  double o[4] = {0, 0, 0, 1}, d[4] = {0, 0, 0, 0};
  double intensity = 100.0;
  switch (src.kind) {
    case PRT_SRC_LINE:  // components.py:516-530
      if (n_total > 1) o[1] = linspace_at(-src.p0 / 2, src.p0 / 2, n_total, i);
      d[0] = 1;
      break;
    case PRT_SRC_CIRCLE: {  // :545-558
      const double theta = linspace_at(0.0, two_pi, n_total, i);
      o[1] = src.p0 / 2 * sin(theta);
      o[2] = src.p0 / 2 * cos(theta);
      d[0] = 1;
    } break;
    case PRT_SRC_CONE: {  // :575-585
      if (n_total > 1) {
        const double az = two_pi * (double)i / (double)n_total;
        d[1] = sin(src.p0) * sin(az);
        d[2] = sin(src.p0) * cos(az);
      }
      d[0] = cos(src.p0);
    } break;
    case PRT_SRC_WEDGE: {  // :600-613
      const double a = linspace_at(-src.p0 / 2, src.p0 / 2, n_total, i);
      d[0] = cos(a);
      d[1] = sin(a);
    } break;
    default: {  // PRT_SRC_LAMP :637-654, inverse-CDF polar angle of :56-70
      const double theta = acos(1 - uniform01(src.seed, i, 0) * (1 - cos(src.p2)));
      const double phi = uniform01(src.seed, i, 1) * two_pi;
      o[1] = src.p0 * (uniform01(src.seed, i, 2) - 0.5);
      o[2] = src.p1 * (uniform01(src.seed, i, 3) - 0.5);
      d[0] = cos(theta);
      d[1] = sin(theta) * cos(phi);
      d[2] = sin(theta) * sin(phi);
      intensity = 100.0 * cos(theta);
    } break;
  }
  // world transform (dgemm-style FMA chain) and unit direction (components.py:490-495)
  double wo[4], wd[4];
  for (int r = 0; r < 4; ++r) {
    wo[r] = row_dot(src.world, r, o[0], o[1], o[2], o[3]);
    wd[r] = row_dot(src.world, r, d[0], d[1], d[2], d[3]);
  }
  const double len = norm4(wd[0], wd[1], wd[2], wd[3]);
  const int64_t c = col_offset + k;
  for (int r = 0; r < 4; ++r) {
    rays[r * ld + c] = wo[r];
    rays[(4 + r) * ld + c] = wd[r] / len;
  }
  rays[8 * ld + c] = 0.0;
  rays[9 * ld + c] = intensity;
  rays[10 * ld + c] = src.wavelength;
  rays[11 * ld + c] = 1.0;
  rays[12 * ld + c] = (double)(id_first + k);
}
