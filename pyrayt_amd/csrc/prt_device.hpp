// prt_device.hpp -- per-ray device arithmetic of the ray-propagation path (gfx950 / CDNA4).
//
// One ray per lane.  Everything here is float64 and deliberately *not* contracted into FMAs
// (the library is built with -ffp-contract=off) so that a lane computes what one column of
// the reference's numpy expressions computes; the exceptions are the 4x4 transforms, which
// numpy hands to a BLAS dgemm that accumulates with FMAs -- those use explicit fma() chains.
//
// np.isclose(x, 0) == |x| <= 1e-8 ; np.isclose(x, h) == |x - h| <= 1e-8 + 1e-5 |h|.
//
// Reference (paths under the PyRayT tree):
//   tinygfx/g3d/primitives.py  Sphere :241-296  Paraboloid :320-419  Plane :436-498
//                              Cube :516-602    Cylinder :650-741
//   tinygfx/g3d/operations.py  binomial_root :28-63  reflect :86-107  refract :110-162
//   tinygfx/g3d/world_objects.py  intersect :360-383  get_world_normals :401-418
//   tinygfx/g3d/csg.py         array_csg :13-61  CSGSurface.intersect :118-160
//   pyrayt/materials.py        :47-50 :58-62 :70-75 :112-118 :136-145
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prt_math.hpp"

#ifndef PRT_BLOCK
#define PRT_BLOCK 256
#endif
#define PRT_INF (__builtin_inf())

// ---- device-side scene ----------------------------------------------------------------------
// All of it is wave-uniform data: it is fetched with scalar loads into SGPRs (the constant
// cache), which costs no VGPRs, no LDS bandwidth and no bank conflicts.  LDS is spent on the
// thing that is genuinely per-lane and dynamically indexed: the CSG hit lists.
struct DevPrim {
  double minv[16];
  double params[6];
  double coef[6];
  double surface_id;  // as float64: that is how the result frame stores it
  int32_t type, mat_kind, normal_scale, pad;
};

// the same table seen through the constant address space (scalar loads when the index is uniform)
typedef const __attribute__((address_space(4))) DevPrim* ConstPrimPtr;
__device__ __forceinline__ ConstPrimPtr const_prim(const DevPrim* prims, int index) {
  return (ConstPrimPtr)(unsigned long long)(prims + index);
}

enum { I_LEAF = 0, I_CSG = 1, I_ROOT = 2, I_BOX = 3, I_CHAIN = 4 };
// I_BOX steps ("jump a0 steps ahead if no lane of the wave wants what follows"), a1 says what is asked:
//   BOX_TEST  can the ray reach the box (data[0..5]) before its nearest hit so far?   (component cull step)
//   BOX_PICK  does the wave mostly run AGAINST the program's axis (a2)?  the mirror image follows; if not,
//             jump over it to the program proper
//   BOX_JUMP  nothing is wanted: the end of the mirror image, over the program proper
enum { BOX_TEST = 0, BOX_PICK = 1, BOX_JUMP = 2, BOX_LINE = 3 };  // (LINE: render programs, see may_reach)
enum { OPER_REGA = 0, OPER_REGB = 1, OPER_LDS = 2 };
enum { CSG_UNION = 1, CSG_INTERSECT = 2, CSG_DIFFERENCE = 3 };
enum { PRIM_SPHERE = 0, PRIM_CYLINDER = 1, PRIM_PLANE = 2, PRIM_CUBE = 3, PRIM_PARABOLOID = 4 };
enum { MAT_NONE = 0, MAT_ABSORBER = 1, MAT_MIRROR = 2, MAT_CONST = 3, MAT_SELLMEIER = 4, MAT_TABLE = 5, MAT_HOST = 6 };

// One step of the linearised (post-order) component program.  A step is self-contained: a LEAF
// step carries its primitive's kind, parameters and the three used rows of M^-1, a CSG step its
// cull box, so the interpreter fetches a whole step with one batch of scalar loads (one round
// trip to the constant cache per step instead of one per field group it happens to touch).
struct DevInstr {
  int32_t kind;   // I_*
  int32_t a0;     // LEAF: prim        CSG: operation    ROOT: operand mode
  int32_t a1;     // LEAF: dst mode    CSG: left mode    ROOT: lds base
  int32_t a2;     // LEAF: lds base    CSG: left base    ROOT: list length
  int32_t a3;     //                   CSG: left length
  int32_t a4;     //                   CSG: right mode
  int32_t a5;     //                   CSG: right base
  int32_t a6;     //                   CSG: right length
  int32_t a7;     //                   CSG: out base
  int32_t pad[2]; // CSG: [0] root node of a component, [1] cull box implied (see csg_node); LEAF: [0] == 1 skippable right leaf, == 2 a whole component
  int32_t type;   // LEAF: PRIM_*
  double data[18];// LEAF: params[0..5], M^-1 rows 0..2 [6..17]   CSG: cull box [0..5]
};

// A whole component in one step (trace programs): a left-deep chain of two or three leaf surfaces,
//     (leaf0 op1 leaf1)            or            ((leaf0 op1 leaf1) op2 leaf2),
// which is what every part factory builds (components.py:73-198, 269-468).  The record takes three
// DevInstr slots; `shape` selects a body compiled for the leaves' primitive types, so the whole
// component is one basic block per stage: no step dispatch, no hit lists in LDS, and the
// independent transform -> quadratic -> sqrt -> divide chains of sibling leaves interleave.
struct DevChain {
  int32_t kind;          // I_CHAIN
  int32_t shape;         // CHAIN_* below
  int32_t n_leaves;      // 2 or 3
  int32_t op1, op2;      // CSG_*
  int32_t prim[3];
  int32_t implied1, implied2;  // cull box implied by the survivors (see csg_keep)
  int32_t clearance;           // a third leaf that is a cylinder may be skipped for a wave whose chords it provably leaves alone (chord_inside_cylinder)
  int32_t intervals;           // every node INTERSECT with an implied box: the interval form applies (chain_candidate)
  double box1[6], box2[6];     // upstream cull boxes of the two nodes (csg.py:126-128)
  double leaf[3][18];          // params[0..5], M^-1 rows 0..2 [6..17]
};
static_assert(sizeof(DevChain) == 3 * sizeof(DevInstr), "a chain record spans three step slots");
enum { CHAIN_SLOTS = 3 };
// leaf types per shape, in chain order (S sphere, C cylinder, P plane, Q cube, B paraboloid)
enum {
  CHAIN_SSC = 0, CHAIN_SSQ, CHAIN_CSS, CHAIN_QSS, CHAIN_QQQ,                        // three leaves
  CHAIN_SC, CHAIN_SQ, CHAIN_CS, CHAIN_QS, CHAIN_CB, CHAIN_QB, CHAIN_PC, CHAIN_PQ, CHAIN_BC,  // two
  CHAIN_SHAPES
};

struct Ray8 {
  double ox, oy, oz, ow, dx, dy, dz, dw;
  // (set by nearest_hit(); a ray built from its eight numbers alone has gated == false: no shortcut)
  bool gated;                 // the ray may take shortcuts if well_formed() says so
  bool any_w;                 // its w components are not known to be 1 / 0 (well_formed() looks at them)
  bool lex;                   // the program may visit components out of list order: see beats()
  unsigned long long* paths;  // PRT_TRACE_COUNT_PATHS counters, or null
};

// ---- the one gate of every shortcut ----------------------------------------------------------------
// The kernels replace four of the reference's predicates by cheaper ones: the CSG cull box implied by a
// node's survivors (csg_keep), the component cull steps (may_reach), the right-leaf skip of the
// interpreter and the third-leaf skip of the chain steps.  Each is argued geometrically -- "a finite
// positive entry of a hit list is a point of the surface, hence of its box" -- and the argument holds
// for a ray that is what a RaySet holds (pyrayt/_pyrayt.py:29-36, sources and materials keep it so):
//   * homogeneous coordinates w = 1 (origin) and 0 (direction): the object-space ray is then the image
//     M^-1 of the world ray (world_objects.py:367-369 multiplies the translation by w);
//   * a direction that is not short, here |d|^2 >= 0.81 (RaySet directions are unit vectors): upstream's
//     `isclose(.., 0)` branches use absolute 1e-8 thresholds on object-space quantities and fire for any
//     SHORT direction, parallel or not, and what they then report is not a point of the surface; the
//     scene compiler allows a shortcut only where |d|^2 >= 0.81 keeps every leaf's |d_obj|^2 >= 1e-3
//     (short_direction_bound).  (A NaN direction fails the test; an infinite or NaN origin needs no
//     clause of its own: hit lists of such a ray hold no robust pair of survivors, and may_reach answers
//     "yes" whenever a comparison involves a NaN.)
// Any other ray takes none of them: every component, every upstream cull box exactly, every leaf.
// One predicate for all of them; DESIGN.md section 4 has the argument per shortcut.  (It is a pure function
// of the ray and is written out at every site: holding its value across the hit phase instead costs the
// generation kernel its register allocation -- the lane mask lives in SGPRs, of which the kernel has none
// to spare -- while the compiler is free to share it between neighbouring sites as it is.)
#define kWellFormedLen2Lo 0.81
__device__ __forceinline__ bool well_formed(const Ray8& r) {
  if (!r.gated) return false;
  const double len2 = (r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz;
  bool ok = len2 >= kWellFormedLen2Lo;  // (a NaN fails)
  if (r.any_w) ok = ok && r.ow == 1.0 && r.dw == 0.0;
  return ok;
}
// The running nearest hit over components (pyrayt/_pyrayt.py:380-386): a candidate replaces it when it
// is strictly nearer -- so among equal parameters the component that comes first in the LIST keeps it.
// A program that visits components in list order gets that from the strict '<' alone.  One whose cull
// steps are grouped by position (prt_scene.hpp) visits them in another order and compares
// lexicographically on (t, primitive index) instead: primitives are numbered component by component
// (checked by the scene compiler), so that is (t, list index).  In a list-order program the second clause
// never fires.
__device__ __forceinline__ bool beats(const Ray8& ray, double t, int prim, double best_t, int best_prim) {
  return t < best_t || (ray.lex && t == best_t && prim < best_prim);
}
// PRT_TRACE_COUNT_PATHS: one atomic per wave and site; `paths` is null (a compile-time constant in the
// fused kernel, which carries none of this) unless the trace asked for the counts
__device__ __forceinline__ void count_paths(unsigned long long* paths, int k, bool flag) {
  if (paths == nullptr) return;
  const unsigned long long m = __ballot(flag);
  if (m != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1)
    atomicAdd(&paths[k], (unsigned long long)__popcll(m));
}

// a sorted pair of hit parameters with the primitive that produced it
struct Pair {
  double t0, t1;
  int prim;
};

__device__ __forceinline__ bool near0(double x) { return fabs(x) <= 1e-8; }
__device__ __forceinline__ bool close_to(double x, double h) {
  return fabs(x - h) <= 1e-8 + 1e-5 * fabs(h);
}
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
// NaN only arises from 0/0 for zero-direction rays; numpy's sort moves it behind +inf and the
// tracer masks it with `> 0`, so mapping it to +inf at the leaf is observationally the same.
__device__ __forceinline__ double nan_to_inf(double x) { return x == x ? x : PRT_INF; }

// row r of (M . v) with the accumulation order of a dgemm micro-kernel: fma chain from 0
// (M: pointer to 16 doubles in any address space -- the shading reads the table through the
// constant address space so that a wave-uniform record is fetched with scalar loads)
template <class M>
__device__ __forceinline__ double row_dot(M m, int r, double x, double y, double z, double w) {
  double acc = m[4 * r + 0] * x;
  acc = fma(m[4 * r + 1], y, acc);
  acc = fma(m[4 * r + 2], z, acc);
  acc = fma(m[4 * r + 3], w, acc);
  return acc;
}
// row r of (M^T . v)
template <class M>
__device__ __forceinline__ double col_dot(M m, int r, double x, double y, double z, double w) {
  double acc = m[0 + r] * x;
  acc = fma(m[4 + r], y, acc);
  acc = fma(m[8 + r], z, acc);
  acc = fma(m[12 + r], w, acc);
  return acc;
}

__device__ __forceinline__ double norm4(double x, double y, double z, double w) {
  return prt_sqrt(((x * x + y * y) + z * z) + w * w);
}
__device__ __forceinline__ double norm3(double x, double y, double z) {
  return prt_sqrt((x * x + y * y) + z * z);
}

// Quotients that share a denominator.  Three of them (the normalisations) go through one refined reciprocal
// (prt_math.hpp), the same bits as separate `/`; for a pair the operand-window check costs what the shared
// reciprocal saves (measured in round 2), so a pair is two plain divisions.
__device__ __forceinline__ void div2(double n0, double n1, double d, double& q0, double& q1) {
  q0 = n0 / d;
  q1 = n1 / d;
}
__device__ __forceinline__ void div3(double n0, double n1, double n2, double d, double& q0, double& q1,
                                     double& q2) {
  prt_div3(n0, n1, n2, d, q0, q1, q2);
}

// ---- quadratic with the reference's degenerate branches (operations.py:28-63) ----------------
__device__ __forceinline__ void binomial_root(double a, double b, double c, double& p0, double& p1) {
  const double disc = b * b - 4 * a * c;
  const bool lin = near0(a);
  const double s = prt_sqrt_clamped(disc);
  const double den = 2 * a + (lin ? 1.0 : 0.0);
  div2(-b + s, -b - s, den, p0, p1);
  if (!(disc >= 0)) { p0 = PRT_INF; p1 = PRT_INF; }
  if (lin) {
    const double root = -c / (b + (b == 0 ? 1.0 : 0.0));
    p0 = root; p1 = root;
    if (near0(b)) {
      p0 = (c <= 0) ? -PRT_INF : PRT_INF;
      p1 = PRT_INF;
    }
  }
}

// crossing parameters of the planes z = lo and z = hi (primitives.py:683-703 / :372-390)
__device__ __forceinline__ void z_slab(double oz, double dz, double lo, double hi, double& c0,
                                       double& c1) {
  const bool par = near0(dz);
  const double den = dz + (par ? 1.0 : 0.0);
  div2(lo - oz, hi - oz, den, c0, c1);
  if (par) {
    c0 = (oz >= lo && oz <= hi) ? -PRT_INF : PRT_INF;
    c1 = PRT_INF;
  }
}

// [max(lo), min(hi)] of two unsorted pairs; miss unless lo <= hi (primitives.py:705-711)
__device__ __forceinline__ void overlap(double a0, double a1, double b0, double b1, double& h0,
                                        double& h1) {
  const double lo = dmax(dmin(a0, a1), dmin(b0, b1));
  const double hi = dmin(dmax(a0, a1), dmax(b0, b1));
  const bool ok = lo <= hi;
  h0 = ok ? lo : PRT_INF;
  h1 = ok ? hi : PRT_INF;
}

// one axis of the cube / plane-patch slab test (primitives.py:531-565 / :454-469)
// (`inside` is only looked at for a lane whose direction has no component along the axis)
__device__ __forceinline__ void axis_slab(double o, double d, double lo, double hi, bool inside,
                                          double& s_lo, double& s_hi) {
  const bool z = near0(d);
  double first, second;
  const double den = d + (z ? 1.0 : 0.0);
  div2(-(o - lo), -(o - hi), den, first, second);
  if (z) {
    first = inside ? -PRT_INF : PRT_INF;
    second = PRT_INF;
  }
  s_lo = dmin(first, second);
  s_hi = dmax(first, second);
}

// world-space cube test used as the CSG cull predicate (csg.py:126-128): "touches" iff the
// test yields a finite parameter, i.e. enter < leave with either one finite.
__device__ __forceinline__ void cube_pair(const double* __restrict__ span, double ox, double oy,
                                          double oz, double dx, double dy, double dz, double& h0,
                                          double& h1) {
  double lx, hx, ly, hy, lz, hz;
  axis_slab(ox, dx, span[0], span[1], ox <= span[1] && ox >= span[0], lx, hx);
  axis_slab(oy, dy, span[2], span[3], oy <= span[3] && oy >= span[2], ly, hy);
  axis_slab(oz, dz, span[4], span[5], oz <= span[5] && oz >= span[4], lz, hz);
  const double enter = dmax(dmax(lx, ly), lz);
  const double leave = dmin(dmin(hx, hy), hz);
  const bool ok = enter < leave;  // strict (primitives.py:578)
  h0 = ok ? enter : PRT_INF;
  h1 = ok ? leave : PRT_INF;
}

__device__ __forceinline__ bool is_finite(double x) { return fabs(x) < PRT_INF; }

__device__ __forceinline__ bool box_touched(const double* __restrict__ span, const Ray8& r) {
  double h0, h1;
  cube_pair(span, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, h0, h1);
  return is_finite(h0) || is_finite(h1);
}

// ---- primitive.intersect in object space (primitives.py Sphere :241-271, Cylinder :650-712,
// Plane :436-492, Cube :516-581, Paraboloid :320-399): the raw pair, in upstream's order, NaN where
// upstream yields NaN (0/0 of a zero direction)
__device__ __forceinline__ void primitive_pair(int type, const double* __restrict__ q, double ox,
                                               double oy, double oz, double dx, double dy,
                                               double dz, double& h0, double& h1) {
  switch (type) {
    case PRIM_SPHERE: {  // primitives.py:241-271 (no guard on a == 0)
      const double a = (dx * dx + dy * dy) + dz * dz;
      const double b = 2 * ((dx * ox + dy * oy) + dz * oz);
      const double c = ((ox * ox + oy * oy) + oz * oz) - q[0] * q[0];
      const double disc = b * b - 4 * a * c;
      const double s = prt_sqrt_clamped(disc);
      const double den = 2 * a;
      div2(-b + s, -b - s, den, h0, h1);
      if (!(disc >= 0)) { h0 = PRT_INF; h1 = PRT_INF; }
    } break;
    case PRIM_CYLINDER: {  // primitives.py:650-712
      const double a = dx * dx + dy * dy;
      const double b = 2 * (dx * ox + dy * oy);
      const double c = (ox * ox + oy * oy) - q[0] * q[0];
      double s0, s1, c0, c1;
      binomial_root(a, b, c, s0, s1);
      z_slab(oz, dz, q[1], q[2], c0, c1);
      overlap(s0, s1, c0, c1, h0, h1);
    } break;
    case PRIM_PLANE: {  // primitives.py:436-492, the hit is reported twice
      const double hw = q[0] / 2, hl = q[1] / 2;
      double lx, hx, ly, hy;
      axis_slab(ox, dx, hw, -hw, fabs(ox) <= hw, lx, hx);
      axis_slab(oy, dy, hl, -hl, fabs(oy) <= hl, ly, hy);
      const double enter = dmax(lx, ly), leave = dmin(hx, hy);
      const bool skew = near0(dz);
      double t = -oz / (dz + (skew ? 1.0 : 0.0));
      if (skew) t = PRT_INF;
      if (!(t >= enter && t <= leave)) t = PRT_INF;
      h0 = t; h1 = t;
    } break;
    case PRIM_CUBE: {  // primitives.py:516-581
      cube_pair(q, ox, oy, oz, dx, dy, dz, h0, h1);
    } break;
    default: {  // PRIM_PARABOLOID, primitives.py:320-399
      const double f4 = 4 * q[0];
      const double a = dx * dx + dy * dy;
      const double b = 2 * (ox * dx + oy * dy) - f4 * dz;
      const double c = (ox * ox + oy * oy) - f4 * oz;
      const double disc = b * b - 4 * a * c;
      const bool lin = near0(a);
      const double s = prt_sqrt_clamped(disc);
      double p0, p1;
      const double den = 2 * a + (lin ? 1.0 : 0.0);
      div2(-b + s, -b - s, den, p0, p1);
      if (!(disc >= 0)) { p0 = PRT_INF; p1 = PRT_INF; }
      if (lin) {
        p0 = -c / (b + (near0(b) ? 1.0 : 0.0));
        p1 = (dz >= 0) ? PRT_INF : -PRT_INF;
      }
      double c0, c1;
      z_slab(oz, dz, 0.0, q[1], c0, c1);
      overlap(p0, p1, c0, c1, h0, h1);
    } break;
  }
}

// ---- TracerSurface.intersect: world -> object, primitive test, ascending pair -----------------
// `type`, q = params[6] and m = rows 0..2 of M^-1 (12 values) come from the step record.
// the primitive test on an object-space ray, reduced to the ascending pair
__device__ __forceinline__ void object_pair(int type, const double* __restrict__ q, double ox, double oy, double oz,
                                            double dx, double dy, double dz, double& t0, double& t1) {
  double h0, h1;
  primitive_pair(type, q, ox, oy, oz, dx, dy, dz, h0, h1);
  h0 = nan_to_inf(h0);  // (0 / 0 of a zero direction)
  h1 = nan_to_inf(h1);
  t0 = fmin(h0, h1);  // NaN-free here: v_min / v_max order the pair like np.sort (signed zeros compare equal)
  t1 = fmax(h0, h1);
}
__device__ __forceinline__ void surface_pair(int type, const double* __restrict__ q,
                                             const double* __restrict__ m, const Ray8& r,
                                             double& t0, double& t1) {
  const double ox = row_dot(m, 0, r.ox, r.oy, r.oz, r.ow);
  const double oy = row_dot(m, 1, r.ox, r.oy, r.oz, r.ow);
  const double oz = row_dot(m, 2, r.ox, r.oy, r.oz, r.ow);
  const double dx = row_dot(m, 0, r.dx, r.dy, r.dz, r.dw);
  const double dy = row_dot(m, 1, r.dx, r.dy, r.dz, r.dw);
  const double dz = row_dot(m, 2, r.dx, r.dy, r.dz, r.dw);
  object_pair(type, q, ox, oy, oz, dx, dy, dz, t0, t1);
}

// ---- primitive.normal in object space (primitives.py Sphere :273-296, Paraboloid :401-419,
// Plane :494-498, Cube :583-602, Cylinder :714-741): unit normal (w = 0) at an object-space point.
// For the stand-alone entry point only.  world_normal() below carries the same switch inline on
// purpose: routing it through this function changes the register allocation of the generation
// kernel enough to spill (measured: 12 B of scratch per lane), so the two are kept side by side.
__device__ __forceinline__ void object_normal(int type, const double* __restrict__ q, double lx,
                                              double ly, double lz, double& ax, double& ay,
                                              double& az) {
  bool normalise = true;
  switch (type) {
    case PRIM_SPHERE:  // primitives.py:291-293
      ax = lx; ay = ly; az = lz;
      break;
    case PRIM_CYLINDER:  // :722-738
      ax = lx; ay = ly; az = 0.0;
      if (close_to(lz, q[1])) { ax = 0.0; ay = 0.0; az = -1.0; }
      if (close_to(lz, q[2])) { ax = 0.0; ay = 0.0; az = 1.0; }
      break;
    case PRIM_PLANE:  // :496-498
      ax = 0.0; ay = 0.0; az = 1.0;
      normalise = false;
      break;
    case PRIM_CUBE:  // :593-599 (a point on no face gives 0/0 = NaN, as upstream)
      ax = close_to(lx, q[1]) ? 1.0 : (close_to(lx, q[0]) ? -1.0 : 0.0);
      ay = close_to(ly, q[3]) ? 1.0 : (close_to(ly, q[2]) ? -1.0 : 0.0);
      az = close_to(lz, q[5]) ? 1.0 : (close_to(lz, q[4]) ? -1.0 : 0.0);
      break;
    default:  // PRIM_PARABOLOID :405-418
      ax = lx; ay = ly; az = -2 * q[0];
      if (close_to(lz, q[1])) { ax = 0.0; ay = 0.0; az = 1.0; }
      break;
  }
  if (normalise) {
    const double len = norm4(ax, ay, az, 0.0);
    div3(ax, ay, az, len, ax, ay, az);
  }
}

// ---- TracerSurface.get_world_normals (world_objects.py:401-418) -------------------------------
// p = world-space point (4 comps).  Returns the world-space unit normal times normal_scale.
template <class PrimPtr>
__device__ __forceinline__ void world_normal(PrimPtr p, double px, double py, double pz, double pw,
                                             double& nx, double& ny, double& nz) {
  const auto m = p->minv;
  const double lx = row_dot(m, 0, px, py, pz, pw);
  const double ly = row_dot(m, 1, px, py, pz, pw);
  const double lz = row_dot(m, 2, px, py, pz, pw);
  const auto q = p->params;
  double ax, ay, az;  // object-space normal (w = 0)
  bool normalise = true;
  switch (p->type) {
    case PRIM_SPHERE:  // primitives.py:291-293
      ax = lx; ay = ly; az = lz;
      break;
    case PRIM_CYLINDER:  // :722-738
      ax = lx; ay = ly; az = 0.0;
      if (close_to(lz, q[1])) { ax = 0.0; ay = 0.0; az = -1.0; }
      if (close_to(lz, q[2])) { ax = 0.0; ay = 0.0; az = 1.0; }
      break;
    case PRIM_PLANE:  // :496-498
      ax = 0.0; ay = 0.0; az = 1.0;
      normalise = false;
      break;
    case PRIM_CUBE:  // :593-599 (a point on no face gives 0/0 = NaN, as upstream)
      ax = close_to(lx, q[1]) ? 1.0 : (close_to(lx, q[0]) ? -1.0 : 0.0);
      ay = close_to(ly, q[3]) ? 1.0 : (close_to(ly, q[2]) ? -1.0 : 0.0);
      az = close_to(lz, q[5]) ? 1.0 : (close_to(lz, q[4]) ? -1.0 : 0.0);
      break;
    default:  // PRIM_PARABOLOID :405-418
      ax = lx; ay = ly; az = -2 * q[0];
      if (close_to(lz, q[1])) { ax = 0.0; ay = 0.0; az = 1.0; }
      break;
  }
  if (normalise) {
    const double len = norm4(ax, ay, az, 0.0);
    ax /= len; ay /= len; az /= len;
  }
  double wx = col_dot(m, 0, ax, ay, az, 0.0);
  double wy = col_dot(m, 1, ax, ay, az, 0.0);
  double wz = col_dot(m, 2, ax, ay, az, 0.0);
  const double len = norm4(wx, wy, wz, 0.0);
  const double sgn = (double)p->normal_scale;
  div3(wx, wy, wz, len, wx, wy, wz);
  nx = wx * sgn;
  ny = wy * sgn;
  nz = wz * sgn;
}

// ---- materials --------------------------------------------------------------------------------
// A user-defined glass (PRT_MAT_TABLE): Glass.index_at (materials.py:88-99) is host code, evaluated by the caller
// on the distinct wavelengths of its rays; the primitive record carries the device addresses of the material's
// ascending wavelengths (coef[0], as bits) and their indices (coef[2]), the entry count (coef[1]) and the index
// of a NaN wavelength (coef[3]).  Exact look-up: a wavelength that is not there is reported, not interpolated.
template <class PrimPtr>
__device__ __forceinline__ double table_index(PrimPtr p, double wavelength, bool& found) {
  const double* __restrict__ lam = reinterpret_cast<const double*>((unsigned long long)__double_as_longlong(p->coef[0]));
  const double* __restrict__ idx = reinterpret_cast<const double*>((unsigned long long)__double_as_longlong(p->coef[2]));
  const int count = (int)p->coef[1];
  int lo = 0, hi = count;  // first entry that is not below the wavelength
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (lam[mid] < wavelength) lo = mid + 1; else hi = mid;
  }
  found = lo < count && lam[lo] == wavelength;
  double n = found ? idx[lo] : __builtin_nan("");
  if (wavelength != wavelength) { found = true; n = p->coef[3]; }
  return n;
}

template <class PrimPtr>
__device__ __forceinline__ double glass_index(PrimPtr p, double wavelength, bool& found) {
  found = true;
  if (p->mat_kind == MAT_CONST) return p->coef[0];  // materials.py:112-118
  if (p->mat_kind == MAT_TABLE) return table_index(p, wavelength, found);
  const auto k = p->coef;                             // materials.py:136-145
  const double w2 = wavelength * wavelength;
  return prt_sqrt(((1 + (k[0] * w2) / (w2 - k[3])) + (k[1] * w2) / (w2 - k[4])) +
                  (k[2] * w2) / (w2 - k[5]));
}

// operations.reflect (operations.py:104-107): v - (2 n) (v.n), in place
__device__ __forceinline__ void reflect4(double& dx, double& dy, double& dz, double& dw, double nx,
                                         double ny, double nz, double nw) {
  const double dot = ((dx * nx + dy * ny) + dz * nz) + dw * nw;
  dx = dx - (2 * nx) * dot;
  dy = dy - (2 * ny) * dot;
  dz = dz - (2 * nz) * dot;
  dw = dw - (2 * nw) * dot;
}

// operations.refract (operations.py:110-162) for a direction v that is already normalised
// (:125): exit detection by the sign of v.n, vector Snell with the total-internal-reflection
// fallback, renormalisation, index update.  n1 = index the ray travels in, n2 = index of the
// medium behind the surface, n_global = index used instead of n2 when the ray is leaving.
__device__ __forceinline__ void refract4(double vx, double vy, double vz, double vw, double nx,
                                         double ny, double nz, double nw, double n1, double n2_in,
                                         double n_global, double& ox, double& oy, double& oz,
                                         double& ow, double& index_out) {
  const double cos_p = ((vx * nx + vy * ny) + vz * nz) + vw * nw;
  const double cos_n = ((vx * -nx + vy * -ny) + vz * -nz) + vw * -nw;
  const bool leaving = cos_p > 0;
  const double n2 = leaving ? n_global : n2_in;
  const double mx = leaving ? -nx : nx, my = leaving ? -ny : ny, mz = leaving ? -nz : nz,
               mw = leaving ? -nw : nw;
  const double r = n1 / n2;
  const double cos1 = leaving ? cos_p : cos_n;
  const double radicand = 1 - (r * r) * (1 - cos1 * cos1);
  const double cos2 = prt_sqrt_clamped(radicand);  // (only looked at when radicand > 0)
  double ux, uy, uz, uw;
  if (radicand > 0) {
    const double k = r * cos1 - cos2;
    ux = r * vx + k * mx; uy = r * vy + k * my; uz = r * vz + k * mz; uw = r * vw + k * mw;
  } else {  // total internal reflection
    const double k = 2 * cos1;
    ux = vx + k * mx; uy = vy + k * my; uz = vz + k * mz; uw = vw + k * mw;
  }
  const double ulen = norm4(ux, uy, uz, uw);
  div3(ux, uy, uz, ulen, ox, oy, oz);
  ow = uw;  // (+-0) / ulen = +-0 for the finite positive ulen of a non-degenerate ray
  if (uw != 0.0 || !(ulen > 0.0 && ulen < PRT_INF)) ow = uw / ulen;
  index_out = (radicand > 0) ? n2 : n1;
}

// material.trace for a ray whose origin (px..pw) already sits on the surface.
// d (4 comps) and index are updated in place.  Returns 0, or the PRT_ERR_* the ray raises: UNTRACABLE for a
// surface without a traceable material (and for a caller-shaded one, which only prt_interact serves),
// WAVELENGTH for a table glass that does not hold the ray's wavelength.
// (ux, uy, uz) is the incoming direction already divided by norm3(dx,dy,dz) -- the record row's
// tilt columns.  When dw == 0, norm4(d) is bit-identical to norm3(d) (adding +0 is exact), so
// the refraction's own normalisation d / |d| (operations.py:125) IS that vector and its
// square root and divisions are not repeated.
template <class PrimPtr>
__device__ __forceinline__ int shade(PrimPtr p, double px, double py, double pz, double pw, double& dx,
                                     double& dy, double& dz, double& dw, double wavelength,
                                     double& index, double tx, double ty, double tz) {
  const int kind = p->mat_kind;
  if (kind == MAT_NONE || kind == MAT_HOST) return -5;  // PRT_ERR_UNTRACABLE
  if (kind == MAT_ABSORBER) {  // materials.py:47-50
    dx = 0.0; dy = 0.0; dz = 0.0; dw = 0.0;
    return 0;
  }
  double nx, ny, nz;
  world_normal(p, px, py, pz, pw, nx, ny, nz);
  const double nw = 0.0 * (double)p->normal_scale;
  if (kind == MAT_MIRROR) {
    reflect4(dx, dy, dz, dw, nx, ny, nz, nw);
    return 0;
  }
  // glass: operations.py:110-162, n_global = 1 always
  bool found;
  const double n_mat = glass_index(p, wavelength, found);
  double vx = tx, vy = ty, vz = tz, vw = dw;  // dw == 0: (+-0) / |d| = +-0
  if (dw != 0.0) {
    const double len = norm4(dx, dy, dz, dw);
    vx = dx / len; vy = dy / len; vz = dz / len; vw = dw / len;
  }
  refract4(vx, vy, vz, vw, nx, ny, nz, nw, index, n_mat, 1.0, dx, dy, dz, dw, index);
  return found ? 0 : -6;  // PRT_ERR_WAVELENGTH
}

// ---- per-lane hit lists in LDS -------------------------------------------------------------------
// Slot s of lane `tid` lives at lds_dyn[s * PRT_BLOCK + tid] (float64 parameter) and, behind
// the `slots` parameter rows, at the same position of an int32 array (primitive index): a
// lane's slots are PRT_BLOCK elements apart, so whatever (per-lane, data dependent) slot index
// the lanes of a wave use, lane l always touches bank (2l mod 64): conflict free by
// construction.  Accesses go through the __shared__ array itself so that they compile to
// ds_read/ds_write (a pointer kept in a struct degrades them to flat_load/flat_store).
extern __shared__ double lds_dyn[];

struct LaneLists {
  int total;  // slots (rows) per lane over all rays a lane carries: the int32 rows start behind them
  int off;    // first slot of this ray's lists
  __device__ __forceinline__ double get_t(int slot) const {
    return lds_dyn[(off + slot) * PRT_BLOCK + threadIdx.x];
  }
  __device__ __forceinline__ int get_id(int slot) const {
    return reinterpret_cast<const int*>(lds_dyn + total * PRT_BLOCK)[(off + slot) * PRT_BLOCK + threadIdx.x];
  }
  __device__ __forceinline__ void put(int slot, double v, int i) const {
    lds_dyn[(off + slot) * PRT_BLOCK + threadIdx.x] = v;
    reinterpret_cast<int*>(lds_dyn + total * PRT_BLOCK)[(off + slot) * PRT_BLOCK + threadIdx.x] = i;
  }
};

// an operand of a CSG step: either a register pair (a leaf just evaluated) or a list in LDS
struct Operand {
  int mode, base, len;
};

__device__ __forceinline__ double operand_t(const Operand& o, const LaneLists& l, const Pair& a,
                                            const Pair& b, int i) {
  if (o.mode == OPER_LDS) return l.get_t(o.base + i);
  const double first = (o.mode == OPER_REGA) ? a.t0 : b.t0;
  const double second = (o.mode == OPER_REGA) ? a.t1 : b.t1;
  return i == 0 ? first : second;
}
__device__ __forceinline__ int operand_id(const Operand& o, const LaneLists& l, const Pair& a,
                                          const Pair& b, int i) {
  if (o.mode == OPER_LDS) return l.get_id(o.base + i);
  return (o.mode == OPER_REGA) ? a.prim : b.prim;
}

// array_csg + the id bookkeeping of CSGSurface.intersect (csg.py:13-61, 137-149) for one lane:
// stable two-pointer merge of two ascending lists (ties: left first), +-1 by source-position
// parity (right child flipped for DIFFERENCE), running depth, keep rule with the np.roll
// wrap-around, survivors compacted to the front of the output list in order, rest = +inf.
// `touched` is the node's cull predicate: an untouched ray gets an all-inf list.
// The output may start len(right) slots below the left list (see scene compiler): position
// out+k is written only after left[k - len(right)] has been read.
__device__ __forceinline__ void csg_merge(int op, const Operand& L, const Operand& R, int out_base,
                                          bool touched, const LaneLists& lists, const Pair& ra,
                                          const Pair& rb) {
  const int total = L.len + R.len;
  int i = 0, j = 0, outn = 0;
  int depth = (op == CSG_DIFFERENCE) ? 1 : 0;
  int prev = depth;  // depth of the last merged entry: the totals always cancel
  double a = operand_t(L, lists, ra, rb, 0);
  double b = operand_t(R, lists, ra, rb, 0);
  for (int s = 0; s < total; ++s) {
    const bool take_left = (i < L.len) && ((j >= R.len) || (a <= b));
    const double v = take_left ? a : b;
    const int pos = take_left ? i : j;
    int step = (pos & 1) ? -1 : 1;
    if (op == CSG_DIFFERENCE && !take_left) step = -step;
    depth += step;
    const bool keep = (op == CSG_UNION) ? ((depth != 0) != (prev != 0)) : (depth == 2 || prev == 2);
    prev = depth;
    int id = -1;
    if (keep && touched && v < PRT_INF) id = operand_id(take_left ? L : R, lists, ra, rb, pos);
    if (take_left) {
      ++i;
      a = (i < L.len) ? operand_t(L, lists, ra, rb, i) : PRT_INF;
    } else {
      ++j;
      b = (j < R.len) ? operand_t(R, lists, ra, rb, j) : PRT_INF;
    }
    if (id >= 0) {
      lists.put(out_base + outn, v, id);
      ++outn;
    }
  }
  for (int k = outn; k < total; ++k) lists.put(out_base + k, PRT_INF, -1);
}

// nearest positive entry of an ascending list = its first positive one (_pyrayt.py:380-383)
__device__ __forceinline__ void first_positive(const Operand& o, const LaneLists& lists,
                                               const Pair& ra, const Pair& rb, double& t, int& prim) {
  t = PRT_INF;
  prim = -1;
  for (int k = o.len - 1; k >= 0; --k) {
    const double v = operand_t(o, lists, ra, rb, k);
    if (v > 0 && v < PRT_INF) {
      t = v;
      prim = operand_id(o, lists, ra, rb, k);
    }
  }
}

// The renderers pick differently (renderers.py:79-86, 196-203): argmin over the list masked to
// its positive entries, but the value gathered from the *unmasked* list -- so a list without a
// positive entry offers its first (smallest, negative) entry, and the strict '<' running minimum
// then prefers it.  draw(view="xz") looks away from the parts and sees them only this way.
__device__ __forceinline__ void first_positive_else_first(const Operand& o, const LaneLists& lists,
                                                          const Pair& ra, const Pair& rb, double& t,
                                                          int& prim) {
  t = operand_t(o, lists, ra, rb, 0);
  prim = (t < PRT_INF) ? operand_id(o, lists, ra, rb, 0) : -1;
  for (int k = o.len - 1; k >= 0; --k) {
    const double v = operand_t(o, lists, ra, rb, k);
    if (v > 0 && v < PRT_INF) {
      t = v;
      prim = operand_id(o, lists, ra, rb, k);
    }
  }
}

// The same node, branch-free, for the list shapes the part factories produce (a register
// pair against a register pair or against a short LDS list).  Instead of walking the merged
// order it computes, for every entry independently, what the walk would have seen:
//   with c(i,j) = (L_i <= R_j)            [ties: left first, the stable order]
//   #R before L_i = sum_j !c(i,j)         #L before-or-at R_j = sum_i c(i,j)
//   depth after an entry = own-list signs up to it + other-list signs before it (+1 if DIFF);
//   both lists alternate +,-,... (the right one negated for DIFFERENCE), so a partial sum is
//   the first sign if the count is odd and 0 if even;  depth before it = depth - own sign
//   (for the first merged entry that is the wrap-around value of np.roll).
// Keep rule as in csg_merge.  Survivors are scattered to LDS at their compacted position
// (= kept entries of lower merged rank), or -- for the root node of a component -- reduced
// directly to the nearest positive survivor, ties to the lower merged rank, without
// materialising the list (the tracer only ever looks at that entry, _pyrayt.py:380-386).
//
// The cull box (csg.py:126-128) is evaluated lazily where it is implied: for an INTERSECT or
// DIFFERENCE node over UNION-free subtrees the node's box contains the node's solid (it is the
// intersection of the children's boxes, resp. the left child's box, and a surface's box bounds
// the surface), so a ray with two survivors a robust distance apart runs through the box for
// at least that chord and the strict enter < leave test of primitives.py:578 holds; a ray with
// no survivor gets the all-inf list either way.  Only rays whose survivors are (nearly)
// coincident -- tangent rays, a Plane's double hit (t,t) -- take the exact six-division test.
// UNION nodes always take it (their upstream box can be smaller than the solid).

// The cull box of a node whose box is implied by its survivors (see above): `any` -- the node has survivors, the
// smallest `lo` and the largest `hi`.  Returns upstream's "touched" (csg.py:126-128) for this ray.
__device__ __forceinline__ bool implied_touch(const double* __restrict__ aabb, const Ray8& ray, bool any, double lo,
                                              double hi) {
  bool touched = true;
  // (a ray that is not well formed -- a short direction may send a leaf into one of upstream's degenerate
  // branches without being parallel to anything, and what survives then need not lie in the solid; with
  // w other than 1 / 0 the object-space ray is not the image of the world ray the box was tested
  // against -- gets the exact test, like a thin chord.  Found by the short-direction and odd-w fuzz families.)
  const bool robust = any && well_formed(ray) && lo > -PRT_INF && (hi - lo) > 1e-6 * ((1.0 + fabs(lo)) + fabs(hi));
  count_paths(ray.paths, 2, any);
  count_paths(ray.paths, 3, any && !robust);
  if (any && !robust) touched = box_touched(aabb, ray);
  // The chord argument covers the axes the ray really moves along: with |d| >= 1e-4 a slab crossing
  // is off by at most ~1e-12 / |d| <= 1e-8 (rounding, and the 1e-12 by which the compiler lets the
  // upstream box fall short of the solid's bounds), far below the robust margin.  An axis with
  // |d| <= 1e-8 has no crossing at all: there upstream's test is the bare comparison lo <= o <= hi
  // (primitives.py:531-565), which a ray running along a face one ulp outside the box fails however
  // long its chord through the (object-space) solid is -- found by the adversarial fixtures
  // (tests/scenes.py adv_lens).  In between (1e-8 < |d| < 1e-4) the exact test decides.
  const double ax = fabs(ray.dx), ay = fabs(ray.dy), az = fabs(ray.dz);
  if (__ballot(ax < 1e-4 || ay < 1e-4 || az < 1e-4) != 0ull) {
    const bool px = ax <= 1e-8, py = ay <= 1e-8, pz = az <= 1e-8;
    const bool outside = (px && !(ray.ox >= aabb[0] && ray.ox <= aabb[1])) ||
                         (py && !(ray.oy >= aabb[2] && ray.oy <= aabb[3])) ||
                         (pz && !(ray.oz >= aabb[4] && ray.oz <= aabb[5]));
    // (no crossing on any axis -- a zero-direction ray, e.g. an absorbed one carried along: upstream's
    // box test then returns (-inf, +inf) or (+inf, +inf), never a finite entry, and csg.py:126-128
    // drops the ray wherever it sits; a paraboloid child would still report the finite -c / 1 of its
    // linear branch, primitives.py:361 -- found by fuzz seed 8061 of a 12 000-seed run)
    touched = touched && !outside && !(px && py && pz);
    // A grazing axis (1e-8 < |d_k| < 1e-4): upstream's slab parameters (lo_k - o_k) / d_k, (hi_k - o_k) / d_k are
    // huge numbers with absolute errors to match, and the chord argument does not cover them: such a ray takes the
    // exact test.  (Only near a face of the box would it have to -- measured in round 4: config 3 -1.5 %, config 2
    // +0.8 % with 20 B of scratch per lane; not adopted, tools/experiments/.)
    const bool grazing = (!px && ax < 1e-4) || (!py && ay < 1e-4) || (!pz && az < 1e-4);
    if (__ballot(grazing && any && robust) != 0ull) {
      const bool exact = grazing && any && robust;
      if (__ballot(exact) != 0ull) {
        count_paths(ray.paths, 3, exact);
        if (exact) touched = touched && box_touched(aabb, ray);
      }
    }
  }
  return touched;
}

// keep flags of one node (the part of csg_node below that needs no list storage): c(i,j), the
// parities, the keep rule per operation and the (lazily evaluated) cull box.  Neither list has to
// be sorted or compacted for this: every quantity is a count of entries of the *other* list that
// sort before an entry, and a dropped entry left in place as +inf sorts before nothing finite.
template <int ML, int MR>
__device__ __forceinline__ void csg_keep(int op, const double (&lv)[ML], const double (&rv)[MR],
                                         const double* __restrict__ aabb, const Ray8& ray,
                                         bool box_implied, bool (&keep_l)[ML], bool (&keep_r)[MR],
                                         bool (&c)[ML][MR]) {
  // c(i,j) = L_i sorts before R_j.  Everything the keep rule needs is a parity: with p_i = (number
  // of R entries before L_i) mod 2 and q_j = (number of L entries before-or-at R_j) mod 2, working the
  // +-1 depths of array_csg through (own list alternates from +1, or from -1 for the right list
  // of a DIFFERENCE, whose depth also starts at 1; "depth before" of the first merged entry is the
  // np.roll wrap value, which the same algebra yields) gives
  //     UNION       keep L_i = !p_i   keep R_j = !q_j      (depth != 0  xor  before != 0)
  //     INTERSECT   keep L_i =  p_i   keep R_j =  q_j      (depth == 2  or   before == 2)
  //     DIFFERENCE  keep L_i = !p_i   keep R_j =  q_j
  // -- lane masks and scalar logic only; per-lane integers are needed only to place survivors.
  bool p[ML], q[MR];
#pragma unroll
  for (int i = 0; i < ML; ++i) p[i] = false;
#pragma unroll
  for (int j = 0; j < MR; ++j) q[j] = false;
#pragma unroll
  for (int i = 0; i < ML; ++i) {
#pragma unroll
    for (int j = 0; j < MR; ++j) {
      c[i][j] = lv[i] <= rv[j];
      q[j] = q[j] != c[i][j];
      p[i] = p[i] == c[i][j];  // flips when R_j is before L_i, i.e. when !c
    }
  }
  const bool flip_l = op != CSG_INTERSECT, flip_r = op == CSG_UNION;
#pragma unroll
  for (int i = 0; i < ML; ++i) keep_l[i] = (p[i] != flip_l) && lv[i] < PRT_INF;
#pragma unroll
  for (int j = 0; j < MR; ++j) keep_r[j] = (q[j] != flip_r) && rv[j] < PRT_INF;
  bool touched = true;
  if (!box_implied) {  // wave-uniform
    touched = box_touched(aabb, ray);
  } else {
    // list entries are finite or +inf, never NaN (leaves map NaN to +inf): v_min / v_max are exact
    double lo = PRT_INF, hi = -PRT_INF;
#pragma unroll
    for (int i = 0; i < ML; ++i) {
      lo = fmin(lo, keep_l[i] ? lv[i] : PRT_INF);
      hi = fmax(hi, keep_l[i] ? lv[i] : -PRT_INF);
    }
#pragma unroll
    for (int j = 0; j < MR; ++j) {
      lo = fmin(lo, keep_r[j] ? rv[j] : PRT_INF);
      hi = fmax(hi, keep_r[j] ? rv[j] : -PRT_INF);
    }
    touched = implied_touch(aabb, ray, hi >= lo, lo, hi);  // (hi >= lo: false when nothing survived: lo = +inf, hi = -inf)
  }
#pragma unroll
  for (int i = 0; i < ML; ++i) keep_l[i] = keep_l[i] && touched;
#pragma unroll
  for (int j = 0; j < MR; ++j) keep_r[j] = keep_r[j] && touched;
}

// nearest positive survivor of a node; ties go to the lower merged rank, and among equal values the
// merged order is "left entries by index, then right entries by index" (stable, left first) -- the
// order of this scan, so a strict '<' keeps the right one
template <int ML, int MR>
__device__ __forceinline__ void csg_root_pick(const double (&lv)[ML], const int (&lid)[ML],
                                              const bool (&keep_l)[ML], const double (&rv)[MR],
                                              const int (&rid)[MR], const bool (&keep_r)[MR],
                                              double& node_t, int& node_prim) {
  double best = PRT_INF;
  int best_id = -1;
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    const bool better = keep_l[i] && lv[i] > 0 && lv[i] < best;
    best = better ? lv[i] : best;
    best_id = better ? lid[i] : best_id;
  }
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    const bool better = keep_r[j] && rv[j] > 0 && rv[j] < best;
    best = better ? rv[j] : best;
    best_id = better ? rid[j] : best_id;
  }
  node_t = best;
  node_prim = best_id;
}

template <int ML, int MR, bool LREG, bool RREG>
__device__ __forceinline__ void csg_node(int op, int l_base, int r_base, int out_base,
                                         const double* __restrict__ aabb, const Ray8& ray,
                                         bool box_implied, bool is_root, const LaneLists& lists,
                                         const Pair& ra, const Pair& rb, double& node_t,
                                         int& node_prim) {
  double lv[ML], rv[MR];
  int lid[ML], rid[MR];
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    if (LREG) { lv[i] = (i == 0) ? ra.t0 : ra.t1; lid[i] = ra.prim; }
    else { lv[i] = lists.get_t(l_base + i); lid[i] = lists.get_id(l_base + i); }
  }
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    if (RREG) { rv[j] = (j == 0) ? rb.t0 : rb.t1; rid[j] = rb.prim; }
    else { rv[j] = lists.get_t(r_base + j); rid[j] = lists.get_id(r_base + j); }
  }
  bool c[ML][MR];
  bool keep_l[ML], keep_r[MR];
  csg_keep<ML, MR>(op, lv, rv, aabb, ray, box_implied, keep_l, keep_r, c);
  if (is_root) {
    csg_root_pick<ML, MR>(lv, lid, keep_l, rv, rid, keep_r, node_t, node_prim);
    return;
  }
  int kept = 0;
#pragma unroll
  for (int i = 0; i < ML; ++i) {
    int pos = 0;
#pragma unroll
    for (int k = 0; k < i; ++k) pos += keep_l[k] ? 1 : 0;
#pragma unroll
    for (int j = 0; j < MR; ++j) pos += (!c[i][j] && keep_r[j]) ? 1 : 0;
    if (keep_l[i]) lists.put(out_base + pos, lv[i], lid[i]);
    kept += keep_l[i] ? 1 : 0;
  }
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    int pos = 0;
#pragma unroll
    for (int k = 0; k < j; ++k) pos += keep_r[k] ? 1 : 0;
#pragma unroll
    for (int i = 0; i < ML; ++i) pos += (c[i][j] && keep_l[i]) ? 1 : 0;
    if (keep_r[j]) lists.put(out_base + pos, rv[j], rid[j]);
    kept += keep_r[j] ? 1 : 0;
  }
#pragma unroll
  for (int k = 0; k < ML + MR; ++k)
    if (k >= kept) lists.put(out_base + k, PRT_INF, -1);
}

// One CSG step of the program: pick the specialised node for the common shapes, fall back
// to the serial merge for anything else (two LDS operands, very long chains).
// `in->pad[0]` marks the root node of a component in the trace program.
__device__ __forceinline__ void csg_step(const DevInstr* __restrict__ in, const Ray8& ray,
                                         const LaneLists& lists, const Pair& ra, const Pair& rb,
                                         bool& produced_root, double& node_t, int& node_prim) {
  const int op = in->a0, lmode = in->a1, lbase = in->a2, ml = in->a3;
  const int rmode = in->a4, rbase = in->a5, mr = in->a6, obase = in->a7;
  const bool is_root = in->pad[0] != 0;
  const bool implied = in->pad[1] != 0;
  produced_root = is_root;
  if (lmode == OPER_REGA && rmode == OPER_REGB) {
    csg_node<2, 2, true, true>(op, 0, 0, obase, in->data, ray, implied, is_root, lists, ra, rb, node_t, node_prim);
    return;
  }
  if (lmode == OPER_LDS && rmode == OPER_REGB && ml == 4) {
    csg_node<4, 2, false, true>(op, lbase, 0, obase, in->data, ray, implied, is_root, lists, ra, rb, node_t, node_prim);
    return;
  }
  if (lmode == OPER_REGA && rmode == OPER_LDS && mr == 4) {
    csg_node<2, 4, true, false>(op, 0, rbase, obase, in->data, ray, implied, is_root, lists, ra, rb, node_t, node_prim);
    return;
  }
  const Operand L = {lmode, lbase, ml};
  const Operand R = {rmode, rbase, mr};
  csg_merge(op, L, R, obase, box_touched(in->data, ray), lists, ra, rb);
  if (is_root) {
    const Operand o = {OPER_LDS, obase, ml + mr};
    first_positive(o, lists, ra, rb, node_t, node_prim);
  }
}

// Component cull (I_BOX step, trace programs of scenes with several components): can this ray
// reach the component's box at a parameter in (0, best_t]?  `box` bounds every leaf surface of the
// component in world space, padded on the host by 1e-3 of its diagonal; every positive finite entry
// of the component's hit list is a parameter at which the ray is (numerically) on one of those
// surfaces, so "no" means the component cannot change the running nearest hit: a tie with best_t
// keeps the earlier component (_pyrayt.py:384), a larger value loses.  The test only ever errs
// towards "yes": reciprocal by v_rcp_f64 with a 1e-6 relative allowance, NaN compares false.
// LINE (render programs): does the whole LINE of the ray meet the box?  The renderers' rule lets a
// component with no positive hit offer its first entry, a parameter behind the camera (renderers.py:79-83),
// so only a component whose list holds no finite entry at all may be skipped: one whose box the line misses.
template <bool LINE = false>
__device__ __forceinline__ bool may_reach(const double* __restrict__ box, const Ray8& r, double best_t) {
  double t_in = LINE ? -PRT_INF : 0.0, t_out = LINE ? PRT_INF : best_t;
  bool never = false;
  const double o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
  // (asked only for well-formed rays: with a direction so short that a leaf of the component may take one
  // of upstream's degenerate branches without being parallel to anything, the hit such a branch reports
  // need not lie on the surface, and the box says nothing about it -- see well_formed)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double inv = __builtin_amdgcn_rcp(d[k]);
    const double a = (box[2 * k] - o[k]) * inv, b = (box[2 * k + 1] - o[k]) * inv;
    const bool parallel = fabs(d[k]) < 1e-300;  // the slab either contains the whole ray or none of it
    never = never || (parallel && (o[k] < box[2 * k] || o[k] > box[2 * k + 1]));
    t_in = parallel ? t_in : fmax(t_in, fmin(a, b));
    t_out = parallel ? t_out : fmin(t_out, fmax(a, b));
  }
  if (LINE && (t_in == -PRT_INF || t_out == PRT_INF)) return !never;  // parallel to every slab: inside all of them or not
  const double slack = 1e-6 * (fabs(t_in) + fabs(t_out));
  return !never && !(t_in - slack > t_out + slack);
}

// The same question asked more cheaply and more coarsely, in front of may_reach: does the axis-aligned box of the
// SEGMENT the ray covers over (0, best_t] meet the component's box at all?  Along axis k the segment runs from o_k to
// p_k = o_k + best_t d_k (monotonically), a hit at a parameter in (0, best_t] lies between the two, and every such
// hit is a point of the component's (padded) box: no overlap on some axis means no such hit.  Five instructions per
// axis and no division against may_reach's twenty with a reciprocal; it only ever answers "no" when may_reach would
// (the segment's box contains the segment), so asking it first changes no result.  In an optical train it settles
// most steps: parts behind the ray, and parts beyond its nearest hit so far, differ along the axis alone.
// (best_t = inf: p_k is +-inf, or NaN for d_k = 0, which v_min / v_max drop: the "segment" is then the half line,
// resp. the point o_k.)
__device__ __forceinline__ bool segment_meets(const double* __restrict__ box, const Ray8& r, double best_t) {
  const double o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
  bool apart = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double p = fma(best_t, d[k], o[k]);
    apart = apart || fmax(o[k], p) < box[2 * k] || fmin(o[k], p) > box[2 * k + 1];
  }
  return !apart;
}

// ---- chain steps (I_CHAIN) ----------------------------------------------------------------------
// The component's candidate for the running nearest hit, computed entirely in registers.  The first
// node's survivors are not compacted: a dropped entry stays in place as +inf (see csg_keep), and the
// scan order "left operand's entries, then right operand's" is the stable merged order among equal
// values at every level, so ties resolve as in csg_node.  When no lane of the wave holds a positive
// survivor after the first node, the third leaf is not evaluated (an INTERSECT / DIFFERENCE result
// lies inside its left operand: the same argument as the interpreter's right-leaf skip).
// Third leaf of a lens chain, a CYLINDER (every lens factory cuts its two spherical faces to the aperture with one,
// components.py:73-198): can it change the chord [lo, hi] the first two leaves left?  Not if the ray is inside the
// cylinder's solid over a slightly longer stretch: the solid is convex, so two points inside it -- the ray at
// lo - m and at hi + m, m = 1e-6 (1 + |lo| + |hi|) -- put every point between them inside, the cylinder's own interval
// [enter, exit] then contains [lo - m, hi + m], and `enter >= lo` / `exit < hi` (chain_candidate) are both false
// whatever the last bits of enter and exit are: upstream's roots carry absolute errors of ~1e-16 of the magnitudes
// involved, ten orders below m.  The points are tested in the leaf's own object space, on the ray exactly as
// surface_pair hands it to the primitive, against the solid shrunk by 1e-9 of those magnitudes (the test's own
// rounding); a ray on one of upstream's degenerate branches (|dx^2 + dy^2| <= 1e-8: operations.py:43, SURVEY Q5 -- such
// a ray MISSES the cylinder it runs through --, or |dz| <= 1e-8) is not cleared, and the thresholds are evaluated on
// the very numbers primitive_pair evaluates them on.  About 70 instructions against the leaf's 190.
__device__ __forceinline__ bool chord_inside_cylinder(const double* __restrict__ q, const double* __restrict__ m,
                                                      const Ray8& r, double lo, double hi) {
  const double ox = row_dot(m, 0, r.ox, r.oy, r.oz, r.ow), oy = row_dot(m, 1, r.ox, r.oy, r.oz, r.ow),
               oz = row_dot(m, 2, r.ox, r.oy, r.oz, r.ow);
  const double dx = row_dot(m, 0, r.dx, r.dy, r.dz, r.dw), dy = row_dot(m, 1, r.dx, r.dy, r.dz, r.dw),
               dz = row_dot(m, 2, r.dx, r.dy, r.dz, r.dw);
  const double a = dx * dx + dy * dy;  // (as primitive_pair computes it)
  const double margin = 1e-6 * ((1.0 + fabs(lo)) + fabs(hi));
  const double rr = q[0] * q[0];
  // |x| <= 2 |ox| + |x'| for x' = ox + t dx: the magnitudes that went into a point bound its rounding
  const double radial_room = rr - 1e-9 * (8.0 * (ox * ox + oy * oy) + rr);
  const double axial_slack = 1e-9 * ((2.0 * fabs(oz) + fabs(q[1])) + fabs(q[2]));
  bool inside = !near0(a) && !near0(dz);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double t = k == 0 ? lo - margin : hi + margin;
    const double x = fma(t, dx, ox), y = fma(t, dy, oy), z = fma(t, dz, oz);
    const double rho2 = x * x + y * y;
    const double slack = axial_slack + 1e-9 * fabs(z);
    inside = inside && rho2 * (1.0 + 2e-9) <= radial_room && z >= q[1] + slack && z <= q[2] - slack;  // (NaN: not inside)
  }
  return inside;
}

// what an interval chain offers the running nearest hit: its chord's entry if that lies ahead, else its exit
__device__ __forceinline__ void offer_chord(const Ray8& ray, bool some, double lo, double hi, int id_lo, int id_hi,
                                            double& best_t, int& best_prim) {
  const bool first = some && lo > 0;  // (lo < hi: finite)
  const bool second = some && hi > 0 && hi < PRT_INF;
  const double t = first ? lo : (second ? hi : PRT_INF);
  const int prim = first ? id_lo : (second ? id_hi : -1);
  if (beats(ray, t, prim, best_t, best_prim)) {
    best_t = t;
    best_prim = prim;
  }
}

template <int T0, int T1, int T2>
__device__ __forceinline__ void chain_candidate(const DevChain* __restrict__ ch, const Ray8& ray,
                                                double& best_t, int& best_prim) {
  double lv[2], rv[2], r2[2] = {PRT_INF, PRT_INF};
  const int id0 = ch->prim[0], id1 = ch->prim[1];
  bool have_third = false;
  // ---- the aperture cylinder as FIRST leaf (thick_lens: (stock & face) & face, components.py:73-127) ----------------
  // An interval chain is an intersection of intervals, so the two faces can be intersected first: [lo, hi] of leaves 1
  // and 2 alone, left before right as the chain has them.  If the cylinder's own interval contains that chord with a
  // margin (chord_inside_cylinder), none of its entries is the later enter or the earlier exit at either node, nor
  // tied with one, and the chain's result is this chord with these ids; the two cull boxes are functions of the ray
  // alone (implied_touch only chooses how to evaluate them), the final chord lies inside both nodes' solids.  A wave
  // with a lane that is not cleared -- a chord near the rim, a tie between the faces, a degenerate branch -- evaluates
  // the cylinder and runs the chain in its own order with the pairs it already has.  No wanted chord at all (no lane's
  // faces overlap ahead of the ray): the chain, a subset of that overlap, has nothing to offer either.
  if (T0 == PRIM_CYLINDER && T2 >= 0 && ch->intervals != 0 && ch->clearance != 0) {  // (uniform)
    surface_pair(T1, ch->leaf[1], ch->leaf[1] + 6, ray, rv[0], rv[1]);
    surface_pair(T2 < 0 ? 0 : T2, ch->leaf[2], ch->leaf[2] + 6, ray, r2[0], r2[1]);
    have_third = true;
    const int id2 = ch->prim[2];
    const bool b_in = r2[0] >= rv[0], b_out = r2[1] < rv[1];
    const double lo = b_in ? r2[0] : rv[0], hi = b_out ? r2[1] : rv[1];
    const bool tie = lo == hi && hi < PRT_INF;
    bool some = lo < hi;
    const bool wanted = (some && hi > 0) || tie;
    if (__ballot(wanted) == 0ull) return;
    if (__ballot(wanted && (tie || !chord_inside_cylinder(ch->leaf[0], ch->leaf[0] + 6, ray, lo, hi))) == 0ull) {
      some = some && implied_touch(ch->box1, ray, some, lo, hi);
      some = some && implied_touch(ch->box2, ray, some, lo, hi);
      offer_chord(ray, some, lo, hi, b_in ? id2 : id1, b_out ? id2 : id1, best_t, best_prim);
      return;
    }
    surface_pair(T0, ch->leaf[0], ch->leaf[0] + 6, ray, lv[0], lv[1]);
  } else {
    surface_pair(T0, ch->leaf[0], ch->leaf[0] + 6, ray, lv[0], lv[1]);
    surface_pair(T1, ch->leaf[1], ch->leaf[1] + 6, ray, rv[0], rv[1]);
  }
  // ---- the interval form ------------------------------------------------------------------------------------
  // Every leaf reports ONE interval [enter, exit] of the ray inside its solid.  array_csg's INTERSECT of two
  // such lists keeps, for values in general position, exactly the later of the two enters and the earlier of the
  // two exits when the former comes first, and nothing otherwise -- the intersection of the intervals -- and a
  // chain of INTERSECT nodes nests that.  Ties between entries of different leaves resolve by the stable merge
  // (csg.py:13-61: left operand first): of two equal enters the RIGHT one is the depth-2 entry that is kept, of
  // two equal exits the LEFT one (the parities of csg_keep say the same), which is what `>=` / `<` below select.
  // An enter that equals an exit (lo == hi: a chord of zero length, which the merge order may or may not keep) is
  // not decided here: a wave holding such a lane takes the general path below, with the pairs it already has.
  // The cull boxes go through implied_touch exactly as in csg_keep.
  if (ch->intervals != 0) {  // (uniform)
    const bool b_in = rv[0] >= lv[0], b_out = rv[1] < lv[1];
    double lo = b_in ? rv[0] : lv[0], hi = b_out ? rv[1] : lv[1];
    int id_lo = b_in ? id1 : id0, id_hi = b_out ? id1 : id0;
    bool undecided = lo == hi && hi < PRT_INF;
    bool some = lo < hi;
    some = some && implied_touch(ch->box1, ray, some, lo, hi);
    if (T2 >= 0) {
      // (a chord that ends behind the ray leaves nothing positive for the third leaf to cut: see below)
      const bool wanted = (some && hi > 0) || undecided;
      if (__ballot(wanted) == 0ull) return;
      // (a cylinder that provably leaves every wanted chord of the wave as it is -- the beam runs well inside the
      // aperture -- is not evaluated: chord_inside_cylinder)
      bool cut = true;
      if (T2 == PRIM_CYLINDER && ch->clearance != 0)  // (uniform)
        cut = __ballot(wanted && (undecided || !chord_inside_cylinder(ch->leaf[2], ch->leaf[2] + 6, ray, lo, hi))) != 0ull;
      if (cut) {
        if (!have_third) surface_pair(T2 < 0 ? 0 : T2, ch->leaf[2], ch->leaf[2] + 6, ray, r2[0], r2[1]);
        have_third = true;
        const int id2 = ch->prim[2];
        const bool c_in = r2[0] >= lo, c_out = r2[1] < hi;
        lo = c_in ? r2[0] : lo; hi = c_out ? r2[1] : hi;
        id_lo = c_in ? id2 : id_lo; id_hi = c_out ? id2 : id_hi;
        undecided = undecided || (some && lo == hi && hi < PRT_INF);
      }
      some = some && lo < hi;
      some = some && implied_touch(ch->box2, ray, some, lo, hi);
    }
    if (__ballot(undecided) == 0ull) {
      offer_chord(ray, some, lo, hi, id_lo, id_hi, best_t, best_prim);
      return;
    }
  }
  bool keep_l[2], keep_r[2], c1[2][2];
  const int op1 = ch->op1, op2 = ch->op2;
  const bool implied1 = ch->implied1 != 0, implied2 = ch->implied2 != 0;
  csg_keep<2, 2>(op1, lv, rv, ch->box1, ray, implied1, keep_l, keep_r, c1);
  double t;
  int prim;
  if (T2 < 0) {
    const int lid[2] = {id0, id0}, rid[2] = {id1, id1};
    csg_root_pick<2, 2>(lv, lid, keep_l, rv, rid, keep_r, t, prim);
  } else {
    const double l4[4] = {keep_l[0] ? lv[0] : PRT_INF, keep_l[1] ? lv[1] : PRT_INF,
                          keep_r[0] ? rv[0] : PRT_INF, keep_r[1] ? rv[1] : PRT_INF};
    // a positive survivor, or an odd number of crossings behind the ray: it is inside the first node's
    // solid at 0+ (see the interpreter's right-leaf skip)
    bool positive = false;
    int behind = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      positive = positive || (l4[k] > 0 && l4[k] < PRT_INF);
      behind += l4[k] <= 0 ? 1 : 0;
    }
    positive = positive || (behind & 1) || !well_formed(ray);
    if (op2 != CSG_UNION && __ballot(positive) == 0ull) return;
    if (!have_third) {  // (the interval form may have evaluated it already)
      surface_pair(T2 < 0 ? 0 : T2, ch->leaf[2], ch->leaf[2] + 6, ray, r2[0], r2[1]);
    }
    bool keep4[4], keep2[2], c2[4][2];
    csg_keep<4, 2>(op2, l4, r2, ch->box2, ray, implied2, keep4, keep2, c2);
    const int id2 = ch->prim[2];
    const int lid[4] = {id0, id0, id1, id1}, rid[2] = {id2, id2};
    csg_root_pick<4, 2>(l4, lid, keep4, r2, rid, keep2, t, prim);
  }
  if (beats(ray, t, prim, best_t, best_prim)) {
    best_t = t;
    best_prim = prim;
  }
}

__device__ __forceinline__ void chain_step(const DevChain* __restrict__ ch, int shape, const Ray8& ray,
                                           double& best_t, int& best_prim) {
  switch (shape) {  // wave-uniform
    case CHAIN_SSC: chain_candidate<PRIM_SPHERE, PRIM_SPHERE, PRIM_CYLINDER>(ch, ray, best_t, best_prim); break;
    case CHAIN_SSQ: chain_candidate<PRIM_SPHERE, PRIM_SPHERE, PRIM_CUBE>(ch, ray, best_t, best_prim); break;
    case CHAIN_CSS: chain_candidate<PRIM_CYLINDER, PRIM_SPHERE, PRIM_SPHERE>(ch, ray, best_t, best_prim); break;
    case CHAIN_QSS: chain_candidate<PRIM_CUBE, PRIM_SPHERE, PRIM_SPHERE>(ch, ray, best_t, best_prim); break;
    case CHAIN_QQQ: chain_candidate<PRIM_CUBE, PRIM_CUBE, PRIM_CUBE>(ch, ray, best_t, best_prim); break;
    case CHAIN_SC: chain_candidate<PRIM_SPHERE, PRIM_CYLINDER, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_SQ: chain_candidate<PRIM_SPHERE, PRIM_CUBE, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_CS: chain_candidate<PRIM_CYLINDER, PRIM_SPHERE, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_QS: chain_candidate<PRIM_CUBE, PRIM_SPHERE, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_CB: chain_candidate<PRIM_CYLINDER, PRIM_PARABOLOID, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_QB: chain_candidate<PRIM_CUBE, PRIM_PARABOLOID, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_PC: chain_candidate<PRIM_PLANE, PRIM_CYLINDER, -1>(ch, ray, best_t, best_prim); break;
    case CHAIN_PQ: chain_candidate<PRIM_PLANE, PRIM_CUBE, -1>(ch, ray, best_t, best_prim); break;
    default: chain_candidate<PRIM_PARABOLOID, PRIM_CYLINDER, -1>(ch, ray, best_t, best_prim); break;
  }
}

// One program step for one ray (state: the two register pairs and the running nearest hit).
// RENDER selects the renderers' root rule at compile time, so the tracer's kernels carry none of it.
template <bool RENDER = false>
__device__ __forceinline__ void run_step(const DevInstr* in, const Ray8& ray, const LaneLists& lists,
                                         Pair& ra, Pair& rb, double& best_t, int& best_prim) {
  const int kind = in->kind;
  if (kind == I_LEAF) {
    const int p = in->a0;
    double t0, t1;
    if (in->pad[0] == 1) {
      // right leaf of an INTERSECT / DIFFERENCE node: not worth evaluating for this wave when no
      // lane's left operand (a3 mode, a4 base, a5 length) holds a positive entry
      // "positive": a finite entry beyond 0 -- or the ray is inside the left solid at 0+, which the
      // boundary crossings behind it tell by their parity (they alternate enter / exit): a slab or
      // linear branch that reports (-inf, +inf) or (-5, +inf) has no finite positive entry and is all
      // around the ray nevertheless (found by the short-direction fuzz family)
      bool has_positive;
      if (in->a3 == OPER_LDS) {
        has_positive = false;
        int behind = 0;
        for (int k = 0; k < in->a5; ++k) {
          const double v = lists.get_t(in->a4 + k);
          has_positive = has_positive || (v > 0 && v < PRT_INF);
          behind += v <= 0 ? 1 : 0;
        }
        has_positive = has_positive || (behind & 1);
      } else {
        has_positive = (ra.t0 > 0 && ra.t0 < PRT_INF) || (ra.t1 > 0 && ra.t1 < PRT_INF) ||
                       ((ra.t0 <= 0) != (ra.t1 <= 0));
      }
      has_positive = has_positive || !well_formed(ray);
      if (__ballot(has_positive) == 0ull) {
        rb.t0 = PRT_INF; rb.t1 = PRT_INF; rb.prim = p;
        return;
      }
    }
    surface_pair(in->type, in->data, in->data + 6, ray, t0, t1);
    const int dst = in->a1;
    if (in->pad[0] == 2) {
      // the leaf is a whole component (a bare surface): its sorted pair reduces right here to the
      // component's candidate -- first positive entry, or for the renderers the first entry when
      // none is positive -- and meets the running minimum; no list, no separate root step
      const bool pos0 = t0 > 0 && t0 < PRT_INF, pos1 = t1 > 0 && t1 < PRT_INF;
      double t = pos0 ? t0 : (pos1 ? t1 : PRT_INF);
      bool hit = pos0 || pos1;
      if (RENDER && !hit) { t = t0; hit = t0 < PRT_INF; }
      if (hit && beats(ray, t, p, best_t, best_prim)) {
        best_t = t;
        best_prim = p;
      }
    } else if (dst == OPER_REGA) {
      ra.t0 = t0; ra.t1 = t1; ra.prim = p;
    } else if (dst == OPER_REGB) {
      rb.t0 = t0; rb.t1 = t1; rb.prim = p;
    } else {
      lists.put(in->a2, t0, p);
      lists.put(in->a2 + 1, t1, p);
    }
  } else if (kind == I_CSG) {
    bool is_root;
    double t = PRT_INF;
    int prim = -1;
    csg_step(in, ray, lists, ra, rb, is_root, t, prim);
    if (is_root && beats(ray, t, prim, best_t, best_prim)) {
      best_t = t;
      best_prim = prim;
    }
  } else {  // I_ROOT: reduce a component's finished list to its candidate (render programs of CSG components)
    const Operand o = {in->a0, in->a1, in->a2};
    double t;
    int prim;
    if (RENDER)
      first_positive_else_first(o, lists, ra, rb, t, prim);
    else
      first_positive(o, lists, ra, rb, t, prim);
    if (beats(ray, t, prim, best_t, best_prim)) {
      best_t = t;
      best_prim = prim;
    }
  }
}

// Run the scene program for R rays held by one lane: nearest positive hit over all components
// with the strict '<' running minimum of _pyrayt.py:384-386.  `code` is wave-uniform; each
// step record is fetched once (one batch of scalar loads) and applied to the lane's R rays,
// whose arithmetic is independent and interleaves.
// CULL: the program may contain I_BOX steps (compiled out of the kernels that never see one).
// ANY_W: the rays may carry homogeneous w components other than 1 / 0 (the object-space ray is then
// M^-1 (o, w_o), M^-1 (d, w_d), not the image of the world ray: world_objects.py:360-383 multiplies the
// translation by w): well_formed() then looks at them (the compact kernels know they are 1 / 0).
template <int R, bool RENDER = false, bool CULL = true, bool ANY_W = true>
__device__ __forceinline__ void nearest_hit_n(const DevInstr* __restrict__ code, int n_instr,
                                              const Ray8 (&ray)[R], int slots, double (&best_t)[R],
                                              int (&best_prim)[R]) {
  Pair ra[R], rb[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    best_t[k] = PRT_INF;
    best_prim[k] = -1;
    ra[k] = Pair{PRT_INF, PRT_INF, -1};
    rb[k] = Pair{PRT_INF, PRT_INF, -1};
    if (ray[k].paths != nullptr) count_paths(ray[k].paths, 1, !well_formed(ray[k]));
  }
  for (int pc = 0; pc < n_instr; ++pc) {
    if (!RENDER && R == 1) {  // trace programs: a whole component may be one chain record
      const int kind = code[pc].kind, shape = code[pc].a0;
      if (kind == I_CHAIN) {
        chain_step(reinterpret_cast<const DevChain*>(code + pc), shape, ray[0], best_t[0], best_prim[0]);
        pc += CHAIN_SLOTS - 1;
        continue;
      }
    }
    const DevInstr& step = code[pc];  // fields are fetched as they are used (a whole record is 48 SGPRs at once)
    if (CULL && RENDER && step.kind == I_BOX) {  // render programs: a component whose box the line of sight misses
      bool wanted = false;
#pragma unroll
      for (int k = 0; k < R; ++k) wanted = wanted || !well_formed(ray[k]) || may_reach<true>(step.data, ray[k], PRT_INF);
      if (__ballot(wanted) == 0ull) pc += step.a0;
      continue;
    }
    if (CULL && !RENDER && step.kind == I_BOX) {
      if (step.a1 != BOX_TEST) {  // (uniform) the frame of a program that is stored in both directions
        bool jump = true;
        if (step.a1 == BOX_PICK) {
          const int axis = step.a2 & 3;
          const double along = axis == 0 ? ray[0].dx : (axis == 1 ? ray[0].dy : ray[0].dz);
          const int down = __popcll(__ballot(along < 0.0)), up = __popcll(__ballot(along > 0.0));
          jump = (step.a2 & 4) ? down >= up : up >= down;  // most rays run along the program: skip the mirror image
        }
        if (jump) pc += step.a0;
        continue;
      }
      bool wanted = false;
      // (the coarse test first: a wave none of whose segments comes near the box is done with five instructions
      // per axis; only a wave that does come near pays for the exact one)
#pragma unroll
      for (int k = 0; k < R; ++k)
        wanted = wanted || !well_formed(ray[k]) || segment_meets(step.data, ray[k], best_t[k]);
      if (__ballot(wanted) == 0ull) { pc += step.a0; continue; }
      wanted = false;
#pragma unroll
      for (int k = 0; k < R; ++k)
        wanted = wanted || !well_formed(ray[k]) || may_reach(step.data, ray[k], best_t[k]);
      if (__ballot(wanted) == 0ull) pc += step.a0;  // no lane of the wave needs this component
      continue;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const LaneLists lists = {R * slots, k * slots};
      run_step<RENDER>(&step, ray[k], lists, ra[k], rb[k], best_t[k], best_prim[k]);
    }
  }
}

// One component's candidate with a *per-lane* program counter: steps [first, last) of `code`, which may
// differ from lane to lane (the k-lanes-per-ray kernels give every lane of a ray's group its own
// components).  Step records are then per-lane data: they arrive through vector loads, or from an LDS
// copy of the program, not through the scalar cache.  The wave-level shortcuts inside the steps
// (ballots over the active lanes) stay valid: each only ever skips work when *no* active lane needs it.
__device__ __forceinline__ void component_candidate(const DevInstr* code, int first, int last, const Ray8& ray,
                                                    const LaneLists& lists, double& t, int& prim) {
  // (`ray` is gated by the caller: well_formed() decides about the shortcuts in here too)
  Pair ra = {PRT_INF, PRT_INF, -1}, rb = {PRT_INF, PRT_INF, -1};
  t = PRT_INF;
  prim = -1;
  for (int pc = first; pc < last; ++pc) {
    const DevInstr* in = code + pc;
    const int kind = in->kind;
    if (kind == I_CHAIN) {
      chain_step(reinterpret_cast<const DevChain*>(in), in->a0, ray, t, prim);
      pc += CHAIN_SLOTS - 1;
    } else if (kind == I_BOX) {
      if (well_formed(ray) && !may_reach(in->data, ray, PRT_INF)) return;  // the cull step leads its component
    } else {
      run_step<false>(in, ray, lists, ra, rb, t, prim);
    }
  }
}

template <bool RENDER = false, bool CULL = true, bool ANY_W = true>
__device__ __forceinline__ void nearest_hit(const DevPrim* __restrict__ prims,
                                            const DevInstr* __restrict__ code, int n_instr,
                                            const Ray8& ray, const LaneLists& lists, double& best_t,
                                            int& best_prim) {
  (void)prims;
  // from here on well_formed() is the one gate of every shortcut.  (The gate is opened on this copy and
  // not on one made inside nearest_hit_n: that form cost the generation kernel 12 B of scratch per lane.)
  Ray8 gated_ray = ray;
  gated_ray.gated = true;
  gated_ray.any_w = ANY_W;
  gated_ray.lex = CULL && !RENDER;  // (only programs with cull steps are ever grouped by position)
  const Ray8 rays1[1] = {gated_ray};
  double t1[1];
  int p1[1];
  nearest_hit_n<1, RENDER, CULL, ANY_W>(code, n_instr, rays1, lists.total, t1, p1);
  best_t = t1[0];
  best_prim = p1[0];
}
