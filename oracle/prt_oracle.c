/*
 * prt_oracle.c -- scalar C restatement of PyRayT's batched ray-propagation path.
 *
 * TEST INFRASTRUCTURE ONLY: a second, independent CPU checker next to oracle/prt_oracle.py
 * (and the faster CPU baseline for bench.py).  It may be linked/loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
 *
 * One ray at a time, recursion over the component tree, plain arrays for the hit lists.
 * Every function cites the reference file:line it restates (paths under the PyRayT tree).
 * Parity is PINNED by tests/test_oracle_golden.py against golden vectors generated from the
 * genuine reference (tests/golden/generate_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (see oracle/Makefile).  Contraction is off so that
 * a*b+c is two roundings like numpy's ufuncs; the 4x4 transforms call fma() explicitly
 * because numpy hands those to a BLAS dgemm whose kernels accumulate with FMAs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXLIST 64

typedef struct {
  int n_prims, n_nodes, n_roots, n_mats;
  const int32_t *prim_type, *prim_material, *prim_normal_scale;
  const int64_t* prim_surface_id;
  const double *prim_params, *prim_minv; /* (P,6), (P,16) */
  const int32_t *node_op, *node_left, *node_right, *node_prim;
  const double* node_aabb; /* (N,6) */
  const int32_t* roots;
  const int32_t* mat_kind;
  const double* mat_coef; /* (M,6) */
} scene_t;

enum { SPHERE, CYLINDER, PLANE, CUBE, PARABOLOID };
enum { LEAF, UNION, INTERSECT, DIFFERENCE };
enum { MAT_NONE, MAT_ABSORBER, MAT_MIRROR, MAT_CONST, MAT_SELLMEIER };

/* np.isclose(x, 0) and np.isclose(x, h) with the default tolerances */
static int near0(double x) { return fabs(x) <= 1e-8; }
static int close_to(double x, double h) { return fabs(x - h) <= 1e-8 + 1e-5 * fabs(h); }
static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

/* row r of M.v / M^T.v, dgemm-style FMA accumulation (world_objects.py:367-369, :411) */
static double mrow(const double* m, int r, const double* v) {
  double acc = m[4 * r] * v[0];
  acc = fma(m[4 * r + 1], v[1], acc);
  acc = fma(m[4 * r + 2], v[2], acc);
  acc = fma(m[4 * r + 3], v[3], acc);
  return acc;
}
static double mcol(const double* m, int r, const double* v) {
  double acc = m[r] * v[0];
  acc = fma(m[4 + r], v[1], acc);
  acc = fma(m[8 + r], v[2], acc);
  acc = fma(m[12 + r], v[3], acc);
  return acc;
}
static double norm4(const double* v) { return sqrt(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]); }

/* operations.py:28-63 */
static void binomial_root(double a, double b, double c, double* p) {
  double disc = b * b - 4 * a * c;
  int lin = near0(a);
  double s = sqrt(dmax(0.0, disc));
  double den = 2 * a + (lin ? 1.0 : 0.0);
  p[0] = (-b + s) / den;
  p[1] = (-b - s) / den;
  if (!(disc >= 0)) p[0] = p[1] = INFINITY;
  if (lin) {
    p[0] = p[1] = -c / (b + (b == 0 ? 1.0 : 0.0));
    if (near0(b)) {
      p[0] = (c <= 0) ? -INFINITY : INFINITY;
      p[1] = INFINITY;
    }
  }
}

/* primitives.py:683-703 */
static void z_slab(double oz, double dz, double lo, double hi, double* c) {
  int par = near0(dz);
  double den = dz + (par ? 1.0 : 0.0);
  c[0] = (lo - oz) / den;
  c[1] = (hi - oz) / den;
  if (par) {
    c[0] = (oz >= lo && oz <= hi) ? -INFINITY : INFINITY;
    c[1] = INFINITY;
  }
}

/* primitives.py:705-711 */
static void overlap(const double* a, const double* b, double* h) {
  double lo = dmax(dmin(a[0], a[1]), dmin(b[0], b[1]));
  double hi = dmin(dmax(a[0], a[1]), dmax(b[0], b[1]));
  if (lo <= hi) { h[0] = lo; h[1] = hi; } else { h[0] = h[1] = INFINITY; }
}

/* primitives.py:531-565 / :454-469 */
static void axis_slab(double o, double d, double lo, double hi, int inside, double* s) {
  int z = near0(d);
  double den = d + (z ? 1.0 : 0.0);
  double first = -(o - lo) / den, second = -(o - hi) / den;
  if (z) { first = inside ? -INFINITY : INFINITY; second = INFINITY; }
  s[0] = dmin(first, second);
  s[1] = dmax(first, second);
}

/* primitives.py:516-581 */
static void cube_pair(const double* span, const double* o, const double* d, double* h) {
  double sx[2], sy[2], sz[2];
  axis_slab(o[0], d[0], span[0], span[1], o[0] <= span[1] && o[0] >= span[0], sx);
  axis_slab(o[1], d[1], span[2], span[3], o[1] <= span[3] && o[1] >= span[2], sy);
  axis_slab(o[2], d[2], span[4], span[5], o[2] <= span[5] && o[2] >= span[4], sz);
  double enter = dmax(dmax(sx[0], sy[0]), sz[0]);
  double leave = dmin(dmin(sx[1], sy[1]), sz[1]);
  if (enter < leave) { h[0] = enter; h[1] = leave; } else { h[0] = h[1] = INFINITY; }
}

/* TracerSurface.intersect: world_objects.py:360-383 + the primitive tests */
static void surface_pair(const scene_t* s, int p, const double* ray, double* t) {
  const double* m = s->prim_minv + 16 * p;
  const double* q = s->prim_params + 6 * p;
  double o[3], d[3], h[2];
  for (int r = 0; r < 3; ++r) { o[r] = mrow(m, r, ray); d[r] = mrow(m, r, ray + 4); }
  switch (s->prim_type[p]) {
    case SPHERE: { /* primitives.py:241-271 */
      double a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
      double b = 2 * ((d[0] * o[0] + d[1] * o[1]) + d[2] * o[2]);
      double c = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) - q[0] * q[0];
      double disc = b * b - 4 * a * c;
      double sq = sqrt(dmax(0.0, disc));
      h[0] = (-b + sq) / (2 * a);
      h[1] = (-b - sq) / (2 * a);
      if (!(disc >= 0)) h[0] = h[1] = INFINITY;
    } break;
    case CYLINDER: { /* :650-712 */
      double side[2], cap[2];
      binomial_root(d[0] * d[0] + d[1] * d[1], 2 * (d[0] * o[0] + d[1] * o[1]),
                    (o[0] * o[0] + o[1] * o[1]) - q[0] * q[0], side);
      z_slab(o[2], d[2], q[1], q[2], cap);
      overlap(side, cap, h);
    } break;
    case PLANE: { /* :436-492 */
      double hw = q[0] / 2, hl = q[1] / 2, sx[2], sy[2];
      axis_slab(o[0], d[0], hw, -hw, fabs(o[0]) <= hw, sx);
      axis_slab(o[1], d[1], hl, -hl, fabs(o[1]) <= hl, sy);
      double enter = dmax(sx[0], sy[0]), leave = dmin(sx[1], sy[1]);
      int skew = near0(d[2]);
      double tt = -o[2] / (d[2] + (skew ? 1.0 : 0.0));
      if (skew) tt = INFINITY;
      if (!(tt >= enter && tt <= leave)) tt = INFINITY;
      h[0] = h[1] = tt;
    } break;
    case CUBE:
      cube_pair(q, o, d, h);
      break;
    default: { /* PARABOLOID :320-399 */
      double f4 = 4 * q[0];
      double a = d[0] * d[0] + d[1] * d[1];
      double b = 2 * (o[0] * d[0] + o[1] * d[1]) - f4 * d[2];
      double c = (o[0] * o[0] + o[1] * o[1]) - f4 * o[2];
      double disc = b * b - 4 * a * c;
      int lin = near0(a);
      double sq = sqrt(dmax(0.0, disc));
      double den = 2 * a + (lin ? 1.0 : 0.0);
      double par[2] = {(-b + sq) / den, (-b - sq) / den}, cap[2];
      if (!(disc >= 0)) par[0] = par[1] = INFINITY;
      if (lin) {
        par[0] = -c / (b + (near0(b) ? 1.0 : 0.0));
        par[1] = (d[2] >= 0) ? INFINITY : -INFINITY;
      }
      z_slab(o[2], d[2], 0.0, q[1], cap);
      overlap(par, cap, h);
    } break;
  }
  /* np.sort puts NaN (0/0 of a zero-direction ray) last; the tracer masks it with `> 0` */
  if (h[0] != h[0]) h[0] = INFINITY;
  if (h[1] != h[1]) h[1] = INFINITY;
  t[0] = dmin(h[0], h[1]);
  t[1] = dmax(h[0], h[1]);
}

/* stable insertion sort of (value, tag) by value */
static void stable_sort(double* v, int* tag, int n) {
  for (int i = 1; i < n; ++i) {
    double x = v[i];
    int g = tag[i], j = i - 1;
    while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; tag[j + 1] = tag[j]; --j; }
    v[j + 1] = x;
    tag[j + 1] = g;
  }
}

/* CSGSurface.intersect + array_csg: csg.py:118-160, :13-61.  Returns the list length. */
static int node_hits(const scene_t* s, int node, const double* ray, double* hits, int* prims) {
  int op = s->node_op[node];
  if (op == LEAF) {
    int p = s->node_prim[node];
    surface_pair(s, p, ray, hits);
    prims[0] = prims[1] = p;
    return 2;
  }
  double lv[MAXLIST], rv[MAXLIST], box[2];
  int lp[MAXLIST], rp[MAXLIST];
  int ml = node_hits(s, s->node_left[node], ray, lv, lp);
  int mr = node_hits(s, s->node_right[node], ray, rv, rp);
  int m = ml + mr;
  cube_pair(s->node_aabb + 6 * node, ray, ray + 4, box); /* csg.py:126-128 */
  if (!(isfinite(box[0]) || isfinite(box[1]))) {
    for (int k = 0; k < m; ++k) { hits[k] = INFINITY; prims[k] = -1; }
    return m;
  }
  double v[MAXLIST];
  int src[MAXLIST];
  for (int k = 0; k < ml; ++k) { v[k] = lv[k]; src[k] = k; }
  for (int k = 0; k < mr; ++k) { v[ml + k] = rv[k]; src[ml + k] = ml + k; }
  stable_sort(v, src, m);
  int depth[MAXLIST], running = (op == DIFFERENCE) ? 1 : 0;
  for (int k = 0; k < m; ++k) {
    int odd = src[k] & 1, from_right = src[k] >= ml;
    int step = odd ? -1 : 1;
    if (op == DIFFERENCE && from_right) step = -step;
    running += step;
    depth[k] = running;
  }
  int tag[MAXLIST];
  for (int k = 0; k < m; ++k) {
    int before = depth[(k + m - 1) % m]; /* np.roll wrap-around */
    int keep = (op == UNION) ? ((depth[k] != 0) != (before != 0)) : (depth[k] == 2 || before == 2);
    if (!keep) v[k] = INFINITY;
    tag[k] = src[k];
  }
  stable_sort(v, tag, m);
  for (int k = 0; k < m; ++k) {
    hits[k] = v[k];
    prims[k] = isfinite(v[k]) ? (tag[k] < ml ? lp[tag[k]] : rp[tag[k] - ml]) : -1;
  }
  return m;
}

/* RayTracer._st_propagate: _pyrayt.py:370-392 */
static void propagate(const scene_t* s, const double* ray, double* t_out, int* prim_out) {
  double best = INFINITY;
  int best_prim = -1;
  for (int c = 0; c < s->n_roots; ++c) {
    double hits[MAXLIST];
    int prims[MAXLIST];
    int m = node_hits(s, s->roots[c], ray, hits, prims);
    double t = INFINITY;
    int p = -1;
    for (int k = m - 1; k >= 0; --k)
      if (hits[k] > 0 && hits[k] < INFINITY) { t = hits[k]; p = prims[k]; }
    if (t < best) { best = t; best_prim = p; }
  }
  *t_out = best;
  *prim_out = best_prim;
}

/* TracerSurface.get_world_normals: world_objects.py:401-418 + primitive normals */
static void world_normal(const scene_t* s, int p, const double* point, double* n) {
  const double* m = s->prim_minv + 16 * p;
  const double* q = s->prim_params + 6 * p;
  double l[3], a[4] = {0, 0, 0, 0};
  for (int r = 0; r < 3; ++r) l[r] = mrow(m, r, point);
  int normalise = 1;
  switch (s->prim_type[p]) {
    case SPHERE: a[0] = l[0]; a[1] = l[1]; a[2] = l[2]; break;
    case CYLINDER:
      a[0] = l[0]; a[1] = l[1];
      if (close_to(l[2], q[1])) { a[0] = 0; a[1] = 0; a[2] = -1; }
      if (close_to(l[2], q[2])) { a[0] = 0; a[1] = 0; a[2] = 1; }
      break;
    case PLANE: a[2] = 1; normalise = 0; break;
    case CUBE:
      for (int k = 0; k < 3; ++k)
        a[k] = close_to(l[k], q[2 * k + 1]) ? 1.0 : (close_to(l[k], q[2 * k]) ? -1.0 : 0.0);
      break;
    default:
      a[0] = l[0]; a[1] = l[1]; a[2] = -2 * q[0];
      if (close_to(l[2], q[1])) { a[0] = 0; a[1] = 0; a[2] = 1; }
      break;
  }
  if (normalise) {
    double len = norm4(a);
    for (int k = 0; k < 3; ++k) a[k] /= len;
  }
  double w[4] = {mcol(m, 0, a), mcol(m, 1, a), mcol(m, 2, a), 0.0};
  double len = norm4(w), sgn = (double)s->prim_normal_scale[p];
  for (int k = 0; k < 4; ++k) n[k] = (w[k] / len) * sgn;
}

/* materials.py:47-50, :58-62, :70-75 with operations.py:86-162.  point/dir are 4-vectors. */
static int shade(const scene_t* s, int p, const double* point, double* d, double wavelength,
                 double* index) {
  int mat = s->prim_material[p];
  int kind = s->mat_kind[mat];
  const double* k = s->mat_coef + 6 * mat;
  if (kind == MAT_NONE) return 0;
  if (kind == MAT_ABSORBER) { d[0] = d[1] = d[2] = d[3] = 0.0; return 1; }
  double n[4];
  world_normal(s, p, point, n);
  if (kind == MAT_MIRROR) {
    double dot = ((d[0] * n[0] + d[1] * n[1]) + d[2] * n[2]) + d[3] * n[3];
    for (int c = 0; c < 4; ++c) d[c] = d[c] - (2 * n[c]) * dot;
    return 1;
  }
  double n_mat;
  if (kind == MAT_CONST) n_mat = k[0];
  else {
    double w2 = wavelength * wavelength;
    n_mat = sqrt(((1 + (k[0] * w2) / (w2 - k[3])) + (k[1] * w2) / (w2 - k[4])) + (k[2] * w2) / (w2 - k[5]));
  }
  double len = norm4(d), v[4], mn[4], u[4];
  for (int c = 0; c < 4; ++c) v[c] = d[c] / len;
  double cos_p = ((v[0] * n[0] + v[1] * n[1]) + v[2] * n[2]) + v[3] * n[3];
  double cos_n = ((v[0] * -n[0] + v[1] * -n[1]) + v[2] * -n[2]) + v[3] * -n[3];
  int leaving = cos_p > 0;
  double n2 = leaving ? 1.0 : n_mat, n1 = *index;
  for (int c = 0; c < 4; ++c) mn[c] = leaving ? -n[c] : n[c];
  double r = n1 / n2, cos1 = leaving ? cos_p : cos_n;
  double radicand = 1 - (r * r) * (1 - cos1 * cos1);
  double cos2 = sqrt(dmax(0.0, radicand));
  if (radicand > 0) {
    double kk = r * cos1 - cos2;
    for (int c = 0; c < 4; ++c) u[c] = r * v[c] + kk * mn[c];
  } else {
    double kk = 2 * cos1;
    for (int c = 0; c < 4; ++c) u[c] = v[c] + kk * mn[c];
  }
  double ulen = norm4(u);
  for (int c = 0; c < 4; ++c) d[c] = u[c] / ulen;
  *index = (radicand > 0) ? n2 : n1;
  return 1;
}

/* ---- exported --------------------------------------------------------------------------- */

/* nearest hit of n rays ((13,n) row-major, leading dimension ld): t (n), surface id (n) */
void prt_oracle_propagate(const scene_t* s, const double* rays, int64_t n, int64_t ld, double* t_out,
                          int64_t* surf_out) {
  for (int64_t i = 0; i < n; ++i) {
    double ray[8];
    int prim;
    for (int r = 0; r < 8; ++r) ray[r] = rays[r * ld + i];
    propagate(s, ray, &t_out[i], &prim);
    surf_out[i] = prim >= 0 ? s->prim_surface_id[prim] : -1;
  }
}

/* RayTracer.trace() from an initial ray set: _pyrayt.py:329-339 driving :370-452 and the
 * record writer :168-186.  rows_out is (rows_cap, 15) row-major (the frame's own layout).
 * Returns the number of rows, -1 if rows_cap is too small, -5 on an untracable surface. */
int64_t prt_oracle_trace(const scene_t* s, const double* rays, int64_t n, int64_t ld,
                         int generation_limit, double ray_offset, double* rows_out,
                         int64_t rows_cap, int64_t* rows_per_generation) {
  double* cur = (double*)malloc(sizeof(double) * 13 * (size_t)(n > 0 ? n : 1));
  double* nxt = (double*)malloc(sizeof(double) * 13 * (size_t)(n > 0 ? n : 1));
  for (int r = 0; r < 13; ++r)
    for (int64_t i = 0; i < n; ++i) cur[r * n + i] = rays[r * ld + i];
  int64_t count = n, total = 0, status = 0;
  for (int g = 0; g < generation_limit && count > 0; ++g) {
    int64_t live = 0;
    for (int64_t i = 0; i < count && status == 0; ++i) {
      double ray[13], t;
      int prim;
      for (int r = 0; r < 13; ++r) ray[r] = cur[r * n + i];
      propagate(s, ray, &t, &prim);
      /* dead: absorbed (|d| ~ 0 before the interaction) or no hit (_pyrayt.py:415-420) */
      if (near0(norm4(ray + 4)) || prim < 0) continue;
      if (total + live >= rows_cap) { status = -1; break; }
      double point[4], d[4] = {ray[4], ray[5], ray[6], ray[7]}, index = ray[11];
      for (int c = 0; c < 4; ++c) point[c] = ray[c] + ray[4 + c] * t;
      if (!shade(s, prim, point, d, ray[10], &index)) { status = -5; break; }
      double* row = rows_out + 15 * (total + live);
      double tilt = sqrt((ray[4] * ray[4] + ray[5] * ray[5]) + ray[6] * ray[6]);
      for (int c = 0; c < 5; ++c) row[c] = ray[8 + c];
      row[5] = (double)s->prim_surface_id[prim];
      for (int c = 0; c < 3; ++c) { row[6 + c] = ray[c]; row[9 + c] = point[c]; row[12 + c] = ray[4 + c] / tilt; }
      /* survivors keep their order; absorbed rays stay in the set until the next generation
       * finds them dead (Q3) */
      for (int c = 0; c < 4; ++c) {
        double q = point[c];
        if (g + 1 != generation_limit) q = point[c] + ray_offset * d[c];
        nxt[c * n + live] = q;
        nxt[(4 + c) * n + live] = d[c];
      }
      nxt[8 * n + live] = (double)(g + 1);
      nxt[9 * n + live] = ray[9];
      nxt[10 * n + live] = ray[10];
      nxt[11 * n + live] = index;
      nxt[12 * n + live] = ray[12];
      ++live;
    }
    if (status != 0 || live == 0) break;
    rows_per_generation[g] = live;
    total += live;
    count = live;
    double* swap = cur; cur = nxt; nxt = swap;
  }
  free(cur);
  free(nxt);
  return status != 0 ? status : total;
}

/* build a scene_t view over caller-owned arrays (plain pointers: ctypes friendly) */
scene_t* prt_oracle_scene(int n_prims, int n_nodes, int n_roots, int n_mats, const int32_t* prim_type,
                          const int32_t* prim_material, const int32_t* prim_normal_scale,
                          const int64_t* prim_surface_id, const double* prim_params,
                          const double* prim_minv, const int32_t* node_op, const int32_t* node_left,
                          const int32_t* node_right, const int32_t* node_prim, const double* node_aabb,
                          const int32_t* roots, const int32_t* mat_kind, const double* mat_coef) {
  scene_t* s = (scene_t*)calloc(1, sizeof(scene_t));
  s->n_prims = n_prims; s->n_nodes = n_nodes; s->n_roots = n_roots; s->n_mats = n_mats;
  s->prim_type = prim_type; s->prim_material = prim_material; s->prim_normal_scale = prim_normal_scale;
  s->prim_surface_id = prim_surface_id; s->prim_params = prim_params; s->prim_minv = prim_minv;
  s->node_op = node_op; s->node_left = node_left; s->node_right = node_right; s->node_prim = node_prim;
  s->node_aabb = node_aabb; s->roots = roots; s->mat_kind = mat_kind; s->mat_coef = mat_coef;
  return s;
}
void prt_oracle_scene_free(scene_t* s) { free(s); }
