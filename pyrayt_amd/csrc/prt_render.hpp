// prt_render.hpp -- the renderers' per-pixel work on the device.
//
// Second consumer of the intersect path: an orthographic camera grid is pushed through the same
// scene program as the tracer's rays, then shaded (Gooch) or edge-detected.
//   tinygfx/g3d/world_objects.py  OrthographicCamera.generate_rays :519-537, TracerSurface.shade :385-399
//   tinygfx/g3d/renderers.py      _st_propagate :70-92 / :187-209, _st_interact :94-116 / :211-236
//   tinygfx/g3d/materials/gooch.py  GoochMaterial.shade :30-65
// Included by prt_kernels.hip after the source kernels (uses SceneDev, nearest_hit, linspace_at).
#pragma once

struct DevCamera {
  double world[16];
  int64_t h_pixels, v_pixels;
  double h_width, v_width;
};

// pixel k of the row-major v x h grid: origin (0, y_h, z_v, 1), direction +x, in camera space;
// world transform as a dgemm-style FMA chain, direction renormalised (world_objects.py:519-537)
__device__ __forceinline__ Ray8 camera_ray(const DevCamera& cam, int64_t pixel) {
  const int64_t iv = pixel / cam.h_pixels, ih = pixel - iv * cam.h_pixels;
  const double y = linspace_at(cam.h_width / 2, -cam.h_width / 2, cam.h_pixels, ih);
  const double z = linspace_at(cam.v_width / 2, -cam.v_width / 2, cam.v_pixels, iv);
  double wo[4], wd[4];
  for (int r = 0; r < 4; ++r) {
    wo[r] = row_dot(cam.world, r, 0.0, y, z, 1.0);
    wd[r] = row_dot(cam.world, r, 1.0, 0.0, 0.0, 0.0);
  }
  const double len = norm4(wd[0], wd[1], wd[2], wd[3]);
  return Ray8{wo[0], wo[1], wo[2], wo[3], wd[0] / len, wd[1] / len, wd[2] / len, wd[3] / len};
}

// TracerSurface.shade + GoochMaterial.shade for one pixel and a single light:
// hit point, world normal, unit vector to the light, warm/cool mix by 0.5 (1 + l.n).
// g = shade_warm[4] | shade_cool[4] of the surface's material (gooch.py:36-37, host-side).
__device__ __forceinline__ void gooch_pixel(const DevPrim* __restrict__ prim, const double* __restrict__ g,
                                            const Ray8& r, double t, double lx, double ly, double lz,
                                            double (&rgba)[4]) {
  const double px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz, pw = r.ow + t * r.dw;
  double nx, ny, nz;
  world_normal(prim, px, py, pz, pw, nx, ny, nz);
  double vx = lx - px, vy = ly - py, vz = lz - pz;
  const double len = norm3(vx, vy, vz);
  vx /= len; vy /= len; vz /= len;
  const double cosine = (vx * nx + vy * ny) + vz * nz;
  const double mix = 0.5 * (1 + cosine);
  const double rest = 1 - mix;
  for (int c = 0; c < 4; ++c) rgba[c] = g[c] * mix + g[4 + c] * rest;
}

// the mix alone, for callers that bring their own points and normals (GoochMaterial.shade)
struct GoochShade { double warm[4], cool[4]; };
__global__ void __launch_bounds__(PRT_BLOCK)
k_gooch_mix(const double* __restrict__ points, const double* __restrict__ normals, int64_t ld, int64_t n,
            GoochShade shade, double lx, double ly, double lz, double* __restrict__ rgba_out, int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double vx = lx - points[i], vy = ly - points[ld + i], vz = lz - points[2 * ld + i];
  const double len = norm3(vx, vy, vz);
  vx /= len; vy /= len; vz /= len;
  const double cosine = (vx * normals[i] + vy * normals[ld + i]) + vz * normals[2 * ld + i];
  const double mix = 0.5 * (1 + cosine);
  const double rest = 1 - mix;
  for (int c = 0; c < 4; ++c) rgba_out[c * ld_out + i] = shade.warm[c] * mix + shade.cool[c] * rest;
}

__global__ void __launch_bounds__(PRT_BLOCK)
k_camera(DevCamera cam, int64_t first, int64_t count, double* __restrict__ rays, int64_t ld) {
  const int64_t k = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (k >= count) return;
  const Ray8 r = camera_ray(cam, first + k);
  rays[0 * ld + k] = r.ox; rays[1 * ld + k] = r.oy; rays[2 * ld + k] = r.oz; rays[3 * ld + k] = r.ow;
  rays[4 * ld + k] = r.dx; rays[5 * ld + k] = r.dy; rays[6 * ld + k] = r.dz; rays[7 * ld + k] = r.dw;
}

// [_st_propagate of both renderers] nearest hit under the renderers' rule for prepared rays
__global__ void __launch_bounds__(PRT_BLOCK)
k_render_hits(SceneDev scene, const double* __restrict__ rays, int64_t ld, int64_t n,
              double* __restrict__ hit_t, int64_t* __restrict__ surf_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const LaneLists lists = lane_lists(scene.lds_slots);
  if (i >= n) return;
  double t;
  int prim;
  nearest_hit<true>(scene.prims, scene.code, scene.n_instr, load_ray8(rays, ld, i), lists, t, prim);
  hit_t[i] = t;
  surf_out[i] = prim >= 0 ? (int64_t)scene.prims[prim].surface_id : -1;
}

// [ShadedRenderer._st_interact] pixels that saw a surface get its Gooch colour, the rest (0,0,0,0)
__global__ void __launch_bounds__(PRT_BLOCK)
k_gooch(const DevPrim* __restrict__ prims, int n_prims, const double* __restrict__ rays, int64_t ld,
        int64_t n, const double* __restrict__ hit_t, const int64_t* __restrict__ surf,
        const double* __restrict__ gooch, double lx, double ly, double lz, double* __restrict__ rgba_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double rgba[4] = {0.0, 0.0, 0.0, 0.0};
  const int prim = prim_of_surface(prims, n_prims, surf[i]);
  if (prim >= 0) gooch_pixel(prims + prim, gooch + 8 * prim, load_ray8(rays, ld, i), hit_t[i], lx, ly, lz, rgba);
  for (int c = 0; c < 4; ++c) rgba_out[4 * i + c] = rgba[c];
}

// Whole frame in one pass: camera ray in registers -> nearest hit (renderers' rule) -> Gooch
// colour.  No HBM reads besides the scene program; 32 B/pixel written (+16 B with t/surface).
__global__ void __launch_bounds__(PRT_BLOCK)
k_render(SceneDev scene, DevCamera cam, int64_t first, int64_t count, const double* __restrict__ gooch,
         double lx, double ly, double lz, double* __restrict__ rgba_out, double* __restrict__ t_out,
         int64_t* __restrict__ surf_out) {
  const int64_t k = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  const LaneLists lists = lane_lists(scene.lds_slots);
  if (k >= count) return;
  const Ray8 r = camera_ray(cam, first + k);
  double t;
  int prim;
  nearest_hit<true>(scene.prims, scene.code, scene.n_instr, r, lists, t, prim);
  if (t_out) t_out[k] = t;
  if (surf_out) surf_out[k] = prim >= 0 ? (int64_t)scene.prims[prim].surface_id : -1;
  if (rgba_out) {
    double rgba[4] = {0.0, 0.0, 0.0, 0.0};
    if (prim >= 0) gooch_pixel(scene.prims + prim, gooch + 8 * prim, r, t, lx, ly, lz, rgba);
    for (int c = 0; c < 4; ++c) rgba_out[4 * k + c] = rgba[c];
  }
}

// [EdgeRender._st_interact, first half] a pixel is an edge seed when its surface id differs from
// its left or its upper neighbour; outside the picture counts as -1 (np.diff(..., prepend=-1))
__global__ void __launch_bounds__(PRT_BLOCK)
k_edge_seed(const int64_t* __restrict__ surf, int64_t h, int64_t v, unsigned char* __restrict__ seed) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= h * v) return;
  const int64_t row = i / h, col = i - row * h;
  const int64_t here = surf[i];
  const int64_t left = col > 0 ? surf[i - 1] : -1;
  const int64_t up = row > 0 ? surf[i - h] : -1;
  seed[i] = (here != left || here != up) ? 1 : 0;
}

// [second half] `rings` binary dilations with the full 3x3 structure = any seed within Chebyshev
// distance `rings`; edges are opaque black, everything else transparent white (renderers.py:103-115)
__global__ void __launch_bounds__(PRT_BLOCK)
k_edge_canvas(const unsigned char* __restrict__ seed, int64_t h, int64_t v, int rings,
              double* __restrict__ rgba_out) {
  const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (i >= h * v) return;
  const int64_t row = i / h, col = i - row * h;
  bool edge = false;
  for (int64_t rr = row - rings; rr <= row + rings; ++rr) {
    if (rr < 0 || rr >= v) continue;
    for (int64_t cc = col - rings; cc <= col + rings; ++cc) {
      if (cc < 0 || cc >= h) continue;
      edge = edge || seed[rr * h + cc] != 0;
    }
  }
  const double ink = edge ? 1.0 : 0.0;
  rgba_out[4 * i + 0] = 1.0 - ink;
  rgba_out[4 * i + 1] = 1.0 - ink;
  rgba_out[4 * i + 2] = 1.0 - ink;
  rgba_out[4 * i + 3] = ink;
}
