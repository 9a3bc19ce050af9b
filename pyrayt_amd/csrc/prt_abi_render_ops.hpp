// prt_abi_render_ops.hpp -- extern "C" entry points of include/prt.h either side of the trace: the renderers
// (prt_camera_rays, prt_render_hits, prt_gooch_*, prt_render, prt_edge_*) and tinygfx.g3d.operations / primitives as
// functions (prt_reflect, prt_refract, prt_binomial_root, prt_dot, prt_array_csg, prt_primitive_*).  Host code; the
// kernels are prt_render.hpp / prt_ops.hpp.  Included by prt_kernels.hip.
#pragma once
// ------------------------------------------------------------------------------------------------
// renderers
// ------------------------------------------------------------------------------------------------
static int camera_of(const prt_camera* cam, DevCamera* out) {
  if (!cam) return fail(PRT_ERR_ARG, "camera is null");
  if (cam->h_pixels < 0 || cam->v_pixels < 0) return fail(PRT_ERR_ARG, "negative camera resolution");
  std::memcpy(out->world, cam->world, sizeof(out->world));
  out->h_pixels = cam->h_pixels; out->v_pixels = cam->v_pixels;
  out->h_width = cam->h_width; out->v_width = cam->v_width;
  return PRT_OK;
}

extern "C" int prt_camera_rays(int device, const prt_camera* camera, int64_t first, int64_t count,
                               double* rays_out, int64_t ld, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  DevCamera cam;
  int rc = camera_of(camera, &cam);
  if (rc) return rc;
  if (first < 0 || count < 0 || first + count > cam.h_pixels * cam.v_pixels || ld < count ||
      (count && !rays_out))
    return fail(PRT_ERR_ARG, "bad pixel range / output buffer");
  if (count == 0) return PRT_OK;
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(k_camera, dim3(blocks_for(count)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, cam,
                     first, count, rays_out, ld);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

static SceneDev render_scene_dev(const prt_scene* s, const DeviceCopy* c) {
  return SceneDev{c->prims, c->render_code, (int)s->render_program.code.size(), s->render_program.lds_slots};
}

extern "C" int prt_render_hits(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                               double* t_out, int64_t* surf_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld < n || (n && (!rays || !t_out || !surf_out)))
    return fail(PRT_ERR_ARG, "bad ray / output buffers");
  if (n == 0) return PRT_OK;
  SceneDev sd = render_scene_dev(s, c);
  hipLaunchKernelGGL(k_render_hits, dim3(blocks_for(n)), dim3(PRT_BLOCK), lds_bytes(sd.lds_slots),
                     (hipStream_t)stream, sd, rays, ld, n, t_out, surf_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_gooch_shade(prt_scene* s, int device, const double* rays, int64_t n, int64_t ld,
                               const double* t, const int64_t* surf, const double* gooch,
                               const double* light, double* rgba_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  if (n < 0 || ld < n || !light || (n && (!rays || !t || !surf || !gooch || !rgba_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (n == 0) return PRT_OK;
  hipLaunchKernelGGL(k_gooch, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, c->prims,
                     (int)s->prims.size(), rays, ld, n, t, surf, gooch, light[0], light[1], light[2],
                     rgba_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_gooch_mix(int device, const double* points, const double* normals, int64_t n,
                             int64_t ld, const double* shade, const double* light, double* rgba_out,
                             int64_t ld_out, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  if (n < 0 || ld < n || ld_out < n || !shade || !light || (n && (!points || !normals || !rgba_out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  if (n == 0) return PRT_OK;
  HIP_TRY(hipSetDevice(device));
  GoochShade g;
  std::memcpy(g.warm, shade, sizeof(g.warm));
  std::memcpy(g.cool, shade + 4, sizeof(g.cool));
  hipLaunchKernelGGL(k_gooch_mix, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, points,
                     normals, ld, n, g, light[0], light[1], light[2], rgba_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_render(prt_scene* s, int device, const prt_camera* camera, int64_t first,
                          int64_t count, const double* gooch, const double* light, double* rgba_out,
                          double* t_out, int64_t* surf_out, void* stream) {
  DeviceCopy* c;
  int rc = on_device(s, device, &c);
  if (rc) return rc;
  DevCamera cam;
  rc = camera_of(camera, &cam);
  if (rc) return rc;
  if (first < 0 || count < 0 || first + count > cam.h_pixels * cam.v_pixels)
    return fail(PRT_ERR_ARG, "bad pixel range");
  if (rgba_out && (!gooch || !light)) return fail(PRT_ERR_ARG, "shading needs the gooch table and a light");
  if (!rgba_out && !t_out && !surf_out) return fail(PRT_ERR_ARG, "no output requested");
  if (count == 0) return PRT_OK;
  SceneDev sd = render_scene_dev(s, c);
  const double lx = light ? light[0] : 0.0, ly = light ? light[1] : 0.0, lz = light ? light[2] : 0.0;
  hipLaunchKernelGGL(k_render, dim3(blocks_for(count)), dim3(PRT_BLOCK), lds_bytes(sd.lds_slots),
                     (hipStream_t)stream, sd, cam, first, count, gooch, lx, ly, lz, rgba_out, t_out,
                     surf_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int64_t prt_edge_workspace_bytes(int64_t h_pixels, int64_t v_pixels) {
  if (h_pixels < 0 || v_pixels < 0) return PRT_ERR_ARG;
  return (int64_t)align_up((size_t)(h_pixels * v_pixels) + 1, 256);
}

extern "C" int prt_edge_canvas(int device, const int64_t* surf, int64_t h_pixels, int64_t v_pixels,
                               int rings, double* rgba_out, void* workspace, void* stream) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  if (h_pixels < 0 || v_pixels < 0 || rings < 0) return fail(PRT_ERR_ARG, "bad picture size");
  const int64_t n = h_pixels * v_pixels;
  if (n == 0) return PRT_OK;
  if (!surf || !rgba_out || !workspace) return fail(PRT_ERR_ARG, "null buffer");
  HIP_TRY(hipSetDevice(device));
  unsigned char* seed = (unsigned char*)workspace;
  hipLaunchKernelGGL(k_edge_seed, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, surf,
                     h_pixels, v_pixels, seed);
  hipLaunchKernelGGL(k_edge_canvas, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream,
                     (const unsigned char*)seed, h_pixels, v_pixels, rings, rgba_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// ------------------------------------------------------------------------------------------------
// tinygfx/g3d/operations.py as entry points
// ------------------------------------------------------------------------------------------------
static int ops_device(int device) {
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  return PRT_OK;
}

extern "C" int prt_reflect(int device, const double* vectors, const double* normals, int rows, int64_t n,
                           int64_t ld, double* out, int64_t ld_out, void* stream) {
  if (rows < 1 || rows > 4 || n < 0 || ld < n || ld_out < n || (n && (!vectors || !normals || !out)))
    return fail(PRT_ERR_ARG, "bad buffers (vectors of 1..4 components)");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_reflect, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, vectors, normals,
                     rows, ld, n, out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_refract(int device, double* vectors, const double* normals, const double* n1,
                           const double* n2, double n_global, int rows, int64_t n, int64_t ld, double* out,
                           int64_t ld_out, double* index_out, void* stream) {
  if (rows < 1 || rows > 4 || n < 0 || ld < n || ld_out < n ||
      (n && (!vectors || !normals || !n1 || !n2 || !out || !index_out)))
    return fail(PRT_ERR_ARG, "bad buffers (vectors of 1..4 components)");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_refract, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, vectors, normals,
                     n1, n2, n_global, rows, ld, n, out, ld_out, index_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_binomial_root(int device, const double* a, const double* b, const double* c, int64_t n,
                                 double* roots_out, int64_t ld_out, void* stream) {
  if (n < 0 || ld_out < n || (n && (!a || !b || !c || !roots_out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_binomial_root, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, a, b, c, n,
                     roots_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_smallest_positive_root(int device, const double* a, const double* b, const double* c,
                                          int64_t n, double* out, void* stream) {
  if (n < 0 || (n && (!a || !b || !c || !out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_smallest_positive_root, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, a,
                     b, c, n, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_dot(int device, const double* m1, const double* m2, int64_t reduce_len,
                       int64_t reduce_stride, int64_t out_len, int64_t out_stride, double* out, void* stream) {
  if (reduce_len < 0 || out_len < 0 || (out_len && (!m1 || !m2 || !out))) return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || out_len == 0) return rc;
  hipLaunchKernelGGL(k_dot, dim3(blocks_for(out_len)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, m1, m2,
                     reduce_len, reduce_stride, out_len, out_stride, out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_array_csg(int device, const double* left, int m_left, const double* right, int m_right,
                             int64_t n, int64_t ld, int op, int sort_output, double* out, int64_t ld_out,
                             void* stream) {
  if (m_left < 0 || m_right < 0 || (m_left & 1) || (m_right & 1))
    return fail(PRT_ERR_ARG, "hit lists hold enter/exit pairs: an even number of rows each");
  if (op < PRT_NODE_UNION || op > PRT_NODE_DIFFERENCE) return fail(PRT_ERR_ARG, "operation is invalid");
  if (n < 0 || ld < n || ld_out < n || (n && ((m_left && !left) || (m_right && !right) || !out)))
    return fail(PRT_ERR_ARG, "bad buffers");
  int rc = ops_device(device);
  if (rc || n == 0 || m_left + m_right == 0) return rc;
  hipLaunchKernelGGL(k_array_csg, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, left, m_left,
                     right, m_right, ld, n, op, sort_output, out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

static int primitive_args(int type, const double* params, PrimParams* out) {
  if (type < PRT_PRIM_SPHERE || type > PRT_PRIM_PARABOLOID) return fail(PRT_ERR_ARG, "unknown primitive type");
  if (!params) return fail(PRT_ERR_ARG, "params is null");
  std::memcpy(out->q, params, sizeof(out->q));
  return PRT_OK;
}

extern "C" int prt_primitive_intersect(int device, int type, const double* params, const double* rays,
                                       int64_t n, int64_t ld, double* hits_out, int64_t ld_out, void* stream) {
  PrimParams q;
  int rc = primitive_args(type, params, &q);
  if (rc) return rc;
  if (n < 0 || ld < n || ld_out < n || (n && (!rays || !hits_out))) return fail(PRT_ERR_ARG, "bad buffers");
  rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_primitive_intersect, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, type, q,
                     rays, ld, n, hits_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

extern "C" int prt_primitive_normal(int device, int type, const double* params, const double* points,
                                    int64_t n, int64_t ld, double* normals_out, int64_t ld_out, void* stream) {
  PrimParams q;
  int rc = primitive_args(type, params, &q);
  if (rc) return rc;
  if (n < 0 || ld < n || ld_out < n || (n && (!points || !normals_out))) return fail(PRT_ERR_ARG, "bad buffers");
  rc = ops_device(device);
  if (rc || n == 0) return rc;
  hipLaunchKernelGGL(k_primitive_normal, dim3(blocks_for(n)), dim3(PRT_BLOCK), 0, (hipStream_t)stream, type, q,
                     points, ld, n, normals_out, ld_out);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}
