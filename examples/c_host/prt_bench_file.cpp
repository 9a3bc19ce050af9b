/*
 * prt_bench_file -- the benchmark's timed region from a host without Python or torch: the ray set of an input
 * file (format: prt_trace_file.cpp) traced `steps` times through prt_trace_batch with `depth` traces in flight on
 * `depth` HIP streams, timed on the host clock between two device synchronisations.
 *
 *   prt_bench_file <scene+rays file> [steps=200] [depth=2] [warmup=20]
 *
 * prints one line: rays, rows per step, ms per step, rows/s.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "prt.h"

#define CHECK_HIP(call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                         \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

static void* read_exact(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read (%zu bytes wanted)\n", bytes);
    exit(2);
  }
  return p;
}

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <input> [steps] [depth] [warmup]\n", argv[0]);
    return 2;
  }
  const int steps = argc > 2 ? atoi(argv[2]) : 200;
  const int depth = argc > 3 ? atoi(argv[3]) : 2;
  const int warmup = argc > 4 ? atoi(argv[4]) : 20;
  if (steps < 1 || depth < 1 || depth > PRT_TRACE_TICKETS || warmup < 0) { fprintf(stderr, "bad arguments\n"); return 2; }
  FILE* in = fopen(argv[1], "rb");
  if (!in) { perror(argv[1]); return 2; }
  int64_t* header = (int64_t*)read_exact(in, 8 * sizeof(int64_t));
  if (header[0] != 0x70727431) { fprintf(stderr, "not a prt_trace_file input\n"); return 2; }
  const int n_prims = (int)header[1], n_nodes = (int)header[2], n_roots = (int)header[3], n_mats = (int)header[4];
  const int64_t n = header[5];
  const int limit = (int)header[6], flags = (int)header[7];
  prt_prim* prims = (prt_prim*)read_exact(in, (size_t)n_prims * sizeof(prt_prim));
  prt_node* nodes = (prt_node*)read_exact(in, (size_t)n_nodes * sizeof(prt_node));
  int32_t* roots = (int32_t*)read_exact(in, ((size_t)n_roots * sizeof(int32_t) + 7) / 8 * 8);
  prt_material* mats = (prt_material*)read_exact(in, (size_t)n_mats * sizeof(prt_material));
  double* rays = (double*)read_exact(in, (size_t)PRT_RAY_ROWS * (size_t)n * sizeof(double));
  fclose(in);
  if (n < 1 || limit < 1) { fprintf(stderr, "nothing to trace\n"); return 2; }

  prt_scene* scene = NULL;
  if (prt_scene_create(prims, n_prims, nodes, n_nodes, roots, n_roots, mats, n_mats, NULL, &scene) != 0) {
    fprintf(stderr, "prt_scene_create: %s\n", prt_last_error());
    return 1;
  }
  CHECK_HIP(hipSetDevice(0));
  const int64_t cap = n * limit;
  double *d_rays = NULL, *d_rows[PRT_TRACE_TICKETS];
  void* d_work[PRT_TRACE_TICKETS];
  hipStream_t streams[PRT_TRACE_TICKETS];
  CHECK_HIP(hipMalloc((void**)&d_rays, (size_t)PRT_RAY_ROWS * (size_t)n * sizeof(double)));
  CHECK_HIP(hipMemcpy(d_rays, rays, (size_t)PRT_RAY_ROWS * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  for (int k = 0; k < depth; ++k) {
    CHECK_HIP(hipMalloc((void**)&d_rows[k], (size_t)PRT_RECORD_COLS * (size_t)cap * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_work[k], (size_t)prt_trace_workspace_bytes(n)));
    CHECK_HIP(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
  }
  const int most = steps > warmup ? steps : warmup;
  prt_trace_job* jobs = (prt_trace_job*)calloc((size_t)most, sizeof(prt_trace_job));
  int64_t* counts = (int64_t*)calloc((size_t)most * (size_t)limit, sizeof(int64_t));
  for (int k = 0; k < most; ++k) {
    jobs[k].rays = d_rays; jobs[k].n = n; jobs[k].ld = n;
    jobs[k].rows_out = d_rows[k % depth]; jobs[k].rows_cap = cap;
    jobs[k].rows_per_generation = counts + (size_t)k * (size_t)limit;
  }
  const int run_flags = flags | PRT_TRACE_NO_TIMING;
  /* spin the clocks up and let the scene take its hints, then warm up */
  for (double t0 = now_ms(); now_ms() - t0 < 60.0;)
    if (prt_trace_batch(scene, 0, jobs, 4 < most ? 4 : most, limit, 1e-6, depth, d_work, (void* const*)streams, run_flags) < 0) {
      fprintf(stderr, "prt_trace_batch: %s\n", prt_last_error());
      return 1;
    }
  if (warmup && prt_trace_batch(scene, 0, jobs, warmup, limit, 1e-6, depth, d_work, (void* const*)streams, run_flags) < 0) {
    fprintf(stderr, "prt_trace_batch: %s\n", prt_last_error());
    return 1;
  }
  CHECK_HIP(hipDeviceSynchronize());
  const double t0 = now_ms();
  const int64_t rows = prt_trace_batch(scene, 0, jobs, steps, limit, 1e-6, depth, d_work, (void* const*)streams, run_flags);
  CHECK_HIP(hipDeviceSynchronize());
  const double ms = now_ms() - t0;
  if (rows < 0) { fprintf(stderr, "prt_trace_batch: %s\n", prt_last_error()); return 1; }
  printf("{\"host\": \"C, no torch\", \"rays\": %lld, \"depth\": %d, \"steps\": %d, \"rows_per_step\": %lld, "
         "\"ms_per_step\": %.5f, \"rows_per_s\": %.4e}\n",
         (long long)n, depth, steps, (long long)(rows / steps), ms / steps, (double)rows / (ms * 1e-3));
  prt_scene_destroy(scene);
  return 0;
}
