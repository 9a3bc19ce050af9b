/*
 * prt_trace_file -- a host for libprt_hip.so that is neither Python nor torch: the C ABI of include/prt.h
 * driven from plain C-style code with the HIP runtime for device memory (what a maintainer binding the
 * library from another language does).
 *
 *   prt_trace_file <scene+rays file> <result file> [depth [surface id]]
 *
 * Input file (little endian, written by tests/test_gpu_c_host.py from a golden fixture):
 *   int64 header[8] = {magic 0x70727431, n_prims, n_nodes, n_roots, n_materials, n_rays, generation_limit, trace flags}
 *   prt_prim[n_prims]  prt_node[n_nodes]  int32 roots[n_roots]  (padded to 8 bytes)  prt_material[n_materials]
 *   double rays[13][n_rays]
 * Result file: int64 total, int64 rows_per_generation[generation_limit], double rows[15][total].
 * depth > 0: the same ray set is traced 2 * depth + 1 times through prt_trace_batch with `depth` traces in flight
 * on as many streams, and every frame must equal the synchronous one bit for bit (exit code 3 otherwise).
 * surface id (with depth 0): the trace runs under a record plan (prt_trace_set_plan) that stores the rows of that surface
 * only and sums them in the generation kernels; behind the rows the result file then holds
 * double sums[generation_limit][PRT_SINK_STATS] (one source group).
 *
 * (Compiled by hipcc as C++ only because the HIP runtime header wants it; nothing below is more than C99.)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <input> <output> [depth]\n", argv[0]);
    return 2;
  }
  const int depth = argc > 3 ? atoi(argv[3]) : 0;
  const int with_plan = argc > 4;
  const long long plan_surface = with_plan ? atoll(argv[4]) : -1;
  FILE* in = fopen(argv[1], "rb");
  if (!in) { perror(argv[1]); return 2; }
  int64_t* header = (int64_t*)read_exact(in, 8 * sizeof(int64_t));
  if (header[0] != 0x70727431) { fprintf(stderr, "not a prt_trace_file input\n"); return 2; }
  const int n_prims = (int)header[1], n_nodes = (int)header[2], n_roots = (int)header[3], n_mats = (int)header[4];
  const int64_t n = header[5];
  const int limit = (int)header[6], flags = (int)header[7];
  prt_prim* prims = (prt_prim*)read_exact(in, (size_t)n_prims * sizeof(prt_prim));
  prt_node* nodes = (prt_node*)read_exact(in, (size_t)n_nodes * sizeof(prt_node));
  const size_t roots_bytes = ((size_t)n_roots * sizeof(int32_t) + 7) / 8 * 8;
  int32_t* roots = (int32_t*)read_exact(in, roots_bytes);
  prt_material* mats = (prt_material*)read_exact(in, (size_t)n_mats * sizeof(prt_material));
  double* rays = (double*)read_exact(in, (size_t)PRT_RAY_ROWS * (size_t)n * sizeof(double));
  fclose(in);

  if (prt_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
  prt_scene* scene = NULL;
  if (prt_scene_create(prims, n_prims, nodes, n_nodes, roots, n_roots, mats, n_mats, NULL, &scene) != 0) {
    fprintf(stderr, "prt_scene_create: %s\n", prt_last_error());
    return 1;
  }
  const int64_t cap = (n > 0 ? n : 1) * (limit > 0 ? limit : 1);
  const int blocks = depth > 0 ? depth + 1 : 1;  /* block 0: the synchronous trace */
  double *d_rays = NULL, *d_rows[PRT_TRACE_TICKETS + 1] = {NULL};
  void* d_work[PRT_TRACE_TICKETS + 1] = {NULL};
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc((void**)&d_rays, (size_t)PRT_RAY_ROWS * (size_t)(n > 0 ? n : 1) * sizeof(double)));
  CHECK_HIP(hipMemcpy(d_rays, rays, (size_t)PRT_RAY_ROWS * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  for (int k = 0; k < blocks; ++k) {
    CHECK_HIP(hipMalloc((void**)&d_rows[k], (size_t)PRT_RECORD_COLS * (size_t)cap * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_work[k], (size_t)prt_trace_workspace_bytes(n)));
  }
  int64_t* counts = (int64_t*)calloc((size_t)(limit > 0 ? limit : 1), sizeof(int64_t));
  double* d_sums = NULL;
  const size_t sums_count = (size_t)(limit > 0 ? limit : 1) * PRT_SINK_STATS;
  if (with_plan) {  /* what the caller keeps, decided in the generation kernel: rows of one surface, and their sums */
    CHECK_HIP(hipMalloc((void**)&d_sums, sums_count * sizeof(double)));
    prt_record_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.struct_size = (int32_t)sizeof(plan);
    plan.n_surfaces = 1;
    plan.surfaces[0] = plan_surface;
    plan.store_rows = 1;
    plan.n_groups = 1;
    plan.sums_out = d_sums;
    plan.ms_quantity = -1;
    plan.generation_limit = limit > 0 ? limit : 1;
    if (prt_trace_set_plan(scene, 0, 0, &plan) != 0) {
      fprintf(stderr, "prt_trace_set_plan: %s\n", prt_last_error());
      return 1;
    }
  }
  const int64_t total = prt_trace(scene, 0, d_rays, n, n, limit, 1e-6, d_rows[0], cap, counts, d_work[0],
                                  flags | PRT_TRACE_SYNC, NULL);
  if (total < 0) {
    fprintf(stderr, "prt_trace: %s\n", prt_last_error());
    return 1;
  }
  double* rows = (double*)malloc((size_t)PRT_RECORD_COLS * (size_t)(total > 0 ? total : 1) * sizeof(double));
  /* the record block is (15, cap): a column is contiguous, the frame's columns are cap apart */
  CHECK_HIP(hipMemcpy2D(rows, (size_t)total * sizeof(double), d_rows[0], (size_t)cap * sizeof(double),
                        (size_t)total * sizeof(double), PRT_RECORD_COLS, hipMemcpyDeviceToHost));

  if (depth > 0) {  /* the same trace, several in flight: prt_trace_batch on `depth` streams of our own */
    const int jobs_n = 2 * depth + 1;
    hipStream_t streams[PRT_TRACE_TICKETS];
    for (int k = 0; k < depth; ++k) CHECK_HIP(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
    prt_trace_job* jobs = (prt_trace_job*)calloc((size_t)jobs_n, sizeof(prt_trace_job));
    int64_t* job_counts = (int64_t*)calloc((size_t)jobs_n * (size_t)(limit > 0 ? limit : 1), sizeof(int64_t));
    for (int k = 0; k < jobs_n; ++k) {
      jobs[k].rays = d_rays; jobs[k].n = n; jobs[k].ld = n;
      jobs[k].rows_out = d_rows[1 + k % depth]; jobs[k].rows_cap = cap;
      jobs[k].rows_per_generation = job_counts + (size_t)k * (size_t)(limit > 0 ? limit : 1);
    }
    const int64_t sum = prt_trace_batch(scene, 0, jobs, jobs_n, limit, 1e-6, depth, d_work + 1, (void* const*)streams,
                                        flags | PRT_TRACE_SYNC);
    if (sum != total * jobs_n) {
      fprintf(stderr, "prt_trace_batch: %lld rows, expected %lld (%s)\n", (long long)sum, (long long)(total * jobs_n),
              sum < 0 ? prt_last_error() : "counts differ");
      return 3;
    }
    double* again = (double*)malloc((size_t)PRT_RECORD_COLS * (size_t)(total > 0 ? total : 1) * sizeof(double));
    for (int k = 0; k < depth; ++k) {  /* the last frame recorded into each of the batch's blocks */
      CHECK_HIP(hipMemcpy2D(again, (size_t)total * sizeof(double), d_rows[1 + k], (size_t)cap * sizeof(double),
                            (size_t)total * sizeof(double), PRT_RECORD_COLS, hipMemcpyDeviceToHost));
      if (memcmp(again, rows, (size_t)PRT_RECORD_COLS * (size_t)total * sizeof(double)) != 0) {
        fprintf(stderr, "frame of batch block %d differs from the synchronous trace\n", k);
        return 3;
      }
    }
    for (int k = 0; k < jobs_n; ++k)
      if (jobs[k].total != total || memcmp(jobs[k].rows_per_generation, counts, (size_t)limit * sizeof(int64_t)) != 0) {
        fprintf(stderr, "job %d: counts differ\n", k);
        return 3;
      }
    for (int k = 0; k < depth; ++k) CHECK_HIP(hipStreamDestroy(streams[k]));
    free(again); free(jobs); free(job_counts);
  }

  FILE* out = fopen(argv[2], "wb");
  if (!out) { perror(argv[2]); return 2; }
  fwrite(&total, sizeof(total), 1, out);
  fwrite(counts, sizeof(int64_t), (size_t)limit, out);
  fwrite(rows, sizeof(double), (size_t)PRT_RECORD_COLS * (size_t)total, out);
  if (with_plan) {  /* (PRT_TRACE_SYNC: the fold behind the trace has run) */
    double* sums = (double*)malloc(sums_count * sizeof(double));
    CHECK_HIP(hipMemcpy(sums, d_sums, sums_count * sizeof(double), hipMemcpyDeviceToHost));
    fwrite(sums, sizeof(double), sums_count, out);
    free(sums);
    (void)hipFree(d_sums);
  }
  fclose(out);
  prt_scene_destroy(scene);
  for (int k = 0; k < blocks; ++k) { (void)hipFree(d_rows[k]); (void)hipFree(d_work[k]); }
  (void)hipFree(d_rays);
  printf("%lld rows over %d generations\n", (long long)total, limit);
  return 0;
}
