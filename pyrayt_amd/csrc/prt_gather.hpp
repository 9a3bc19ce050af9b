// prt_gather.hpp -- re-assembly of the result frame across the GPUs of a node.
//
// The trace itself shards with no communication (contiguous id ranges per rank).  The one exchange
// is putting the per-rank record blocks back into the reference's row order
// (pyrayt/_pyrayt.py:168-186: generation-major, ascending ray id inside a generation, which with
// contiguous shards and order-preserving compaction is rank-major):
//   1. all-gather of the (limit) rows-per-generation vector -> the (G, limit) count matrix; the
//      host needs it to size the assembled frame, so this step synchronises (a few hundred bytes)
//   2. fifteen grouped RCCL all-gathers, one per record column, straight out of the (15, cap) record
//      block (each column is contiguous) into a staging area [column][rank][pad]
//   3. one placement kernel: output column j finds its (generation, rank) segment in a small
//      prefix table and copies its fifteen values -- coalesced on both sides
// RCCL is resolved with dlopen at first use (the library loads, and every other entry point
// works, on a box without it).  The exchange step is separable: prt_place_rows takes a staging area
// filled by any transport (the gloo tests fill it through torch.distributed).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.handle) return PRT_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(PRT_ERR_HIP, std::string("RCCL is not available: ") + dlerror());
  RcclApi api;
  api.handle = h;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
  api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
  api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
  api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GroupStart ||
      !api.GroupEnd || !api.GetErrorString)
    return fail(PRT_ERR_HIP, "librccl lacks an expected symbol");
  g_rccl = api;
  return PRT_OK;
}

#define RCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return fail(PRT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));             \
  } while (0)

struct prt_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  int64_t* dev_counts = nullptr;   // [limit] mine | [world * limit] everybody's
  int64_t* host_counts = nullptr;  // pinned mirror of the same
};

static const int kGatherMaxLimit = 1024;  // == kMaxGenerationSlots

extern "C" int prt_comm_unique_id(char* id128) {
  if (!id128) return fail(PRT_ERR_ARG, "null argument");
  int rc = rccl_load();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return PRT_OK;
}

extern "C" int prt_comm_create(int device, int world, int rank, const char* id128, prt_comm** out) {
  if (!out) return fail(PRT_ERR_ARG, "out is null");
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(PRT_ERR_ARG, "bad communicator arguments");
  int rc = rccl_load();
  if (rc) return rc;
  int devices = 0;
  HIP_TRY(hipGetDeviceCount(&devices));
  if (device < 0 || device >= devices) return fail(PRT_ERR_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  prt_comm* c = new prt_comm();
  c->world = world; c->rank = rank; c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(PRT_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  const size_t words = (size_t)(world + 1) * kGatherMaxLimit;
  if (hipMalloc((void**)&c->dev_counts, words * sizeof(int64_t)) != hipSuccess ||
      hipHostMalloc((void**)&c->host_counts, words * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
    (void)g_rccl.CommDestroy(c->comm);
    if (c->dev_counts) (void)hipFree(c->dev_counts);
    delete c;
    return fail(PRT_ERR_HIP, "communicator buffers");
  }
  *out = c;
  return PRT_OK;
}

// What RCCL itself says about the communicator (not what the caller asked for): out3 = ranks in it (ncclCommCount),
// this process's rank (ncclCommUserRank), the device it was made on.  For run records: a bench line that claims N
// GPUs shows that RCCL saw N ranks.
extern "C" int prt_comm_info(const prt_comm* c, int* out3) {
  if (!c || !out3) return fail(PRT_ERR_ARG, "null argument");
  if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(PRT_ERR_HIP, "librccl lacks ncclCommCount / ncclCommUserRank");
  RCCL_TRY(g_rccl.CommCount(c->comm, &out3[0]));
  RCCL_TRY(g_rccl.CommUserRank(c->comm, &out3[1]));
  out3[2] = c->device;
  return PRT_OK;
}

extern "C" void prt_comm_destroy(prt_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->dev_counts) (void)hipFree(c->dev_counts);
  if (c->host_counts) (void)hipHostFree(c->host_counts);
  delete c;
}

// step 1: counts_all[r * limit + g] = rows rank r recorded in generation g.  Synchronises `stream`.
extern "C" int prt_allgather_counts(prt_comm* c, const int64_t* counts_local, int limit, int64_t* counts_all,
                                    void* stream) {
  if (!c || !counts_local || !counts_all || limit < 1 || limit > kGatherMaxLimit)
    return fail(PRT_ERR_ARG, "bad arguments (generation_limit must be 1..1024)");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  int64_t* mine_h = c->host_counts;
  int64_t* all_h = c->host_counts + kGatherMaxLimit;
  int64_t* mine_d = c->dev_counts;
  int64_t* all_d = c->dev_counts + kGatherMaxLimit;
  std::memcpy(mine_h, counts_local, (size_t)limit * sizeof(int64_t));
  HIP_TRY(hipMemcpyAsync(mine_d, mine_h, (size_t)limit * sizeof(int64_t), hipMemcpyHostToDevice, st));
  RCCL_TRY(g_rccl.AllGather(mine_d, all_d, (size_t)limit, ncclInt64, c->comm, st));
  HIP_TRY(hipMemcpyAsync(all_h, all_d, (size_t)limit * c->world * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::memcpy(counts_all, all_h, (size_t)limit * c->world * sizeof(int64_t));
  return PRT_OK;
}

// ---- placement ----------------------------------------------------------------------------------
// segment s = g * G + r: rows [seg_dst[s], seg_dst[s + 1]) of the assembled frame come from columns
// [seg_src[s], ...) of rank r's block.  staging element (column k, rank r, position p) sits at
// k * stride_col + r * stride_rank + p.
__global__ void __launch_bounds__(PRT_BLOCK)
k_place_rows(const double* __restrict__ staging, int64_t stride_rank, int64_t stride_col, int world,
             int n_seg, const int64_t* __restrict__ seg_dst, const int64_t* __restrict__ seg_src,
             double* __restrict__ out, int64_t ld_out, int64_t total) {
  const int64_t j = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
  if (j >= total) return;
  int lo = 0, hi = n_seg;  // largest s with seg_dst[s] <= j (empty segments share a start: take the last)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_dst[mid] <= j) lo = mid; else hi = mid;
  }
  const int rank = lo % world;
  const double* src = staging + (int64_t)rank * stride_rank + seg_src[lo] + (j - seg_dst[lo]);
#pragma unroll
  for (int k = 0; k < PRT_RECORD_COLS; ++k) out[k * ld_out + j] = src[k * stride_col];
}

struct PlaceTables {
  std::vector<int64_t> host;  // seg_dst[n_seg + 1] | seg_src[n_seg]
  int n_seg = 0;
  int64_t total = 0, widest = 0;
};
static PlaceTables place_tables(const int64_t* counts_all, int world, int limit) {
  PlaceTables t;
  t.n_seg = world * limit;
  t.host.assign((size_t)2 * t.n_seg + 1, 0);
  int64_t* dst = t.host.data();
  int64_t* src = t.host.data() + t.n_seg + 1;
  std::vector<int64_t> local(world, 0);
  int64_t at = 0;
  for (int g = 0; g < limit; ++g)
    for (int r = 0; r < world; ++r) {
      const int64_t count = counts_all[(size_t)r * limit + g];
      dst[g * world + r] = at;
      src[g * world + r] = local[r];
      at += count;
      local[r] += count;
    }
  dst[t.n_seg] = at;
  t.total = at;
  for (int r = 0; r < world; ++r) t.widest = std::max(t.widest, local[r]);
  return t;
}

extern "C" int64_t prt_place_workspace_bytes(int world, int limit) {
  if (world < 1 || limit < 1) return PRT_ERR_ARG;
  return (int64_t)align_up(((size_t)2 * world * limit + 1) * sizeof(int64_t), 256);
}

static int place_rows(int device, const double* staging, int64_t stride_rank, int64_t stride_col, int world,
                      const PlaceTables& t, double* out, int64_t ld_out, void* workspace, hipStream_t st) {
  if (t.total == 0) return PRT_OK;
  int64_t* tables = (int64_t*)workspace;
  // The source of an asynchronous copy has to stay as it is until the copy has run: the tables go through
  // a small ring of page-locked blocks per host thread and device (an event belongs to the device it was
  // made on), each guarded by an event recorded behind its copy (a block is only written again once that
  // event has passed -- any number of calls may be in flight).
  struct Staging {
    int64_t* block = nullptr;
    size_t words = 0;
    hipEvent_t copied = nullptr;
  };
  struct Ring {
    Staging slot[4];
    unsigned turn = 0;
  };
  static thread_local std::unordered_map<int, Ring> rings;
  Ring& ring = rings[device];
  Staging& stage = ring.slot[ring.turn++ % 4];
  if (stage.copied) HIP_TRY(hipEventSynchronize(stage.copied));
  if (stage.words < t.host.size()) {
    if (stage.block) HIP_TRY(hipHostFree(stage.block));
    stage.block = nullptr;
    stage.words = std::max<size_t>(t.host.size(), 1024);
    HIP_TRY(hipHostMalloc((void**)&stage.block, stage.words * sizeof(int64_t), hipHostMallocDefault));
  }
  if (!stage.copied) HIP_TRY(hipEventCreateWithFlags(&stage.copied, hipEventDisableTiming));
  std::memcpy(stage.block, t.host.data(), t.host.size() * sizeof(int64_t));
  HIP_TRY(hipMemcpyAsync(tables, stage.block, t.host.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(stage.copied, st));
  hipLaunchKernelGGL(k_place_rows, dim3(blocks_for(t.total)), dim3(PRT_BLOCK), 0, st, staging, stride_rank,
                     stride_col, world, t.n_seg, (const int64_t*)tables, (const int64_t*)(tables + t.n_seg + 1),
                     out, ld_out, t.total);
  HIP_TRY(hipGetLastError());
  return PRT_OK;
}

// step 3 on its own: the staging area was filled by the caller's transport
extern "C" int prt_place_rows(int device, const double* staging, int64_t stride_rank, int64_t stride_col,
                              int world, const int64_t* counts_all, int limit, double* out, int64_t ld_out,
                              void* workspace, void* stream) {
  if (world < 1 || limit < 1 || !counts_all || !workspace) return fail(PRT_ERR_ARG, "bad arguments");
  int rc = ops_device(device);
  if (rc) return rc;
  const PlaceTables t = place_tables(counts_all, world, limit);
  if (t.total && (!staging || !out || ld_out < t.total)) return fail(PRT_ERR_ARG, "bad buffers");
  return place_rows(device, staging, stride_rank, stride_col, world, t, out, ld_out, workspace, (hipStream_t)stream);
}

// workspace of prt_allgather_rows: tables | staging [15][world][pad]
extern "C" int64_t prt_allgather_workspace_bytes(int world, int limit, int64_t pad_rows) {
  if (world < 1 || limit < 1 || pad_rows < 0) return PRT_ERR_ARG;
  return prt_place_workspace_bytes(world, limit) +
         (int64_t)align_up((size_t)PRT_RECORD_COLS * world * (size_t)std::max<int64_t>(pad_rows, 1) * sizeof(double), 256);
}

// steps 2 + 3, stream-ordered, no host synchronisation.  `rows` is this rank's (15, ld_rows) record
// block; ld_rows must be at least the widest rank's row count (the all-gather sends that many
// elements of every column; what lies beyond this rank's own rows is never placed).
extern "C" int prt_allgather_rows(prt_comm* c, const double* rows, int64_t ld_rows, const int64_t* counts_all,
                                  int limit, double* out, int64_t ld_out, void* workspace, void* stream) {
  if (!c || !counts_all || limit < 1 || !workspace) return fail(PRT_ERR_ARG, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  const PlaceTables t = place_tables(counts_all, c->world, limit);
  if (t.total == 0) return PRT_OK;
  if (!rows || !out || ld_out < t.total || ld_rows < t.widest)
    return fail(PRT_ERR_ARG, "bad buffers (ld_rows must cover the widest rank's rows)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t pad = t.widest;
  double* staging = (double*)((char*)workspace + prt_place_workspace_bytes(c->world, limit));
  RCCL_TRY(g_rccl.GroupStart());
  for (int k = 0; k < PRT_RECORD_COLS; ++k) {
    ncclResult_t r = g_rccl.AllGather(rows + (int64_t)k * ld_rows, staging + (int64_t)k * c->world * pad, (size_t)pad,
                                      ncclDouble, c->comm, st);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return fail(PRT_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
    }
  }
  RCCL_TRY(g_rccl.GroupEnd());
  return place_rows(c->device, staging, pad, (int64_t)c->world * pad, c->world, t, out, ld_out, workspace, st);
}
