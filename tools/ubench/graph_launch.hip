// Host-visible latency of one "trace-shaped" batch of work -- event record, three dependent short
// kernels, event record, one trailing kernel, the last of the three publishing to host-mapped memory --
// submitted (a) as six stream calls, (b) as one hipGraphLaunch of the same sequence captured once.
// The kernels spin for a given time so that the batch is launch-bound (10 us each) or not (60 us each).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, volatile unsigned long long* flag, const unsigned long long* epoch) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  if (flag && threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    *flag = *epoch;
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  unsigned long long *flag, *flag_dev, *epoch, *epoch_dev;
  CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostMalloc((void**)&epoch, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer((void**)&flag_dev, flag, 0));
  CHECK(hipHostGetDevicePointer((void**)&epoch_dev, epoch, 0));
  *flag = 0;
  for (double kernel_us : {10.0, 60.0}) {
    const unsigned long long ticks = (unsigned long long)(kernel_us * 100.0);  // 100 MHz clock
    auto submit = [&]() {
      CHECK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, (volatile unsigned long long*)nullptr, epoch_dev);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, (volatile unsigned long long*)nullptr, epoch_dev);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, (volatile unsigned long long*)flag_dev, epoch_dev);
      CHECK(hipEventRecord(e1, st));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 100ull, (volatile unsigned long long*)nullptr, epoch_dev);
    };
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    submit();
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
      double total = 0, submit_total = 0;
      const int reps = 300;
      for (int r = -20; r < reps; ++r) {
        const unsigned long long want = *epoch + 1;
        const double t0 = now_us();
        *epoch = want;  // (the kernels read it from host-mapped memory: nothing in the graph changes)
        if (mode == 0) submit(); else CHECK(hipGraphLaunch(exec, st));
        const double t1 = now_us();
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) __builtin_ia32_pause();
        const double t2 = now_us();
        CHECK(hipStreamSynchronize(st));
        if (r >= 0) { total += t2 - t0; submit_total += t1 - t0; }
      }
      printf("3 x %.0f us kernels, %s: host sees the result after %.1f us (submission calls return after %.1f us)\n",
             kernel_us, mode == 0 ? "six stream calls " : "one hipGraphLaunch", total / reps, submit_total / reps);
    }
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
  }
  return 0;
}
