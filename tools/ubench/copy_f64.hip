// calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE with this project's access pattern:
// every lane loads one float64 (8 B) from each of R rows and stores one float64 to each of W
// rows (row-major SoA, coalesced 512 B per wave access).  Known bytes: n*8*R read, n*8*W written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) copy_rows(const double* __restrict__ in, double* __restrict__ out,
                                                  long n, int rows_in, int rows_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0;
  for (int r = 0; r < rows_in; ++r) acc += in[r * n + i];
  for (int r = 0; r < rows_out; ++r) out[r * n + i] = acc + r;
}
int main(int argc, char** argv) {
  const long n = 1000000;
  const int R = 13, W = 28;
  double *in, *out;
  hipMalloc(&in, n * R * 8); hipMalloc(&out, n * W * 8);
  hipMemset(in, 0, n * R * 8);
  for (int k = 0; k < 4; ++k) copy_rows<<<(n + 255) / 256, 256>>>(in, out, n, R, W);
  hipDeviceSynchronize();
  printf("known bytes per launch: read %ld write %ld\n", n * R * 8, n * W * 8);
  return 0;
}
