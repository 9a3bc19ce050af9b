// micro-benchmark: issue cost (cycles per wave-instruction at 8 waves/SIMD) of the VALU instructions
// the generation kernel is made of -- the price list behind the "instruction diet" in DESIGN.md.
// Eight independent register chains per instruction kind, so dependent latency does not show.
// hipcc --offload-arch=gfx950 -O3 valu_costs.hip -o valu_costs && ./valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
  double a[8], b = seed + threadIdx.x * 1e-9, c = 0.5;
  int m[8];
  for (int j = 0; j < 8; ++j) { a[j] = 1.0 + j * 0.125 + threadIdx.x * 1e-7; m[j] = threadIdx.x + j; }
  unsigned long long vcc_dummy = 0;
  for (int it = 0; it < iters; ++it) {
#define FMA(j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#define MUL(j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define ADD(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define MINF(j) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));
#define RCP(j) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[j]));
#define RSQ(j) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[j]));
#define SQRT(j) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[j]));
#define FIXUP(j) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#define FMAS(j) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc");
#define SCALE(j) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c) : "vcc");
#define LDEXP(j) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[j]) : "v"(m[j]));
#define FREXP(j) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(m[j]) : "v"(a[j]));
#define CMP(j) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[j]), "v"(b) : "vcc");
#define CMPCLASS(j) asm volatile("v_cmp_class_f64 vcc, %0, %1" :: "v"(a[j]), "v"(m[j]) : "vcc");
#define CND(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(m[j]) : "v"(m[(j + 1) & 7]) : "vcc");
#define MOV32(j) asm volatile("v_mov_b32 %0, %1" : "=v"(m[j]) : "v"(m[(j + 1) & 7]));
#define MOV64(j) asm volatile("v_mov_b64 %0, %1" : "=v"(a[j]) : "v"(a[(j + 1) & 7]));
#define ADD32(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(m[j]) : "v"(m[(j + 1) & 7]));
#define CMP32(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(m[j]), "v"(m[(j + 1) & 7]) : "vcc");
#define BFE(j) asm volatile("v_bfe_u32 %0, %1, 20, 11" : "=v"(m[j]) : "v"(m[(j + 1) & 7]));
#define PKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
    if (OP == 0) { REP8(FMA) }
    if (OP == 1) { REP8(MUL) }
    if (OP == 2) { REP8(ADD) }
    if (OP == 3) { REP8(MINF) }
    if (OP == 4) { REP8(RCP) }
    if (OP == 5) { REP8(RSQ) }
    if (OP == 6) { REP8(SQRT) }
    if (OP == 7) { REP8(FIXUP) }
    if (OP == 8) { REP8(FMAS) }
    if (OP == 9) { REP8(SCALE) }
    if (OP == 10) { REP8(LDEXP) }
    if (OP == 11) { REP8(FREXP) }
    if (OP == 12) { REP8(CMP) }
    if (OP == 13) { REP8(CMPCLASS) }
    if (OP == 14) { REP8(CND) }
    if (OP == 15) { REP8(MOV32) }
    if (OP == 16) { REP8(MOV64) }
    if (OP == 17) { REP8(ADD32) }
    if (OP == 18) { REP8(CMP32) }
    if (OP == 19) { REP8(BFE) }
    if (OP == 20) { REP8(PKFMA) }
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += a[j] + m[j];
  out[blockIdx.x * 256 + threadIdx.x] = s + (double)vcc_dummy;
}

template <int OP>
int run(const char* name, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  double* out;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, 10, 1.000001);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<OP><<<blocks, 256>>>(out, iters, 1.000001);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_ops_per_simd = blocks * 4.0 / 1024.0 * iters * 8.0;
  printf("%-24s %d waves/SIMD %8.3f ms  %6.2f cycles per wave-instruction (at 2.4 GHz)\n", name, blocks_per_cu, ms,
         ms * 1e-3 * 2.4e9 / wave_ops_per_simd);
  hipFree(out);
  return 0;
}

#define RUN(OP, NAME) run<OP>(NAME, 8); run<OP>(NAME, 1);
int main() {
  RUN(0, "v_fma_f64") RUN(1, "v_mul_f64") RUN(2, "v_add_f64") RUN(3, "v_min_f64") RUN(4, "v_rcp_f64")
  RUN(5, "v_rsq_f64") RUN(6, "v_sqrt_f64") RUN(7, "v_div_fixup_f64") RUN(8, "v_div_fmas_f64")
  RUN(9, "v_div_scale_f64") RUN(10, "v_ldexp_f64") RUN(11, "v_frexp_exp_i32_f64") RUN(12, "v_cmp_lt_f64")
  RUN(13, "v_cmp_class_f64") RUN(14, "v_cndmask_b32") RUN(15, "v_mov_b32") RUN(16, "v_mov_b64")
  RUN(17, "v_add_u32") RUN(18, "v_cmp_lt_u32") RUN(19, "v_bfe_u32") RUN(20, "v_pk_fma_f32")
  return 0;
}
