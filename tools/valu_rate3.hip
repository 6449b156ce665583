// packed-fp32 / scalar-operand VALU issue rates on gfx950 (tools/, not part of the library); same harness as valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEF_KERNEL64(NAME, ASM)                                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(double* out, int iters) {                                   \
    double a[8], b = threadIdx.x * 1.5 + 1.0;                                                                  \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                       \
    for (int it = 0; it < iters; ++it) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(ASM : "+v"(a[r]) : "v"(b));                \
      }                                                                                                       \
    }                                                                                                         \
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                  \
  }
#define DEF_KERNEL32(NAME, ASM)                                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(double* out, int iters) {                                   \
    float a[8], b = threadIdx.x * 1.5f + 1.0f; float sc = (float)iters;                                        \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                       \
    for (int it = 0; it < iters; ++it) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(ASM : "+v"(a[r]) : "v"(b), "s"(sc));       \
      }                                                                                                       \
    }                                                                                                         \
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                                      \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                  \
  }
DEF_KERNEL64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEF_KERNEL64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
DEF_KERNEL64(add_f64, "v_add_f64 %0, %0, %1")
DEF_KERNEL32(min_f32, "v_min_f32 %0, %0, %1")
DEF_KERNEL32(sub_f32_s, "v_sub_f32 %0, %2, %0")
DEF_KERNEL32(min_f32_s, "v_min_f32 %0, %2, %0")
DEF_KERNEL32(max_f32_s, "v_max_f32 %0, %2, %0")
DEF_KERNEL32(add_f32_c, "v_add_f32 %0, -1.0, %0")
DEF_KERNEL32(cmp_gt_f32, "v_cmp_gt_f32 vcc, %0, %1")
DEF_KERNEL32(readlane, "v_readlane_b32 s20, %0, 3")
DEF_KERNEL32(fmac_f32, "v_fmac_f32 %0, %1, %1")
DEF_KERNEL32(max_f32_e64, "v_max_f32_e64 %0, %0, %1")
template <class K>
void run(const char* name, K kern, int ninstr) {
  const int wps = 4, blocks = 256 * wps, iters = 4000;
  double* out; hipMalloc(&out, blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 50);
  hipEventRecord(e0); kern<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_simd = (double)wps * iters * 64 * ninstr;
  printf("%-14s %7.3f ms  %.2f ns/instr/SIMD  (%.2f cycles @2.1GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
  hipFree(out);
}
#define RUN(N, C) run(#N, k_##N, C)
int main() {
  RUN(pk_add_f32, 1); RUN(pk_mul_f32, 1); RUN(pk_fma_f32, 1); RUN(add_f64, 1); RUN(min_f32, 1); RUN(sub_f32_s, 1); RUN(min_f32_s, 1);
  RUN(max_f32_s, 1); RUN(add_f32_c, 1); RUN(cmp_gt_f32, 1); RUN(readlane, 1); RUN(fmac_f32, 1); RUN(max_f32_e64, 1);
  return 0;
}
