// per-instruction VALU issue rate on gfx950 (tools/, not part of the library): 8 independent register chains,
// 4 waves per SIMD, inline asm so the compiler cannot rewrite the instruction under test.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define DEF_KERNEL(NAME, ASM2, ASM3)                                                                         \
  __global__ __launch_bounds__(256) void k_##NAME(int* out, int iters) {                                      \
    int a[8], b = threadIdx.x | 1, c = threadIdx.x * 3 + 1;                                                   \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                                       \
    for (int it = 0; it < iters; ++it) {                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                       \
          if (ASM3[0]) asm volatile(ASM3 : "+v"(a[r]) : "v"(b), "v"(c));                                      \
          else asm volatile(ASM2 : "+v"(a[r]) : "v"(b));                                                      \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    int s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                  \
  }
DEF_KERNEL(add_u32, "v_add_u32 %0, %0, %1", "")
DEF_KERNEL(sub_u32, "v_sub_u32 %0, %0, %1", "")
DEF_KERNEL(max_i32, "v_max_i32 %0, %0, %1", "")
DEF_KERNEL(max_u32, "v_max_u32 %0, %0, %1", "")
DEF_KERNEL(and_b32, "v_and_b32 %0, %0, %1", "")
DEF_KERNEL(or_b32, "v_or_b32 %0, %0, %1", "")
DEF_KERNEL(lshlrev, "v_lshlrev_b32 %0, 1, %0", "")
DEF_KERNEL(ashrrev, "v_ashrrev_i32 %0, 1, %0", "")
DEF_KERNEL(mov_b32, "v_mov_b32 %0, %1", "")
DEF_KERNEL(add_f32, "v_add_f32 %0, %0, %1", "")
DEF_KERNEL(max_f32, "v_max_f32 %0, %0, %1", "")
DEF_KERNEL(mul_f32, "v_mul_f32 %0, %0, %1", "")
DEF_KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0", "")
DEF_KERNEL(max3_i32, "", "v_max3_i32 %0, %0, %1, %2")
DEF_KERNEL(max3_f32, "", "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL(add3_u32, "", "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(lshl_add, "", "v_lshl_add_u32 %0, %0, 3, %2")
DEF_KERNEL(lshl_or, "", "v_lshl_or_b32 %0, %0, 3, %2")
DEF_KERNEL(and_or, "", "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(bfe_u32, "", "v_bfe_u32 %0, %0, 3, 8")
DEF_KERNEL(fma_f32, "", "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(mad_i32_i24, "", "v_mad_i32_i24 %0, %0, %1, %2")
DEF_KERNEL(perm_b32, "", "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(cmp_cnd, "", "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL(cmp_only, "v_cmp_gt_i32 vcc, %0, %1", "")
DEF_KERNEL(cnd_only, "", "v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL(cmp_e64, "v_cmp_gt_i32 s[20:21], %0, %1", "")
DEF_KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1", "")
DEF_KERNEL(pk_add_i16, "v_pk_add_i16 %0, %0, %1", "")
DEF_KERNEL(pk_add_f32x, "", "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2")
DEF_KERNEL(max_i32_dpp, "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "")
DEF_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", "")
DEF_KERNEL(min_max, "", "v_max_i32 %0, %0, %1\n v_min_i32 %0, %0, %2")
DEF_KERNEL(med3_i32, "", "v_med3_i32 %0, %0, %1, %2")
DEF_KERNEL(sat_sub, "v_sub_u32 %0, %0, %1 clamp", "")
template <class K>
void run(const char* name, K kern, int ninstr) {
  const int wps = 4, blocks = 256 * wps, iters = 4000;
  int* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, 50);
  hipEventRecord(e0); kern<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_simd = (double)wps * iters * 64 * ninstr;   // wave-instructions per SIMD
  printf("%-14s %7.3f ms  %.2f ns/instr/SIMD  (%.2f cycles @2.1GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e-3 * 2.1e9 / per_simd);
  hipFree(out);
}
#define RUN(N, C) run(#N, k_##N, C)
int main() {
  RUN(add_u32, 1); RUN(sub_u32, 1); RUN(max_i32, 1); RUN(max_u32, 1); RUN(and_b32, 1); RUN(or_b32, 1); RUN(lshlrev, 1); RUN(ashrrev, 1);
  RUN(mov_b32, 1); RUN(add_f32, 1); RUN(max_f32, 1); RUN(mul_f32, 1); RUN(cvt_f32_i32, 1); RUN(max3_i32, 1); RUN(max3_f32, 1);
  RUN(add3_u32, 1); RUN(lshl_add, 1); RUN(lshl_or, 1); RUN(and_or, 1); RUN(bfe_u32, 1); RUN(fma_f32, 1); RUN(mad_i32_i24, 1);
  RUN(perm_b32, 1); RUN(cmp_cnd, 2); RUN(cmp_only, 1); RUN(cnd_only, 1); RUN(cmp_e64, 1); RUN(pk_max_i16, 1); RUN(pk_add_i16, 1);
  RUN(pk_add_f32x, 2); RUN(max_i32_dpp, 1); RUN(mov_dpp, 1); RUN(min_max, 2); RUN(med3_i32, 1); RUN(sat_sub, 1);
  return 0;
}
