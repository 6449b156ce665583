// VALU issue rate against waves per SIMD on gfx950 (tools/, not part of the library): the same dependent-free instruction
// stream (8 chains of v_sub_f32 + v_max_f32 pairs) at 1, 2, 3, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_mix(float* out, int iters) {
  float a[8], b = threadIdx.x * 1.5f + 1.0f;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("v_sub_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_mix_indep(float* out, int iters) {
  float a[8], c[8], b = threadIdx.x * 1.5f + 1.0f;
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; c[i] = i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_sub_f32 %0, %8, %16\n\tv_sub_f32 %1, %9, %16\n\tv_sub_f32 %2, %10, %16\n\tv_sub_f32 %3, %11, %16\n\t"
                   "v_sub_f32 %4, %12, %16\n\tv_sub_f32 %5, %13, %16\n\tv_sub_f32 %6, %14, %16\n\tv_sub_f32 %7, %15, %16\n\t"
                   "v_max_f32 %8, %8, %0\n\tv_max_f32 %9, %9, %1\n\tv_max_f32 %10, %10, %2\n\tv_max_f32 %11, %11, %3\n\t"
                   "v_max_f32 %12, %12, %4\n\tv_max_f32 %13, %13, %5\n\tv_max_f32 %14, %14, %6\n\tv_max_f32 %15, %15, %7"
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                     "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + c[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class K>
void run(const char* name, K kern, int wps) {
  const int blocks = 256 * 4 * wps, iters = 4000;       // one-wave blocks: wps waves on each of the 1024 SIMDs
  float* out; hipMalloc(&out, blocks * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 64>>>(out, 50);
  hipEventRecord(e0); kern<<<blocks, 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_simd = (double)wps * iters * 128;
  printf("%-12s waves/SIMD %d: %7.3f ms  %.2f ns per instruction per SIMD\n", name, wps, ms, ms * 1e6 / per_simd);
  hipFree(out);
}
int main() {
  for (int w = 1; w <= 4; ++w) run("pairs", k_mix, w);
  for (int w = 1; w <= 4; ++w) run("grouped8", k_mix_indep, w);
  return 0;
}
