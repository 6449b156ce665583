// micro-benchmark: integer VALU issue rate on gfx950 with W waves per SIMD (tools/, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
  int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 ^ 5, a3 = a0 + 7, a4 = a0 - 9, a5 = a0 * 5, a6 = a0 ^ 77, a7 = a0 + 1;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) { a0 = max(a0, a1 + i); a1 = max(a1, a2); a2 = max(a2, a3 + 1); a3 = max(a3, a4); a4 = max(a4, a5 + 2); a5 = max(a5, a6); a6 = max(a6, a7 + 3); a7 = max(a7, a0); }
      if (MODE == 1) { a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a4; a4 = a4 + a5; a5 = a5 + a6; a6 = a6 + a7; a7 = a7 + a0; }
      if (MODE == 2) { a0 = (a1 > a2) ? a3 : a0; a1 = (a2 > a3) ? a4 : a1; a2 = (a3 > a4) ? a5 : a2; a3 = (a4 > a5) ? a6 : a3; a4 = (a5 > a6) ? a7 : a4; a5 = (a6 > a7) ? a0 : a5; a6 = (a7 > a0) ? a1 : a6; a7 = (a0 > a1) ? a2 : a7; }
      if (MODE == 3) { f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f4); f3 = fmaf(f3, f4, f5); f4 = fmaf(f4, f5, f6); f5 = fmaf(f5, f6, f7); f6 = fmaf(f6, f7, f0); f7 = fmaf(f7, f0, f1); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}
template <int MODE>
void run(const char* name, int wps, int ops_per_u) {
  int blocks = 256 * wps;   // 256 CUs x wps blocks of 4 waves => wps waves per SIMD
  int* out; hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1);
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winst = (double)blocks * 4 * iters * 8 * ops_per_u;     // wave-instructions
  double per_simd = winst / 1024.0;
  printf("%s waves/SIMD=%d: %.2f ms, %.3f Gwave-instr/s per SIMD -> %.2f cycles/instr @2.4GHz\n", name, wps, ms, per_simd / ms / 1e6, 2.4e9 * ms * 1e-3 / per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("max+add(1.5op)", w, 12); run<1>("add", w, 8); run<2>("cmp+cndmask", w, 16); run<3>("fma_f32", w, 8); }
  return 0;
}
