// hwid_probe.hip — where does the dispatcher put the 4 waves of a 256-thread workgroup when 4 such workgroups share a CU?
// (dp_exact_tiled_kernel's shape: 256 threads, 128 VGPRs, ~20 KB LDS.)  Prints, for the first workgroups, each wave's
// (XCC, SE, CU, SIMD, wave slot).  build: hipcc --offload-arch=gfx950 -O2 -o tools/scratch/hwid_probe tools/scratch/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 4) void probe(unsigned* out, int spin) {
  __shared__ float pad[5000];
  pad[threadIdx.x] = 1.f;
  __syncthreads();
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = pad[(threadIdx.x + i) & 255];
  for (int s = 0; s < spin; ++s)
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) t += acc[i];
  if ((threadIdx.x & 63) == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); // XCC_ID
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc | (t > 1e30f ? 1u << 31 : 0u);
  }
}
int main() {
  const int n = 1024;
  unsigned* d; hipMalloc(&d, n * 4 * 2 * 4);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(n * 8);
  hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
  int simd_of_wave[4][4] = {};
  for (int b = 0; b < n; ++b)
    for (int w = 0; w < 4; ++w) {
      unsigned hw = h[(b * 4 + w) * 2];
      int wave_id = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      simd_of_wave[w][simd]++;
      if (b < 12) printf("wg %d wave %d: xcc %u se %d sh %d cu %d simd %d slot %d\n", b, w, h[(b * 4 + w) * 2 + 1] & 15, se, sh, cu, simd, wave_id);
    }
  for (int w = 0; w < 4; ++w) printf("wave %d -> simd counts %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
  return 0;
}
