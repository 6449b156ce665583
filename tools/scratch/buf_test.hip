#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* p, int ld, int rows, int mode) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, mode == 0 ? ld * 2 : ld * 2 * rows, 0x00020000);
  for (int i = 0; i < rows; ++i) {
    u2 v = {(unsigned)(i + 1) * 0x10001u, (unsigned)(i + 1) * 0x10001u};
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, threadIdx.x * 8, i * ld * 2, 0);
  }
}
int main() {
  const int ld = 12, rows = 4;
  unsigned short* d; hipMalloc(&d, 4096); 
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(d, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ld, rows, mode);
    std::vector<unsigned short> h(2048); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    printf("mode %d:", mode);
    for (int i = 0; i < 80; ++i) printf(" %d", h[i]);
    printf("\n");
  }
  return 0;
}
