// store_pattern.hip — what HBM takes from the config-2 DP kernel's WRITE PATTERN alone (gfx950), no recurrence.
//
// The tagged kernel (csrc/dp_affine_tag.hip, NW=2,R=2,X=8) writes, per pair, two uint16 planes of 2002 rows x 2008 columns; per row
// each of the pair's two waves issues four 16-byte-per-lane stores (two planes x two 512-column groups), i.e. four contiguous 1 KB
// segments; 2048 waves (1024 pairs) do this in loose lock step, two waves per SIMD.  This program replays that pattern with
//   V  dependent VALU instructions per row and wave in front of the stores (0 = stores back to back; the kernel issues ~300),
//   B  = 1: the pair's waves meet at an LDS-only barrier every row, as in the kernel,
// and compares it with a streaming fill of the same bytes (every wave writes its own contiguous region, 4 KB per iteration) and with
// an interleaved layout (one 4 B/cell plane: two contiguous 2 KB segments per row and wave).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/scratch/store_pattern tools/scratch/store_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 2002, kLd = 2008, kPairs = 1024;

template <int V>
__device__ __forceinline__ uint32_t spin(uint32_t x, uint32_t y) {
#pragma unroll
  for (int k = 0; k < V; ++k) x = __builtin_amdgcn_perm(x, y + k, 0x05040100u) + 0x9e3779b9u;   // dependent chain, full rate
  return x;
}

// mode 0: the kernel's two-plane pattern.  mode 1: one interleaved 4 B/cell plane.  mode 2: streaming fill (per-wave contiguous).
template <int V, int B, int MODE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void pattern(uint16_t* __restrict__ P, uint16_t* __restrict__ H, int rows) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t plane = (size_t)kRows * kLd;
  uint32_t acc = threadIdx.x * 2654435761u + blockIdx.x;
  uint16_t* p = P + blockIdx.x * plane;
  uint16_t* h = H + blockIdx.x * plane;
  for (int i = 0; i < rows; ++i) {
    acc = spin<V>(acc, (uint32_t)i);
    const u32x4 v = {acc, acc ^ 1u, acc ^ 2u, acc ^ 3u};
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int c = w * 1024 + r * 512 + lane * 8;
        if (c + 8 <= kLd) {
          *reinterpret_cast<u32x4*>(h + (size_t)i * kLd + c) = v;
          *reinterpret_cast<u32x4*>(p + (size_t)i * kLd + c) = v;
        }
      }
    } else if (MODE == 1) {
      uint32_t* q = reinterpret_cast<uint32_t*>(P) + blockIdx.x * plane;      // P holds 2 planes' worth (caller passes one 4 B/cell buffer)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int c = w * 1024 + r * 512 + lane * 8;
        if (c + 8 <= kLd) {
          *reinterpret_cast<u32x4*>(q + (size_t)i * kLd + c) = v;
          *reinterpret_cast<u32x4*>(q + (size_t)i * kLd + c + 4) = v;
        }
      }
    } else {
      // the same bytes per wave and iteration (4 x 1 KB), but each wave owns one contiguous region
      const size_t per_wave = (size_t)rows * 4096;
      char* base = reinterpret_cast<char*>(P) + ((size_t)blockIdx.x * 2 + w) * per_wave + (size_t)i * 4096;
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<u32x4*>(base + r * 1024 + lane * 16) = v;
    }
    if (B) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

template <int V, int B, int MODE>
static void run(const char* what, uint16_t* P, uint16_t* H, hipStream_t st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((pattern<V, B, MODE>), dim3(kPairs), dim3(128), 0, st, P, H, kRows);
  CK(hipEventRecord(e0, st));
  const int reps = 20;
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((pattern<V, B, MODE>), dim3(kPairs), dim3(128), 0, st, P, H, kRows);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = MODE == 2 ? (double)kPairs * 2 * kRows * 4096 : (double)kPairs * kRows * kLd * 4;
  printf("%-58s V=%3d barrier=%d: %.3f ms  %.0f GB/s\n", what, V, B, ms, bytes / ms * 1e-6);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  const size_t plane = (size_t)kRows * kLd;
  const size_t bytes = (size_t)kPairs * plane * 2;        // one uint16 plane of the whole batch
  uint16_t *P, *H;
  const size_t fill_bytes = (size_t)kPairs * 2 * kRows * 4096;
  CK(hipMalloc(&P, (fill_bytes > bytes * 2 ? fill_bytes : bytes * 2) + (1 << 20)));   // modes 1 / 2 write through P alone
  CK(hipMalloc(&H, bytes));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  {   // hipMemsetAsync as the box's own streaming-store yardstick
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemsetAsync(P, 0, bytes * 2, st));
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < 10; ++k) CK(hipMemsetAsync(P, 0, bytes * 2, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemsetAsync of %.2f GB: %.3f ms  %.0f GB/s\n", bytes * 2e-9, ms / 10, bytes * 2.0 / (ms / 10) * 1e-6);
  }
  run<0, 0, 2>("streaming fill, per-wave contiguous", P, H, st);
  run<0, 0, 0>("kernel pattern (2 planes, 1 KB segments)", P, H, st);
  run<0, 1, 0>("kernel pattern", P, H, st);
  run<0, 0, 1>("interleaved 4 B/cell plane (2 KB segments)", P, H, st);
  run<0, 1, 1>("interleaved 4 B/cell plane", P, H, st);
  run<100, 1, 0>("kernel pattern", P, H, st);
  run<200, 1, 0>("kernel pattern", P, H, st);
  run<300, 1, 0>("kernel pattern", P, H, st);
  run<300, 0, 0>("kernel pattern", P, H, st);
  run<300, 1, 1>("interleaved 4 B/cell plane", P, H, st);
  run<400, 1, 0>("kernel pattern", P, H, st);
  run<300, 0, 2>("streaming fill", P, H, st);
  CK(hipFree(P)); CK(hipFree(H));
  return 0;
}
