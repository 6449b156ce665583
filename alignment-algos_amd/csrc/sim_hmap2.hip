// sim_hmap2.hip — Hmap2Eval / HMAPaliEval similarity matrix on the device (gfx950).
//
// Reference: Hmap2Eval::similarity (hmap2_eval.h:27-39): dot_product of the two 20-value profiles
// (hmath.h:18-26: products then a sequential sum) times exp(alpha * pearson_corr(sse3,sse3) * conf * conf)
// (hmath.h:43-60, :94-103), evaluated for the interior of the SimilarityMatrix whose borders are zero
// (simmatrix.h:51-72); then post_process (hmap2_eval.h:98-101): z-normalise the interior with
// norm_elements (hmath.h:62-79 -> :43-60: SEQUENTIAL fp32 sums of x and x*x in row-major order,
// avg = sum/n, var = sumsq/n - avg*avg, std = sqrt(var), x = (x - avg)/std) and shift by -zero_shift.
//
// Three kernels:
//  hmap2_sim_kernel    one thread per template column, rows in chunks: the column's profile sits in
//                      registers, the row's profile is broadcast from LDS; expf is the host libm's algorithm
//                      (glibc flt-32/e_expf.c: 32-entry 2^(i/32) table, degree-3 polynomial in double) so the
//                      bits match the CPU evaluator;
//  hmap2_stats_kernel  two waves per pair (one per sum); 64 coalesced elements per step, each fp32 sum is
//                      accumulated one element at a time in the reference's order (a parallel reduction would
//                      round differently) by a value that rotates through the lanes;
//  hmap2_apply_kernel  elementwise (x - avg) / std + shift over the interior.
#include <algorithm>
#include <cmath>

#include "aln_device.h"

namespace aln {

__device__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// expf as glibc computes it (sysdeps/ieee754/flt-32/e_expf.c, the table-driven double-precision algorithm);
// validated bit-for-bit against the host libm on 350 k samples.  Outside |x| < 88 fall back to ocml.
__device__ __forceinline__ float expf_glibc(float x) {
  if (!(fabsf(x) < 88.0f)) return expf(x);
  const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
  const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
  const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  double xd = (double)x;
  double z = InvLn2N * xd;
  double kd = z + Shift;
  unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= Shift;
  double r = z - kd;
  unsigned long long t = kExp2fTab[ki % 32];
  t += ki << 47;
  double s = __longlong_as_double((long long)t);
  z = C0 * r + C1;
  double r2 = r * r;
  double y = C2 * r + 1.0;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}

// norm_elements on a 3-vector (hmath.h:43-60)
__device__ __forceinline__ void norm3(const float v[3], float out[3]) {
  float sum = 0.f;
  sum += v[0]; sum += v[1]; sum += v[2];
  float sumsq = 0.f;
  { float s = v[0] * v[0]; sumsq += s; }
  { float s = v[1] * v[1]; sumsq += s; }
  { float s = v[2] * v[2]; sumsq += s; }
  float avg = sum / 3.f;
  float var = sumsq / 3.f - avg * avg;
  float sd = sqrtf(var);
#pragma unroll
  for (int k = 0; k < 3; ++k) { float x = v[k]; x -= avg; x /= sd; out[k] = x; }
}

constexpr int kSimThreads = 256;
constexpr int kSimRows = 32;

__global__ __launch_bounds__(kSimThreads) void hmap2_sim_kernel(const PairDesc* __restrict__ pairs,
                                                                const float* __restrict__ q_aa, const float* __restrict__ q_sse,
                                                                const float* __restrict__ q_conf, const float* __restrict__ t_aa,
                                                                const float* __restrict__ t_sse, const float* __restrict__ t_conf,
                                                                float* __restrict__ Sbase, float alpha) {
  __shared__ float rowp[kSimRows][24];       // aa[20], normalised sse[3], conf
  const PairDesc pd = pairs[blockIdx.z];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int i0 = blockIdx.y * kSimRows;
  const int j = blockIdx.x * kSimThreads + threadIdx.x;
  if (i0 >= Q || blockIdx.x * kSimThreads >= ld) return;
  float* S = Sbase + pd.plane_off;
  // stage the chunk's query rows
  for (int k = threadIdx.x; k < kSimRows * 24; k += kSimThreads) {
    int r = k / 24, c = k % 24, i = i0 + r;
    float v = 0.f;
    if (i < Q) {
      if (c < 20) v = q_aa[(pd.q_off + i) * 20 + c];
      else if (c == 23) v = q_conf[pd.q_off + i];
    }
    rowp[r][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < kSimRows) {
    int i = i0 + threadIdx.x;
    if (i < Q) {
      float v[3] = {q_sse[(pd.q_off + i) * 3 + 0], q_sse[(pd.q_off + i) * 3 + 1], q_sse[(pd.q_off + i) * 3 + 2]}, n[3];
      norm3(v, n);
      rowp[threadIdx.x][20] = n[0]; rowp[threadIdx.x][21] = n[1]; rowp[threadIdx.x][22] = n[2];
    }
  }
  __syncthreads();
  if (j >= ld) return;
  float ta[20], tn[3], tcf = 0.f;
  const bool jin = j >= 1 && j <= T - 2;
  if (jin) {
#pragma unroll
    for (int k = 0; k < 20; ++k) ta[k] = t_aa[(pd.t_off + j) * 20 + k];
    float v[3] = {t_sse[(pd.t_off + j) * 3 + 0], t_sse[(pd.t_off + j) * 3 + 1], t_sse[(pd.t_off + j) * 3 + 2]};
    norm3(v, tn);
    tcf = t_conf[pd.t_off + j];
  }
  for (int r = 0; r < kSimRows; ++r) {
    const int i = i0 + r;
    if (i >= Q) break;
    float sim = 0.f;
    if (jin && i >= 1 && i <= Q - 2) {
      float ip = 0.f;                                  // dot_product, hmath.h:18-26
#pragma unroll
      for (int k = 0; k < 20; ++k) { float p = rowp[r][k] * ta[k]; ip += p; }
      float pc = 0.f;                                  // pearson_corr, hmath.h:94-103
#pragma unroll
      for (int k = 0; k < 3; ++k) { float p = rowp[r][20 + k] * tn[k]; pc += p; }
      pc = pc / 3.f;
      sim = ip * expf_glibc(alpha * pc * rowp[r][23] * tcf);   // hmap2_eval.h:34-37
    }
    S[(size_t)i * ld + j] = sim;                       // borders (and pad columns) are zero, simmatrix.h:58-66
  }
}

// The reference's sums are sequential fp32 chains over the interior in row-major order (hmath.h:43-60): 4 M dependent additions
// per 2000 x 2000 pair and chain, which no reduction tree may reorder.  What can be saved is everything AROUND the additions.
// Two waves per pair, one chain each (wave 0: sum x, wave 1: sum x*x), and the running value travels through the lanes:
//     s = wave_ror:1(s) + x          one v_add_f32_dpp per element
// — after step k the true partial sum sits in lane k mod 64, having just added that lane's own element; all other lanes compute
// values nobody reads.  No v_readlane, no SGPR round trip: the chain is one dependent VALU instruction per element (a lone wave
// issues one every ~2 ns), 8 x 64 elements are in flight while the previous 512 are consumed.
__device__ __forceinline__ float ror1_add(float s, float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), 0x13C, 0xF, 0xF, false)) + x;   // wave_ror:1
}

__global__ __launch_bounds__(128) void hmap2_stats_kernel(const PairDesc* __restrict__ pairs, const float* __restrict__ Sbase,
                                                          float* __restrict__ stats) {
  __shared__ float s_total[2];
  const PairDesc pd = pairs[blockIdx.x];
  const float* S = Sbase + pd.plane_off;
  const int Q = pd.Q, T = pd.T, ld = pd.ld, lane = threadIdx.x & 63;
  const bool squares = (threadIdx.x >> 6) != 0;          // wave-uniform: this wave's chain
  int i0 = 1, i1 = Q - 1, j0 = 1, j1 = T - 1;
  if (i0 >= i1 || j0 >= j1) { i0 = 0; j0 = 0; i1 = Q; j1 = T; }   // hmath.h:65-66
  const int W = j1 - j0, N = (i1 - i0) * W;
  constexpr int kU = 8;
  float v[kU];
  auto fetch = [&](int base) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int n = base + 64 * u + lane;
      v[u] = (n < N) ? S[(size_t)(i0 + n / W) * ld + j0 + n % W] : 0.f;
    }
  };
  float s = 0.f;
  fetch(0);
  for (int base = 0; base < N; base += 64 * kU) {
    float cv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) cv[u] = squares ? v[u] * v[u] : v[u];
    if (base + 64 * kU < N) fetch(base + 64 * kU);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int cnt = min(64, N - (base + 64 * u));
      if (cnt == 64) {
#pragma unroll
        for (int l = 0; l < 64; ++l) s = ror1_add(s, cv[u]);      // element l is added by lane l at step l
      } else {
        for (int l = 0; l < cnt; ++l) s = ror1_add(s, cv[u]);
      }
    }
  }
  // the chain's value sits in the lane that added the last element
  const int last_lane = N > 0 ? (N - 1) & 63 : 0;
  const float mine = __shfl(s, last_lane);
  if (lane == 0) s_total[squares ? 1 : 0] = N > 0 ? mine : 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n = (float)((i1 - i0) * (j1 - j0));
    const float sum = s_total[0], sumsq = s_total[1];
    float avg = sum / n;
    float var = sumsq / n - avg * avg;
    float sd = sqrtf(var);
    stats[2 * blockIdx.x] = avg;
    stats[2 * blockIdx.x + 1] = sd;
  }
}

// Four consecutive elements of a row per thread (rows are 32-byte aligned: ld is a multiple of 8).  The pass also leaves max |S| of
// the pair (every element the DP can read) for the exact-order kernel's rounding margin — one more reduction here instead of
// one more pass over 16 GB there: |x| as an unsigned word orders like the float, a NaN sorts above everything (and then
// nothing is skipped, as it must be).
constexpr int kApplyRows = 16;
__global__ __launch_bounds__(256) void hmap2_apply_kernel(const PairDesc* __restrict__ pairs, float* __restrict__ Sbase,
                                                          const float* __restrict__ stats, float shift, unsigned int* __restrict__ absmax) {
  const PairDesc pd = pairs[blockIdx.z];
  float* S = Sbase + pd.plane_off;
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  int i0 = 1, i1 = Q - 1, j0 = 1, j1 = T - 1;
  const bool whole = (i0 >= i1 || j0 >= j1);
  const float avg = stats[2 * blockIdx.z], sd = stats[2 * blockIdx.z + 1];
  const int jb = (blockIdx.x * 256 + threadIdx.x) * 4;
  unsigned int mx = 0u;
  // kApplyRows rows per workgroup: one atomic per workgroup, not per wave and row (16 M atomics on 1024 words cost 23 ms)
  for (int i = blockIdx.y * kApplyRows; i < (blockIdx.y + 1) * kApplyRows && i < Q; ++i) {
    if (jb >= T) break;
    float4* p4 = reinterpret_cast<float4*>(S + (size_t)i * ld + jb);       // jb + 3 < ld: pad columns are rewritten unchanged
    float4 v4 = *p4;
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + u;
      if (j < T) {
        float x = v[u];
        const bool in_norm = whole || (i >= i0 && i < i1 && j >= j0 && j < j1);
        if (in_norm) { x -= avg; x /= sd; }
        // shift_elements is called with the same (1..rows-1) bounds and applies the same fallback (hmath.h:81-92)
        if (in_norm) x = x + shift;
        v[u] = x;
        const unsigned int a = __float_as_uint(x) & 0x7FFFFFFFu;
        mx = a > mx ? a : mx;
      }
    }
    *p4 = make_float4(v[0], v[1], v[2], v[3]);
  }
  for (int o = 32; o; o >>= 1) { const unsigned int other = (unsigned int)__shfl_xor((int)mx, o); mx = other > mx ? other : mx; }
  __shared__ unsigned int red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int a = red[0] > red[1] ? red[0] : red[1], c = red[2] > red[3] ? red[2] : red[3];
    const unsigned int m = a > c ? a : c;
    if (m) atomicMax(&absmax[blockIdx.z], m);
  }
}

// host entry: profiles (pool order, sentinels included) -> d_S for every pair of the batch
int launch_sim_hmap2(aln_batch* b, const aln_sim* sim) {
  aln_ctx* ctx = b->ctx;
  if (!sim->q_prof.aa || !sim->q_prof.sse || !sim->q_prof.conf || !sim->t_prof.aa || !sim->t_prof.sse || !sim->t_prof.conf) return ALN_E_ARG;
  if (b->n_pairs == 0) return ALN_OK;
  if (!b->d_S) ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_S, (size_t)std::max<int64_t>(b->plane_elems, 1) * 4));
  float *dq_aa = nullptr, *dq_sse = nullptr, *dq_conf = nullptr, *dt_aa = nullptr, *dt_sse = nullptr, *dt_conf = nullptr, *d_stats = nullptr;
  auto cleanup = [&]() { hipFree(dq_aa); hipFree(dq_sse); hipFree(dq_conf); hipFree(dt_aa); hipFree(dt_sse); hipFree(dt_conf); hipFree(d_stats); };
#define HTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return ALN_E_HIP; } } while (0)
  const size_t nq = (size_t)b->q_total, nt = (size_t)b->t_total;
  HTRY(hipMalloc((void**)&dq_aa, nq * 80)); HTRY(hipMalloc((void**)&dq_sse, nq * 12)); HTRY(hipMalloc((void**)&dq_conf, nq * 4));
  HTRY(hipMalloc((void**)&dt_aa, nt * 80)); HTRY(hipMalloc((void**)&dt_sse, nt * 12)); HTRY(hipMalloc((void**)&dt_conf, nt * 4));
  HTRY(hipMalloc((void**)&d_stats, (size_t)std::max(b->n_pairs, 1) * 8));
  HTRY(hipMemcpyAsync(dq_aa, sim->q_prof.aa, nq * 80, hipMemcpyHostToDevice, ctx->stream));
  HTRY(hipMemcpyAsync(dq_sse, sim->q_prof.sse, nq * 12, hipMemcpyHostToDevice, ctx->stream));
  HTRY(hipMemcpyAsync(dq_conf, sim->q_prof.conf, nq * 4, hipMemcpyHostToDevice, ctx->stream));
  HTRY(hipMemcpyAsync(dt_aa, sim->t_prof.aa, nt * 80, hipMemcpyHostToDevice, ctx->stream));
  HTRY(hipMemcpyAsync(dt_sse, sim->t_prof.sse, nt * 12, hipMemcpyHostToDevice, ctx->stream));
  HTRY(hipMemcpyAsync(dt_conf, sim->t_prof.conf, nt * 4, hipMemcpyHostToDevice, ctx->stream));
  const int ldmax = b->maxld;
  dim3 g1((ldmax + kSimThreads - 1) / kSimThreads, (b->maxQ + kSimRows - 1) / kSimRows, b->n_pairs);
  hipLaunchKernelGGL(hmap2_sim_kernel, g1, dim3(kSimThreads), 0, ctx->stream, b->d_pairs, dq_aa, dq_sse, dq_conf, dt_aa, dt_sse,
                     dt_conf, b->d_S, sim->alpha);
  HTRY(hipGetLastError());
  if (sim->normalize) {
    hipLaunchKernelGGL(hmap2_stats_kernel, dim3(b->n_pairs), dim3(128), 0, ctx->stream, b->d_pairs, b->d_S, d_stats);
    HTRY(hipGetLastError());
    if (!b->d_sabs) HTRY(hipMalloc((void**)&b->d_sabs, (size_t)std::max(b->n_pairs, 1) * 4));
    HTRY(hipMemsetAsync(b->d_sabs, 0, (size_t)std::max(b->n_pairs, 1) * 4, ctx->stream));
    dim3 g3((b->maxT + 1023) / 1024, (b->maxQ + kApplyRows - 1) / kApplyRows, b->n_pairs);
    hipLaunchKernelGGL(hmap2_apply_kernel, g3, dim3(256), 0, ctx->stream, b->d_pairs, b->d_S, d_stats, -sim->zero_shift,
                       reinterpret_cast<unsigned int*>(b->d_sabs));
    HTRY(hipGetLastError());
  }
  b->sabs_valid = sim->normalize != 0;
  HTRY(hipStreamSynchronize(ctx->stream));
#undef HTRY
  cleanup();
  return ALN_OK;
}

}  // namespace aln

// Hmap2Eval::pre_calculate (hmap2_eval.cpp:17-25) on the host: per template position gap coefficients from
// p_coil = sse[2].  Uses the host libm's expf exactly like the reference.
extern "C" int aln_hmap2_gap_arrays(const float* t_sse, int64_t n, float gap_init, float gap_extn, float beta, float* t_gap_init,
                                    float* t_gap_extn) {
  if (!t_sse || !t_gap_init || !t_gap_extn || n < 0) return ALN_E_ARG;
  for (int64_t i = 0; i < n; ++i) {
    float Pi = expf(beta * (1.f - 1.25f * t_sse[3 * i + 2]));
    t_gap_init[i] = gap_init * Pi;
    t_gap_extn[i] = gap_extn * Pi;
  }
  return ALN_OK;
}
