// dp_exact_blocked.hip — the exact-order O(n^3) DP of dp_exact.hip, restructured for gfx950 VALU throughput.
//
// Same results, bit for bit, as the literal builders of the reference (dpmatrix.h:356-1030): every candidate is
//   s = D[pred]; s -= gap; s += S[i][j]; s = max(0,s) (local); if (s > opt_s) take it
// in the order match, deletions (k ascending), insertions (k ascending).  What makes the literal loop slow is that the
// add of S, the clip and the compare-and-select sit inside the k loops.  They do not have to:
//
//   * fp32 rounding and the clip are monotone, so  max_k clip(fl(fl(D_k - g_k) + S)) = clip(fl(max_k fl(D_k - g_k) + S)):
//     the scans only need d_k = D_k - g_k (one rounded subtract, exactly the reference's `s -= gap`) and a running max;
//   * the winning k ("first strictly greater") is the first k whose literal s_k equals that maximum.  The scans
//     remember, per cell, the first CHUNK of k that attains max d (and the best d of all earlier chunks, `e`); only the
//     cell whose winner is a gap re-walks that one chunk literally (or everything before it when fl(e + S) rounds to
//     the same score — a rounding tie, rare).
//
// Deletion scan (row a-1 -> row a; gap = min(t[k],t[b]) coefficients, hmap2_eval.h:41-67, or constants,
// aasubalib.h:27-51): a thread owns NS columns b = 1 + tid + 256 j and walks k once for all of them, so one broadcast
// LDS read of (D[a-1][k], tgi[k], tge[k]) feeds up to NS x (min, min, mul, add, sub, max) — 7 VALU ops per candidate
// against ~12 + an LDS read in the literal form.
//
// Insertion scan (column b-1, rows k < a-1; coefficients depend on b only): rows are processed in blocks of 16.  For a
// block starting at a0 the candidates k <= a0-2 ("far") are evaluated for all 16 rows of the block at once: each
// D[k][b-1] is loaded ONCE per block and the gap values G_b(n), n = a-k-2, form a window that slides by one per k
// (one new mul+add per k, 16 x (sub, max)) — 2.2 ops per candidate and 16x fewer loads than a per-row column walk.
// Results (max, e, chunk) wait in a per-workgroup scratch.  The <= 15 "near" candidates (rows of the current block) are
// evaluated literally per cell.
//
// One workgroup (256 threads) per pair, rows in order, one barrier per row.  Frames (reverse builds, sub-rectangles) as
// in dp_exact.hip: per-position arrays are staged in LDS in frame order, only plane addresses see real coordinates.
#include "aln_device.h"

#include <cstdlib>

namespace aln {

constexpr int kBT = 256;       // threads per pair
constexpr int kBR = 16;        // rows per block = window of the far-insertion scan = its chunk size
constexpr int kBC = 32;        // deletion-scan chunk
constexpr int kBPad = 64;

__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Four independent (sub, max) pairs issued as 4 subs then 4 maxes.  Left to itself the register allocator reuses one
// temporary for every pair, which makes each v_max wait for the v_sub right before it; with only 2 waves per SIMD that
// dependency stall is not hidden.
__device__ __forceinline__ void submax4_vv(float& c0, float& c1, float& c2, float& c3, float x, float w0, float w1, float w2, float w3) {
  float t0, t1, t2, t3;
  asm("v_sub_f32 %0, %8, %9\n\tv_sub_f32 %1, %8, %10\n\tv_sub_f32 %2, %8, %11\n\tv_sub_f32 %3, %8, %12\n\t"
      "v_max_f32 %4, %4, %0\n\tv_max_f32 %5, %5, %1\n\tv_max_f32 %6, %6, %2\n\tv_max_f32 %7, %7, %3"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
      : "v"(x), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
__device__ __forceinline__ void submax4_sv(float& c0, float& c1, float& c2, float& c3, float s0, float s1, float s2, float s3, float g) {
  float t0, t1, t2, t3;
  asm("v_sub_f32 %0, %8, %12\n\tv_sub_f32 %1, %9, %12\n\tv_sub_f32 %2, %10, %12\n\tv_sub_f32 %3, %11, %12\n\t"
      "v_max_f32 %4, %4, %0\n\tv_max_f32 %5, %5, %1\n\tv_max_f32 %6, %6, %2\n\tv_max_f32 %7, %7, %3"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
      : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(g));
}
// Two candidates per row at once: (sub, sub, max3) instead of 2 x (sub, max) — v_max_f32 and v_max3_f32 both issue at half rate on
// gfx950, so the second maximum is free.  Four rows per statement, subtractions first (see submax4_vv).
__device__ __forceinline__ void submax3x4_vv(float& c0, float& c1, float& c2, float& c3, float xa, float a0, float a1, float a2, float a3,
                                             float xb, float b0, float b1, float b2, float b3) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm("v_sub_f32 %0, %12, %13\n\tv_sub_f32 %1, %12, %14\n\tv_sub_f32 %2, %12, %15\n\tv_sub_f32 %3, %12, %16\n\t"
      "v_sub_f32 %4, %17, %18\n\tv_sub_f32 %5, %17, %19\n\tv_sub_f32 %6, %17, %20\n\tv_sub_f32 %7, %17, %21\n\t"
      "v_max3_f32 %8, %8, %0, %4\n\tv_max3_f32 %9, %9, %1, %5\n\tv_max3_f32 %10, %10, %2, %6\n\tv_max3_f32 %11, %11, %3, %7"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
      : "v"(xa), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(xb), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
__device__ __forceinline__ void submax3x4_sv(float& c0, float& c1, float& c2, float& c3, float sa0, float sa1, float sa2, float sa3, float ga,
                                             float sb0, float sb1, float sb2, float sb3, float gb) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm("v_sub_f32 %0, %12, %16\n\tv_sub_f32 %1, %13, %16\n\tv_sub_f32 %2, %14, %16\n\tv_sub_f32 %3, %15, %16\n\t"
      "v_sub_f32 %4, %17, %21\n\tv_sub_f32 %5, %18, %21\n\tv_sub_f32 %6, %19, %21\n\tv_sub_f32 %7, %20, %21\n\t"
      "v_max3_f32 %8, %8, %0, %4\n\tv_max3_f32 %9, %9, %1, %5\n\tv_max3_f32 %10, %10, %2, %6\n\tv_max3_f32 %11, %11, %3, %7"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
      : "s"(sa0), "s"(sa1), "s"(sa2), "s"(sa3), "v"(ga), "s"(sb0), "s"(sb1), "s"(sb2), "s"(sb3), "v"(gb));
}
// loads served by L2 (the planes are written by other threads of this workgroup; L1 lines may predate those writes)
__device__ __forceinline__ float aload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int aloadi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NS>
struct ScanState {
  float gib[NS], geb[NS];   // this column's own coefficients (TPOS) — unused for constant gaps
  float fd[NS];             // (float)(b - k - 2) for the current k
  float cm[NS], m[NS], e[NS];
  int cidx[NS];
};

// the value N lanes down the same row of 16 lanes (-inf where the row ends)
template <int N>
__device__ __forceinline__ float row_shr_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xFF800000, __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float vmax3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// k in [k0, k1) (both multiples of kBC), slots JJ..NS-1 active; MASK: slot JJ is in its triangular tail (fd < 0 = beyond b-2).
// The candidate chain (min, min, mul, add, sub, max) is evaluated STAGE by stage over several independent candidates with
// scheduling barriers in between: written chain by chain the compiler issues each chain back to back, and with 2 waves per
// SIMD the dependent-issue latency of those 6 instructions is what the loop then runs at.
// EXM (with MASK, one slot): the lanes' columns are consecutive, so "k <= b - 2" is "lane >= k - k0 + 1" — a mask known without
// a comparison.  The candidate's maximum is then taken under that mask in EXEC (s_and_b64 + v_max_f32) instead of
// (v_cmp, v_cndmask, half a v_max3): 1 half-rate VALU operation per candidate instead of 2.5.
__device__ __forceinline__ unsigned long long lanes_from(int t) { return t >= 64 ? 0ull : (~0ull << (t < 0 ? 0 : t)); }
__device__ __forceinline__ void max4_under_masks(float& c, float d0, float d1, float d2, float d3, unsigned long long m0, unsigned long long m1,
                                                 unsigned long long m2, unsigned long long m3) {
  unsigned long long sv;
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               "s_and_b64 exec, %[sv], %[m0]\n\tv_max_f32 %[c], %[c], %[d0]\n\t"
               "s_and_b64 exec, %[sv], %[m1]\n\tv_max_f32 %[c], %[c], %[d1]\n\t"
               "s_and_b64 exec, %[sv], %[m2]\n\tv_max_f32 %[c], %[c], %[d2]\n\t"
               "s_and_b64 exec, %[sv], %[m3]\n\tv_max_f32 %[c], %[c], %[d3]\n\t"
               "s_mov_b64 exec, %[sv]"
               : [c] "+v"(c), [sv] "=&s"(sv)
               : [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2), [d3] "v"(d3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
               : "scc");
}
template <int JJ, int NS, bool TPOS, bool MASK, bool EXM = false>
__device__ __forceinline__ void scan_range(ScanState<NS>& s, const float* __restrict__ prev, const float2* __restrict__ tg, int k0,
                                           int k1, float gi_c, float ge_c) {
  const float ninf = -__builtin_inff();
  constexpr int NA = NS - JJ;
  constexpr int UNR = (NS == 1) ? 2 : kBC / 4;   // one-slot kernels: a short body keeps the register count down
  for (int kc = k0; kc < k1; kc += kBC) {
#pragma unroll UNR
    for (int u4 = 0; u4 < kBC; u4 += 4) {
      const float4 p4 = *reinterpret_cast<const float4*>(prev + kc + u4);
      float4 ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
      if (TPOS) {
        ga = *reinterpret_cast<const float4*>(tg + kc + u4);
        gb = *reinterpret_cast<const float4*>(tg + kc + u4 + 2);
      }
      const float pk[4] = {p4.x, p4.y, p4.z, p4.w};
      const float gik[4] = {ga.x, ga.z, gb.x, gb.z};
      const float gek[4] = {ga.y, ga.w, gb.y, gb.w};
      if constexpr (NA >= 4) {
        // one source column at a time, NA independent target columns per stage
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float gi[NA], ge[NA];
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            gi[j] = TPOS ? vminf(gik[u], s.gib[JJ + j]) : gi_c;
            ge[j] = TPOS ? vminf(gek[u], s.geb[JJ + j]) : ge_c;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NA; ++j) ge[j] = ge[j] * s.fd[JJ + j];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NA; ++j) gi[j] = gi[j] + ge[j];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NA; ++j) gi[j] = pk[u] - gi[j];
          if (MASK) gi[0] = (s.fd[JJ] >= 0.f) ? gi[0] : ninf;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NA; ++j) { s.cm[JJ + j] = vmaxf(s.cm[JJ + j], gi[j]); s.fd[JJ + j] -= 1.0f; }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // few target columns: the four source columns of this trip are the independent work
        float gi[4][NA], ge[4][NA], fdu[4][NA];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            gi[u][j] = TPOS ? vminf(gik[u], s.gib[JJ + j]) : gi_c;
            ge[u][j] = TPOS ? vminf(gek[u], s.geb[JJ + j]) : ge_c;
            fdu[u][j] = s.fd[JJ + j] - (float)u;
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < NA; ++j) ge[u][j] = ge[u][j] * fdu[u][j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < NA; ++j) gi[u][j] = gi[u][j] + ge[u][j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            gi[u][j] = pk[u] - gi[u][j];
            if (MASK && !EXM && j == 0) gi[u][j] = (fdu[u][j] >= 0.f) ? gi[u][j] : ninf;
          }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MASK && EXM && NA == 1) {
          const int t0 = kc - k0 + u4 + 1;                  // candidate k0 + j counts for the lanes >= j + 1
          max4_under_masks(s.cm[JJ], gi[0][0], gi[1][0], gi[2][0], gi[3][0], lanes_from(t0), lanes_from(t0 + 1), lanes_from(t0 + 2),
                           lanes_from(t0 + 3));
          s.fd[JJ] -= 4.0f;
        } else {
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            const float a = vmax3f(s.cm[JJ + j], gi[0][j], gi[1][j]);
            s.cm[JJ + j] = vmax3f(a, gi[2][j], gi[3][j]);
            s.fd[JJ + j] -= 4.0f;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = JJ; j < NS; ++j) {
      const bool up = s.cm[j] > s.m[j];
      s.e[j] = up ? s.m[j] : s.e[j];
      s.cidx[j] = up ? kc : s.cidx[j];
      s.m[j] = up ? s.cm[j] : s.m[j];
      s.cm[j] = ninf;
    }
  }
}

// the whole deletion scan of one row for this wave: slot jj's columns are B_jj + lane, B_jj = 1 + 64 w + 256 jj
template <int JJ, int NS, bool TPOS>
__device__ __forceinline__ void scan_all(ScanState<NS>& s, const float* prev, const float2* tg, int wave, int kbeg, int nslots,
                                         float gi_c, float ge_c) {
  if constexpr (JJ < NS) {
    if (JJ < nslots) {
      const int tail = 64 * wave + 256 * JJ;          // = B_JJ - 1: first k that is not a candidate of lane 0
      scan_range<JJ, NS, TPOS, false>(s, prev, tg, kbeg, tail, gi_c, ge_c);
      scan_range<JJ, NS, TPOS, true>(s, prev, tg, tail, tail + 64, gi_c, ge_c);
      scan_all<JJ + 1, NS, TPOS>(s, prev, tg, wave, tail + 64, nslots, gi_c, ge_c);
    }
  }
}

template <int NS, bool TPOS, bool LOCAL>
__global__ __launch_bounds__(kBT, 2) void dp_exact_blocked_kernel(const PairDesc* __restrict__ pairs, EvalDev proto,
                                                                   const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                                   const float* __restrict__ tgi, const float* __restrict__ tge,
                                                                   float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
                                                                   const float* __restrict__ Sbase, PairResult* __restrict__ res, int rev,
                                                                   float* __restrict__ scratch_base) {
  constexpr int PT = NS * 256 + kBPad;                 // LDS row pitch: every k the scans can touch
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rowbuf0 = lds;                                // D[a-1][.] / D[a][.] in frame order, double-buffered
  float* rowbuf1 = lds + PT;
  float2* tg = reinterpret_cast<float2*>(lds + 2 * PT);   // (tgi, tge) in frame order
  float* sres_m = lds + 4 * PT;                        // deletion-scan results of this thread's NS cells (own words only)
  float* sres_e = sres_m + NS * kBT;
  int* sres_c = reinterpret_cast<int*>(sres_e + NS * kBT);
  __shared__ float red_v[kBT / 64];
  __shared__ uint32_t red_p[kBT / 64];
  const float ninf = -__builtin_inff();

  const PairDesc pd = pairs[blockIdx.x];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint32_t* __restrict__ P = Pbase + pd.plane_off;
  const int ld = pd.ld;
  const Frame f = {pd.q0, pd.q1, pd.t0, pd.t1, rev};
  const int nQ = f.nQ(), nT = f.nT();
  const int tid = threadIdx.x, wave = tid >> 6;
  const float gi_c = e.gi, ge_c = e.ge;
  // far-insertion results of the current block: planes m / e / chunk, [kBR][PT] each
  float* scr_m = scratch_base + (size_t)blockIdx.x * 3 * kBR * PT;
  float* scr_e = scr_m + kBR * PT;
  int* scr_c = reinterpret_cast<int*>(scr_e + kBR * PT);

  float lmax = 0.f; uint32_t lpos = 0xFFFFFFFFu;
  const uint32_t origin = pack_ptr(f.rq(0), f.rt(0));

  if (nQ >= 2 && nT >= 2) {
    // ---- stage per-position arrays in frame order; pads read as "no candidate" --------------------------------
    for (int x = tid; x < PT; x += kBT) {
      rowbuf0[x] = ninf; rowbuf1[x] = ninf;
      float2 v = {0.f, 0.f};
      if (TPOS && x <= nT) { const int pos = f.rt(x); v.x = e.tgi[pos]; v.y = e.tge[pos]; }
      tg[x] = v;
    }
    __syncthreads();
    // slots this WAVE has to scan (lane 0's column decides; later lanes of a partly valid slot are masked at the store)
    int nslots = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) if (1 + 64 * wave + 256 * j <= nT - 1) nslots = j + 1;

    for (int a0 = 1; a0 <= nQ - 1; a0 += kBR) {
      // ================= far insertions of rows a0 .. a0+15: candidates k = 1 .. a0-2 ============================
      if (a0 >= 3) {
#pragma unroll 1
        for (int j = 0; j < NS; ++j) {
          if (1 + 64 * wave + 256 * j > nT - 1) break;
          const int bc = 1 + tid + 256 * j;
          const bool bv = bc <= nT - 1;
          const int b = (bc < 2 || !bv) ? 2 : bc;     // b = 1 has no insertions; invalid lanes compute a dummy column
          float gi = gi_c, ge = ge_c;                 // insertion coefficients of column b: min over (b-1, b)  (hmap2_eval.h:69-95)
          if (TPOS) { const float2 t0 = tg[b - 1], t1 = tg[b]; gi = fminr(t0.x, t1.x); ge = fminr(t0.y, t1.y); }
          const size_t colb = (size_t)f.rt(b - 1);
          float W[kBR], cm[kBR], m[kBR], ee[kBR]; int cc[kBR];
          float fn = (float)(a0 - 2);                 // n of (row a0, k = 0); row a0+r adds r
#pragma unroll
          for (int i = 0; i < kBR; ++i) { W[i] = gi + ge * (fn + (float)i); cm[i] = ninf; m[i] = ninf; ee[i] = ninf; cc[i] = 0; }
          // the next chunk's 16 column values are in flight while this chunk is evaluated
          constexpr bool PF = NS > 1;                 // the one-slot kernel has no registers to spare for the prefetch
          float xq[kBR];
          if (PF) {
#pragma unroll
            for (int u = 0; u < kBR; ++u) xq[u] = (u >= 1 && u <= a0 - 2) ? aload(&H[(size_t)f.rq(u) * ld + colb]) : ninf;
          }
          for (int kc = 0; kc <= a0 - 2; kc += kBR) {
            float x[kBR];
            if (PF) {
#pragma unroll
              for (int u = 0; u < kBR; ++u) x[u] = xq[u];
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                const int k = kc + kBR + u;
                xq[u] = (k <= a0 - 2) ? aload(&H[(size_t)f.rq(k) * ld + colb]) : ninf;
              }
            } else {
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                const int k = kc + u;
                x[u] = (k >= 1 && k <= a0 - 2) ? aload(&H[(size_t)f.rq(k) * ld + colb]) : ninf;
              }
            }
#pragma unroll
            for (int u = 0; u < kBR; ++u) {
#pragma unroll
              for (int r = 0; r < kBR; r += 4)
                submax4_vv(cm[r], cm[r + 1], cm[r + 2], cm[r + 3], x[u], W[(r - u) & (kBR - 1)], W[(r + 1 - u) & (kBR - 1)],
                           W[(r + 2 - u) & (kBR - 1)], W[(r + 3 - u) & (kBR - 1)]);
              fn -= 1.0f;
              W[kBR - 1 - u] = gi + ge * fn;          // G(n0 - 1) for the next k
            }
#pragma unroll
            for (int r = 0; r < kBR; ++r) {
              const bool up = cm[r] > m[r];
              ee[r] = up ? m[r] : ee[r];
              cc[r] = up ? kc : cc[r];
              m[r] = up ? cm[r] : m[r];
              cm[r] = ninf;
            }
          }
          if (bv && bc >= 2) {
#pragma unroll
            for (int r = 0; r < kBR; ++r) { scr_m[r * PT + bc] = m[r]; scr_e[r * PT + bc] = ee[r]; scr_c[r * PT + bc] = cc[r]; }
          }
        }
        __builtin_amdgcn_s_waitcnt(0);   // own scratch words are read back by this same thread through L2
      }

      // ================= the rows of the block ===================================================================
      const int a_end = (a0 + kBR - 1 < nQ - 1) ? a0 + kBR - 1 : nQ - 1;
      for (int a = a0; a <= a_end; ++a) {
        const int i = f.rq(a);
        const float* prev = (a & 1) ? rowbuf0 : rowbuf1;       // row a-1
        float* cur = (a & 1) ? rowbuf1 : rowbuf0;
        const int r = a - a0;
        if (a == 1) {
          // ---- first row: dpmatrix.h:409-418 -------------------------------------------------------------------
#pragma unroll 1
          for (int j = 0; j < NS; ++j) {
            const int b = 1 + tid + 256 * j;
            if (b > nT - 1) break;
            const int jj = f.rt(b);
            const float sim = dev_sim(e, i, jj);
            float sv = 0.f;
            if (b > 1) sv -= frame_del(e, f, 0, b);
            sv += sim;
            const float opt = clip0(sv, LOCAL);
            H[(size_t)i * ld + jj] = opt; P[(size_t)i * ld + jj] = origin;
            cur[b] = opt;
            if (opt > lmax) { lmax = opt; lpos = ((uint32_t)a << 16) | (uint32_t)b; }
          }
        } else {
          // ---- deletion scan over row a-1 ----------------------------------------------------------------------
          ScanState<NS> s;
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const int bc = 1 + tid + 256 * j;
            if (TPOS) { const float2 t = tg[bc]; s.gib[j] = t.x; s.geb[j] = t.y; } else { s.gib[j] = 0.f; s.geb[j] = 0.f; }
            s.fd[j] = (float)(bc - 2);
            s.cm[j] = ninf; s.m[j] = ninf; s.e[j] = ninf; s.cidx[j] = 0;
          }
          scan_all<0, NS, TPOS>(s, prev, tg, wave, 0, nslots, gi_c, ge_c);
#pragma unroll
          for (int j = 0; j < NS; ++j) { sres_m[j * kBT + tid] = s.m[j]; sres_e[j * kBT + tid] = s.e[j]; sres_c[j * kBT + tid] = s.cidx[j]; }
          // ---- per cell: match, best deletion, best insertion, pointer ----------------------------------------
#pragma unroll 1
          for (int j = 0; j < NS; ++j) {
            if (1 + 64 * wave + 256 * j > nT - 1) break;
            const int b = 1 + tid + 256 * j;
            const bool valid = b <= nT - 1;
            const float dm = sres_m[j * kBT + tid], de = sres_e[j * kBT + tid];
            const int dc = sres_c[j * kBT + tid];
            const int bb = valid ? b : 1;                     // invalid lanes walk a harmless cell and store nothing
            const int jj = f.rt(bb);
            const float sim = dev_sim(e, i, jj);
            float opt; uint32_t optp;
            if (bb == 1) {                                     // dpmatrix.h:421-426
              float sv = 0.f;
              sv -= frame_ins(e, f, 0, a, 0, 1);
              sv += sim;
              opt = clip0(sv, LOCAL); optp = origin;
            } else {
              const size_t colb = (size_t)f.rt(bb - 1);
              // near insertion candidates k = kn0 .. a-2 (rows of this block, and the row just before it)
              const int kn0 = (a0 - 1 > 1) ? a0 - 1 : 1;
              float xn[kBR];
#pragma unroll
              for (int u = 0; u < kBR; ++u) xn[u] = (kn0 + u <= a - 2) ? aload(&H[(size_t)f.rq(kn0 + u) * ld + colb]) : ninf;
              float mf = ninf, ef = ninf; int cf = 0;
              if (a0 >= 3) { mf = aload(&scr_m[r * PT + bb]); ef = aload(&scr_e[r * PT + bb]); cf = aloadi(&scr_c[r * PT + bb]); }
              opt = clip0(prev[bb - 1] + sim, LOCAL);          // match, :447-451
              int cat = 0;
              const float sd = clip0(dm + sim, LOCAL);
              if (sd > opt) { opt = sd; cat = 1; }
              const float sfar = clip0(mf + sim, LOCAL);
              float snear = ninf; int knear = 0;
              float gi = gi_c, ge = ge_c;
              if (TPOS) { const float2 t0 = tg[bb - 1], t1 = tg[bb]; gi = fminr(t0.x, t1.x); ge = fminr(t0.y, t1.y); }
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                if (kn0 + u <= a - 2) {
                  float sv = xn[u];
                  sv -= gi + ge * (float)(a - (kn0 + u) - 2);
                  sv += sim;
                  sv = clip0(sv, LOCAL);
                  if (sv > snear) { snear = sv; knear = kn0 + u; }
                }
              }
              const float si = (snear > sfar) ? snear : sfar;
              if (si > opt) { opt = si; cat = (snear > sfar) ? 3 : 2; }
              int oa = a - 1, ob = bb - 1;
              if (cat == 1) {
                // first deletion k whose literal score is the maximum: inside chunk cidx unless an earlier chunk ties after rounding
                const bool amb = clip0(de + sim, LOCAL) == opt;
                int k = amb ? 1 : (dc > 1 ? dc : 1);
                const float gbi = TPOS ? tg[bb].x : 0.f, gbe = TPOS ? tg[bb].y : 0.f;
                for (; k < bb - 2; ++k) {
                  const float2 tk = tg[k];
                  const float g = (TPOS ? fminr(tk.x, gbi) : gi_c) + (TPOS ? fminr(tk.y, gbe) : ge_c) * (float)(bb - k - 2);
                  float sv = prev[k];
                  sv -= g;
                  sv += sim;
                  sv = clip0(sv, LOCAL);
                  if (sv == opt) break;
                }
                oa = a - 1; ob = k;                           // k == bb-2 is the last candidate: taken if nothing earlier matched
              } else if (cat == 2) {
                const bool amb = clip0(ef + sim, LOCAL) == opt;
                int k = amb ? 1 : (cf > 1 ? cf : 1);
                for (; k < a0 - 2; ++k) {
                  float sv = aload(&H[(size_t)f.rq(k) * ld + colb]);
                  sv -= gi + ge * (float)(a - k - 2);
                  sv += sim;
                  sv = clip0(sv, LOCAL);
                  if (sv == opt) break;
                }
                oa = k; ob = bb - 1;
              } else if (cat == 3) {
                oa = knear; ob = bb - 1;
              }
              optp = pack_ptr(f.rq(oa), f.rt(ob));
            }
            if (valid) {
              H[(size_t)i * ld + jj] = opt; P[(size_t)i * ld + jj] = optp;
              cur[b] = opt;
              if (opt > lmax) { lmax = opt; lpos = ((uint32_t)a << 16) | (uint32_t)b; }
            }
          }
        }
        __threadfence_block();
        __syncthreads();        // row a complete: in LDS for the next row's scans, in L2 for later column walks
      }
    }
  }
  // find_max partial in frame coordinates (first in frame row-major order among the maxima)
  float m = lmax; uint32_t p = lpos;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    float om = __shfl_xor(m, o); uint32_t op = (uint32_t)__shfl_xor((int)p, o);
    bool take = om > m || (om == m && op < p);
    m = take ? om : m; p = take ? op : p;
  }
  if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = m; red_p[threadIdx.x >> 6] = p; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBT / 64; ++w) {
      bool take = red_v[w] > m || (red_v[w] == m && red_p[w] < p);
      if (take) { m = red_v[w]; p = red_p[w]; }
    }
    uint32_t rp = 0xFFFFFFFFu;
    if (p != 0xFFFFFFFFu && m > 0.f) rp = ((uint32_t)f.rq((int)(p >> 16)) << 16) | (uint32_t)f.rt((int)(p & 0xFFFFu));
    res[blockIdx.x].part_max = m;
    res[blockIdx.x].part_pos = rp;
  }
}


// =====================================================================================================================
// dp_exact_tiled_kernel — same results again, with the deletion scan shared between rows.
//
// The gap of a deletion (k -> b) does not depend on the row, only the source value D[a-1][k] does.  Rows go in blocks
// of 16 and columns in tiles of 256 (one column per thread), tiles left to right inside a row block.  For a tile
// starting at column b0 every source column k <= b0-2 of the 16 source rows a0-1 .. a0+14 is final (earlier tiles of
// this block, or the previous block), so the "far-left" part of all 16 deletion scans is one loop: per (k, b) the gap
// g(k,b) is formed ONCE (min, min, mul, add) and applied to 16 rows (sub, max each) — 2.3 VALU ops per candidate instead
// of 7.  The 16 source values per k are wave-uniform: they come through the scalar cache (s_load_dwordx2 from a
// frame-ordered copy of the finished rows) and enter the VALU as SGPR operands, so neither LDS nor VGPRs carry them.
// The remaining candidates k in [b0-1, b-2] (inside the tile, <= 255) are scanned per row from LDS exactly like
// dp_exact_blocked does, starting from the far-left (max, e, chunk) state, so the "first k" bookkeeping is unchanged.
// Far insertions: the 16-row sliding window of dp_exact_blocked, per tile.
//
// Two forms of the schedule (template parameter WF; hint exact_wavefront, default on):
//   256-column tiles: the four waves share a tile, a row of it needs the previous row of ALL of it -> one workgroup barrier per
//     row (18 k per 2000 x 2000 pair), the wave that owns the tile's last 64 columns scans 256 in-tile candidates per cell and the
//     first 64, and the others wait for it;
//   wavefront (round 3): 64-column tiles, wave w owns the row blocks w, w+4, ... and sweeps each left to right one tile behind
//     the wave that owns the block above.  A tile's own candidates are the wave's own previous rows (kept in LDS, which also
//     serves the near insertions), so nothing is shared inside a tile: no workgroup barrier at all — a wave publishes its tile
//     count in LDS once its stores have reached L2, its follower polls that word — every wave does the same work, and the
//     in-tile scan is 64 candidates per cell for all of them (18 % fewer VALU instructions on config 3).  A lone 700 x 700 batch
//     of 8 pairs: 36.8 -> 23.3 ms; 1024 pairs of 2000 x 2000: 408 -> 397 ms (the far scans' memory round trips bound that one).
// Pointers: the scans leave (chunk, tie flag) per cell; which k of the chunk it was is found by the WAVE — the 32 (16)
// candidates of a cell are evaluated by as many lanes at once, two (four) cells per round — not by each lane walking alone.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v sload2(const float* p) {
  f2v v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}
typedef float f4v __attribute__((ext_vector_type(4)));
// s_load_dwordx4 with an immediate byte offset (the 16 source rows of a row block sit PT floats apart in the scratch ring)
template <int OFF>
__device__ __forceinline__ f4v sload4_imm(const float* p) {
  f4v v;
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(v) : "s"(p), "n"(OFF) : "memory");
  return v;
}
template <int R, int NR, int PTC>
__device__ __forceinline__ void sload_rows(f4v (&dst)[NR], const float* p) {
  if constexpr (R < NR) {
    dst[R] = sload4_imm<R * PTC * 4>(p);
    sload_rows<R + 1, NR, PTC>(dst, p);
  }
}
template <int OFF>
__device__ __forceinline__ f2v sload2_imm(const float* p) {
  f2v v;
  asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v) : "s"(p), "n"(OFF) : "memory");
  return v;
}
template <int R, int NR, int PTC>
__device__ __forceinline__ void sload_rows2(f2v (&dst)[NR], const float* p) {
  if constexpr (R < NR) {
    dst[R] = sload2_imm<R * PTC * 4>(p);
    sload_rows2<R + 1, NR, PTC>(dst, p);
  }
}
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f16v sload16(const float* p) {
  f16v v;
  // load and wait in ONE statement: between a lone s_load and a later s_waitcnt the compiler may move or spill the destination
  // registers (this kernel spills SGPRs), which would read them before the data arrives
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ void swait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- skipping whole chunks of far candidates (bit-exact) ------------------------------------------------------------------------
// A far scan keeps, per cell, (m, e, c) = the best d = D_k - g_k, the chunk c that attains it first and the best d of the chunks
// before c.  Chunks are visited NEAREST FIRST.  A chunk K may be left out when, for every cell the wave serves,
//        ub(K) + margin < m            ub(K) = fl(max_{k in K} D_k - g_lb(K)),  g_lb = the smallest gap any k in K can have
// (gaps grow with the distance as long as the extension coefficients are >= 0; rounding is monotone, so ub >= every d of K):
//  * no d of K reaches m, so neither m nor "first chunk attaining m" changes;
//  * `e` would change only if K lies before c — and e is used for one thing, the test fl(e + S) == fl(m + S) that sends a
//    cell won by a gap to the literal re-walk from the start.  fl(x + S) == fl(y + S) with x < y needs y - x <= ulp(fl(y + S))
//    <= 2^-23 (|y| + |S|); the margin 2^-21 (|m| + max|S| + 1) (minus what the two roundings of its own computation can lose)
//    excludes it.  So every cell's score and pointer come out exactly as if K had been scanned.
// The maxima come from what the kernel already produced: per finished row the maximum of every 32-column chunk (deletion
// scans; 16 rows of a chunk = one s_load_dwordx16), per column the maximum of every 16-row block (insertion scans).
// Measured on config 3 (2000 x 2000 profile pairs, global): 88 % of the far-left deletion chunks and 86 % of the far insertion
// chunks are skipped (tools/c3_prune_estimate.py replays the rule on the host).
constexpr float kPruneEps = 4.76837158203125e-07f;     // 2^-21
__device__ __forceinline__ float prune_thr(float m, float ceps) { return __builtin_fmaf(__builtin_fabsf(m), -kPruneEps, m) - ceps; }
__device__ __forceinline__ float vmin_sv(float s, float v) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(v)); return r; }

constexpr int kTW = 256;       // tile width = threads
constexpr int kTRing = 32;     // frame-ordered copies of finished rows kept per workgroup (>= kBR + 1)
constexpr int kWRing = 128;    // wavefront form: four row blocks are in flight (+ the row above the oldest)
constexpr int kWT = 64;        // wavefront form: tile width = one wave
constexpr int kWRow = 68;       // wavefront form: a wave's LDS copy of one row of its tile (column kbase + its 64 columns, padded to 16 bytes)
constexpr int kWHist = (kBR + 1) * kWRow;   // ... rows a0-1 .. a0+15: the sources of the tile's own deletion scans AND of its near insertions
constexpr int kTLoc = 320;     // tile-local row buffer: 257 live entries + pads the masked tail may read

// GM = gap model: 0 constant affine (aasubalib.h:27-77), 1 min(t1,t2) position coefficients (hmap2_eval.h:41-95), 2 Gn2Eval's
// model (gn2_eval.h:100-165): deletion(t1,t2) read from a per-template T x T table — it depends on the two template positions
// only, so one loaded value serves the 16 rows of a block exactly like a computed one — and insertion(dist) =
// (gi[t1] + ge[t1] * (dist - 2)) + cn[t1] with the coefficients of the SMALLER template position alone.
// `delF`: GM 2 only; entry [k * pitch + b] of pair p's table at delF_off[p] is the deletion between FRAME columns k < b
// (forward builds: the caller's table; reverse builds: its flipped transpose, see launch_dp_exact_blocked).
template <int PT, int GM, bool LOCAL, bool WF>
__global__ __launch_bounds__(kTW, 4) void dp_exact_tiled_kernel(const PairDesc* __restrict__ pairs, EvalDev proto,
                                                                 const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                                 const float* __restrict__ tgi, const float* __restrict__ tge,
                                                                 float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
                                                                 const float* __restrict__ Sbase, PairResult* __restrict__ res, int rev,
                                                                 float* __restrict__ scratch_base, int alt_prio,
                                                                 const float* __restrict__ delF_base, const int64_t* __restrict__ delF_off,
                                                                 int prune, int q_blocks, const float* __restrict__ smax_arr, float smax_const,
                                                                 unsigned long long* __restrict__ dbg, unsigned int* __restrict__ gprog) {
  constexpr bool TPOS = GM == 1;
  constexpr bool TAB = GM == 2;
  constexpr int RING = WF ? kWRing : kTRing;            // finished rows kept per workgroup
  constexpr int TWv = WF ? kWT : kTW;                   // tile width
  __shared__ float2 cminl[PT / 32 + 1];                 // per 32-column chunk: the smallest (tgi, tge) in it (frame order)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tg = reinterpret_cast<float2*>(lds);          // (tgi, tge) in frame order, PT entries
  float* rowloc0 = lds + 2 * PT;                         // rows a-1 / a of the current tile, index k - kbase
  float* rowloc1 = rowloc0 + kTLoc;
  float* tcnl = rowloc0 + (WF ? 4 * kWHist : 2 * kTLoc);                         // GM 2: Gn2Eval's v_cn in frame order, PT entries
  __shared__ float red_v[kTW / 64];
  __shared__ int wdone[kTW / 64];                       // WF: tiles finished by each wave
  __shared__ float edgeL[kTW / 64][32];                 // WF: column kbase (the previous tile's last) of rows a0-1 .. a0+15, per wave
  __shared__ uint32_t red_p[kTW / 64];
  const float ninf = -__builtin_inff();

  const PairDesc pd = pairs[blockIdx.x];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  bind_table_model(e, proto, pd);
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint32_t* __restrict__ P = Pbase + pd.plane_off;
  const int ld = pd.ld;
  const Frame f = {pd.q0, pd.q1, pd.t0, pd.t1, rev};
  const int nQ = f.nQ(), nT = f.nT();
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: what the wavefront form derives from it (row block, tile) stays in SGPRs
  const int lt = WF ? (tid & 63) : tid;                  // thread's place in its tile
  const float gi_c = e.gi, ge_c = e.ge;
  // GM 2: the frame-ordered deletion table of this pair, pitch = the template's real length; frame column 0 = real t0 (forward)
  // or t1 (reverse, in the flipped table: index T-1-t1)
  const int TT = pd.T;
  const float* __restrict__ delF = nullptr;
  if (TAB) delF = delF_base + delF_off[blockIdx.x] + (size_t)(rev ? (pd.T - 1 - pd.t1) : pd.t0) * (size_t)(TT + 1);
  auto del_at = [&](int k, int b) -> float { return delF[(size_t)(k < TT ? k : TT - 1) * TT + b]; };   // k clamped: masked candidates only
  // per-workgroup scratch: far-insertion and far-left-deletion results [2][3][kBR][PT] (own words only, read back through L2),
  // finished rows in frame order [kTRing][PT], (tgi, tge) [2][PT]
  float* const scr0 = scratch_base + (size_t)blockIdx.x * (size_t)(6 * kBR + RING + 2 + RING / 32 + q_blocks) * PT;
  // far-scan results (m, e, chunk) x (insertions, deletions), [kBR][SP] each.  WF: a wave keeps its own six planes for the tile it
  // is working on (64 columns; the waves are not in lockstep, so two of them may be at the same columns of different row blocks)
  constexpr int SP = WF ? kWT : PT;
  float* scr_m = WF ? scr0 + (size_t)wave * (6 * kBR * kWT) : scr0;
  float* scr_e = scr_m + kBR * SP;
  int* scr_c = reinterpret_cast<int*>(scr_e + kBR * SP);
  float* fdm = scr_m + 3 * kBR * SP;
  float* fde = fdm + kBR * SP;
  int* fdc = reinterpret_cast<int*>(fde + kBR * SP);
  float* rowsF = scr0 + 6 * kBR * PT;
  float* tgiF = rowsF + RING * PT;
  float* tgeF = tgiF + PT;
  float* delmaxF = tgeF + PT;                            // [PT/32 chunks][kTRing row slots]: chunk maxima of finished rows
  float* insmaxF = delmaxF + (RING / 32) * PT;                         // [q_blocks][PT]: column maxima of finished 16-row blocks
  const float ceps = ((smax_arr ? smax_arr[blockIdx.x] : smax_const) + 1.0f) * kPruneEps;
  const bool prune_del = (prune & 1) && !TAB;            // a tabulated deletion has no monotone lower bound
  const bool prune_ins = (prune & 2) != 0;
  unsigned n_tested_d = 0, n_skip_d = 0, n_tested_i = 0, n_skip_i = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();

  float lmax = 0.f; uint32_t lpos = 0xFFFFFFFFu;
  const uint32_t origin = pack_ptr(f.rq(0), f.rt(0));

  if (nQ >= 2 && nT >= 2) {
    for (int x = tid; x < PT; x += kTW) {
      float2 v = {0.f, 0.f};
      if ((TPOS || TAB) && x <= nT) { const int pos = f.rt(x); v.x = e.tgi[pos]; v.y = e.tge[pos]; }
      tg[x] = v;
      tgiF[x] = v.x; tgeF[x] = v.y;
      if (TAB) tcnl[x] = (x <= nT) ? e.tcn[f.rt(x)] : 0.f;
    }
    for (int x = tid; x < (WF ? 4 * kWHist : 2 * kTLoc); x += kTW) rowloc0[x] = ninf;
    if (tid < kTW / 64) wdone[tid] = 0;
    __threadfence_block();
    __syncthreads();
    for (int c = tid; c < PT / 32; c += kTW) {
      float2 mn = tg[32 * c];
#pragma unroll 8
      for (int u = 1; u < 32; ++u) { const float2 v = tg[32 * c + u]; mn.x = fminr(mn.x, v.x); mn.y = fminr(mn.y, v.y); }
      cminl[c] = mn;
    }
    __syncthreads();
    const int ntiles = (nT - 1 + TWv - 1) / TWv;
    float* const rlw = rowloc0 + wave * kWHist;           // WF: this wave's rows of the current tile

    const int hwslot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_ID.WAVE_ID
    // The SIMD arbiter favours its oldest wave, so of the 4 pairs resident on a CU the first would run ahead and the last
    // finish alone on a half-empty SIMD (measured on the tagged kernel).  Rotating the user priority over
    // the 4 wave slots of a SIMD, one step per row block, keeps the pairs abreast: 854 -> 771 ms on config 3.
    // alt_prio 2: by PROGRESS.  Measured (exact_debug 2): with 1024 pairs of one size the average wave is done after 0.6 of the
    // launch — the arbiter's favourites run ahead, finish, and leave the rest to run with fewer waves per SIMD, which is slower
    // for everyone left.  Every wave counts its row blocks in one device word as it starts them; what it gets back is how far
    // the launch is on average, and a wave ahead of that lowers its priority, a wave behind raises it.
    auto set_prio = [&](int a0) {
      if (alt_prio == 2) {
        unsigned tot = 0;
        if ((tid & 63) == 0) tot = __hip_atomic_fetch_add(gprog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot = __builtin_amdgcn_readfirstlane(tot);
        const int mine = WF ? (a0 - 1) / (4 * kBR) : (a0 - 1) / kBR;       // blocks this wave has started before this one
        const int lead = mine - (int)(tot / (4u * gridDim.x));
        if (lead >= 2) __builtin_amdgcn_s_setprio(0);
        else if (lead == 1) __builtin_amdgcn_s_setprio(1);
        else if (lead == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      } else if (alt_prio) {
        switch (((a0 / kBR) + hwslot) & 3) {
          case 0: __builtin_amdgcn_s_setprio(0); break;
          case 1: __builtin_amdgcn_s_setprio(1); break;
          case 2: __builtin_amdgcn_s_setprio(2); break;
          default: __builtin_amdgcn_s_setprio(3); break;
        }
      }
    };
    // one tile: rows a0 .. a0+15 x columns TWv * cb + 1 .. TWv * cb + TWv
    auto do_tile = [&](const int a0, const int cb) __attribute__((always_inline)) {
      const int a_end = (a0 + kBR - 1 < nQ - 1) ? a0 + kBR - 1 : nQ - 1;
      {
        const int kbase = TWv * cb;                    // = b0 - 1: first near source column of the tile
        const int bc = kbase + 1 + lt;                // this thread's column
        const int sc = WF ? lt : bc;                   // ... and its place in the far-scan result planes
        const bool bv = bc <= nT - 1;
        const bool wave_on = WF ? true : kbase + 1 + 64 * wave <= nT - 1;
        float colmax = ninf;                           // this column's maximum over the rows of the block (tile epilogue)
        // ============ far insertions of rows a0 .. a0+15 for column bc: candidates k = 1 .. a0-2 =====================
        // Source rows in chunks of 16 that coincide with the row blocks (rows 16j+1 .. 16j+16), nearest chunk first; (m, e, c)
        // as a left-to-right walk would leave them: c = the LOWEST chunk that attains m (a tie moves it down: >=), e = the best
        // of the chunks below c (reset whenever c moves).  Chunks other than the nearest are tested against the column maximum
        // of their row block and skipped when they cannot matter ("skipping whole chunks" above).
        if (a0 >= 3 && wave_on) {
          const int b = (bc < 2 || !bv) ? 2 : bc;
          float gi = gi_c, ge = ge_c, cn = 0.f;
          if (TPOS) { const float2 t0 = tg[b - 1], t1 = tg[b]; gi = fminr(t0.x, t1.x); ge = fminr(t0.y, t1.y); }
          if (TAB) { const int lo = rev ? b : b - 1; const float2 t = tg[lo]; gi = t.x; ge = t.y; cn = tcnl[lo]; }   // the smaller real position
          const size_t colb = (size_t)f.rt(b - 1);
          const float* cmaxcol = insmaxF + (b - 1);
          const bool lane_live = bv && bc >= 2;
          const int jtop = (a0 - 3) >> 4;              // the chunk that holds row a0-2
          constexpr int HW = kBR / 2;                  // two 8-row windows: half the registers of one 16-row window
#pragma unroll 1
          for (int h = 0; h < 2; ++h) {
            float W[HW], cm[HW], m[HW], ee[HW]; int cc[HW];
#pragma unroll
            for (int i = 0; i < HW; ++i) { m[i] = ninf; ee[i] = ninf; cc[i] = 0; }
            auto scan_chunk = [&](int j) {                 // rows 16j+1 .. 16j+16 (those <= a0-2) against the window's 8 target rows
              const int kc = 1 + 16 * j;
              float fn = (float)(a0 + HW * h - 2 - kc);   // n of (row a0 + 8h, k = kc)
#pragma unroll
              for (int i = 0; i < HW; ++i) {
                W[i] = gi + ge * (fn + (float)i);
                if (TAB) W[i] = W[i] + cn;                    // gn2_eval.h: gp = gi + ge * (di - 2); gp = gp + cn
                cm[i] = ninf;
              }
              float x[kBR];
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                const int k = kc + u;
                x[u] = (k <= a0 - 2) ? aload(&H[(size_t)f.rq(k) * ld + colb]) : ninf;
              }
#pragma unroll
              for (int u = 0; u < kBR; u += 2) {               // two source rows per step: the window as row u sees it, then slid by one
                float wa[HW], wb[HW];
#pragma unroll
                for (int r = 0; r < HW; ++r) wa[r] = W[(r - u) & (HW - 1)];
                fn -= 1.0f;
                W[(HW - 1 - u) & (HW - 1)] = TAB ? (gi + ge * fn) + cn : gi + ge * fn;
#pragma unroll
                for (int r = 0; r < HW; ++r) wb[r] = W[(r - u - 1) & (HW - 1)];
                fn -= 1.0f;
                W[(HW - 2 - u) & (HW - 1)] = TAB ? (gi + ge * fn) + cn : gi + ge * fn;
#pragma unroll
                for (int r = 0; r < HW; r += 4)
                  submax3x4_vv(cm[r], cm[r + 1], cm[r + 2], cm[r + 3], x[u], wa[r], wa[r + 1], wa[r + 2], wa[r + 3],
                               x[u + 1], wb[r], wb[r + 1], wb[r + 2], wb[r + 3]);
              }
#pragma unroll
              for (int r = 0; r < HW; ++r) {
                const bool up = cm[r] >= m[r];
                ee[r] = up ? ninf : vmaxf(ee[r], cm[r]);
                cc[r] = up ? kc : cc[r];
                m[r] = up ? cm[r] : m[r];
              }
            };
            scan_chunk(jtop);                              // the nearest chunk: always
            // the others in groups of 8, nearest group first: the group's 8 column maxima are loaded together (one memory
            // latency per group instead of one per chunk), every chunk is tested against the state BEFORE the group (m only
            // grows, so a chunk that may be skipped now may be skipped later), the chunks that fail are scanned nearest first.
            // (Measured on one box, 1024 pairs: groups of 16: 334 vs 330 ms; the next group's maxima requested before this
            // group's chunks are scanned: 332 with groups of 8, 380 with 16 — neither is the bound.)
            constexpr int IG = 8;
#pragma unroll 1
            for (int jg = jtop - 1; jg >= 0; jg -= IG) {
              const int ng = jg + 1 < IG ? jg + 1 : IG;
              unsigned todo = (1u << ng) - 1u;             // bit t = chunk jg - t
              if (prune_ins) {
                float cmxv[IG], thr[HW];
#pragma unroll
                for (int t = 0; t < IG; ++t) cmxv[t] = (t < ng) ? aload(cmaxcol + (size_t)(jg - t) * PT) : ninf;
#pragma unroll
                for (int i = 0; i < HW; ++i) thr[i] = (a0 + HW * h + i > a_end) ? __builtin_inff() : prune_thr(m[i], ceps);
                // one comparison first: the window's nearest row has the smallest gap, so its bound against the LOWEST of the eight
                // thresholds dominates all eight row tests; only a chunk that fails it gets them (a tenth does)
                float thr_min = thr[0];
#pragma unroll
                for (int i = 1; i < HW; ++i) thr_min = vminf(thr_min, thr[i]);
                todo = 0u;
#pragma unroll
                for (int t = 0; t < IG; ++t) {
                  if (t < ng) {
                    const float fn0 = (float)(a0 + HW * h - 2 - (1 + 16 * (jg - t) + 15));   // n of (first window row, nearest row of the chunk)
                    float gl0 = gi + ge * fn0;
                    if (TAB) gl0 = gl0 + cn;
                    if (__ballot((cmxv[t] - gl0 < thr_min) || !lane_live) != ~0ull) {
                      bool ok = true;
#pragma unroll
                      for (int i = 0; i < HW; ++i) {
                        float gl = gi + ge * (fn0 + (float)i);
                        if (TAB) gl = gl + cn;
                        ok = ok && (cmxv[t] - gl < thr[i]);
                      }
                      if (__ballot(ok || !lane_live) != ~0ull) todo |= 1u << t;
                    }
                  }
                }
                n_tested_i += (unsigned)ng;
                n_skip_i += (unsigned)(ng - __builtin_popcount(todo));
              }
              while (todo) {
                const int t = __builtin_ctz(todo);
                todo &= todo - 1u;
                scan_chunk(jg - t);
              }
            }
            if (bv && bc >= 2) {
#pragma unroll
              for (int r = 0; r < HW; ++r) {
                scr_m[(HW * h + r) * SP + sc] = m[r]; scr_e[(HW * h + r) * SP + sc] = ee[r]; scr_c[(HW * h + r) * SP + sc] = cc[r];
              }
            }
          }
        }
        // ============ far-left deletions: sources k = 1 .. kbase-1 of rows a0-1 .. a0+14, shared gap values ===========
        // 32-column chunks, nearest first, same (m, e, c) convention and skip rule as the far insertions.
        if (cb >= 1 && a_end >= 2 && wave_on) {
          __builtin_amdgcn_s_dcache_inv();             // the source rows were written through the vector path
          const int b = bv ? bc : kbase + 1;
          float gib = 0.f, geb = 0.f;
          if (TPOS) { const float2 t = tg[b]; gib = t.x; geb = t.y; }
          float cm[kBR], m[kBR], ee[kBR]; int cc[kBR];
#pragma unroll
          for (int r = 0; r < kBR; ++r) { m[r] = ninf; ee[r] = ninf; cc[r] = 0; }
          // source row of target row a0+r is a0+r-1: ring slots (a0-1) & 31 + r — consecutive, because a0-1 is a multiple of 16
          const int slot0 = (a0 - 1) & (RING - 1);
          const float* sbase = rowsF + (size_t)slot0 * PT;
          const float* mbase = delmaxF + slot0;
          const int r_lo = (a0 == 1) ? 1 : 0, r_hi = a_end - a0;      // target rows whose far-left state is used
          const int ctop = kbase / kBC - 1;
          // Skip tests, four chunks at a time: lane l holds the maximum of row (l & 15) of chunk gbase - (l >> 4) — one vector load
          // per group, the NEXT group's already in flight while this one is tested and scanned (a scalar load per chunk was one
          // dependent round trip per test) — and a row's bound reaches the lanes as a v_readlane operand.  A group is tested against
          // the state before it (m only grows: what may be skipped now may be skipped later).
          const int ln = tid & 63;
          const float* mlane = mbase + (ln & 15);
          const int lq = ln >> 4;
          auto gload = [&](int gb) { int ch = gb - lq; ch = ch < 0 ? 0 : ch; return aload(mlane + (size_t)ch * RING); };
          int gbase = ctop;
          unsigned todo = 1u;                            // bit t = chunk gbase - t; the nearest chunk: always
          float nxt = (prune_del && ctop >= 1) ? gload(ctop - 1) : 0.f;
#pragma unroll 1
          for (;;) {
            // 4 source columns x 16 rows per trip: 18 scalar loads, one wait, 148 VALU instructions.  (Double-buffering the
            // SGPRs was tried: under the kernel's SGPR pressure the compiler copies the in-flight registers and waits early.)
#pragma unroll 1
            while (todo) {
              const int c = gbase - __builtin_ctz(todo);
              todo &= todo - 1u;
              const int kc = c * kBC;
              float fd = (float)(b - 2 - kc);
#pragma unroll
              for (int r = 0; r < kBR; ++r) cm[r] = ninf;
#pragma unroll 1
              for (int k = kc; k < kc + kBC; k += 4) {
                f4v src[kBR];
                sload_rows<0, kBR, PT>(src, sbase + k);
                f4v gk = {0.f, 0.f, 0.f, 0.f}, ek = {0.f, 0.f, 0.f, 0.f};
                if (TPOS) { gk = sload4_imm<0>(tgiF + k); ek = sload4_imm<PT * 4>(tgiF + k); }
                float gt[4] = {0.f, 0.f, 0.f, 0.f};
                if (TAB) {
#pragma unroll
                  for (int u = 0; u < 4; ++u) gt[u] = delF[(size_t)(k + u) * TT + b];      // k + u < kbase <= b - 1: inside the table
                }
                swait_lgkm0();
                float g4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const float gi = TPOS ? vmin_sv(gk[u], gib) : gi_c;
                  const float ge = TPOS ? vmin_sv(ek[u], geb) : ge_c;
                  g4[u] = TAB ? gt[u] : gi + ge * fd;
                  fd -= 1.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                  if (u == 0 && k == 0) {                  // column 0 is never a source (dpmatrix.h:459 starts at t0+1): column 1 alone
#pragma unroll
                    for (int r = 0; r < kBR; r += 4)
                      submax4_sv(cm[r], cm[r + 1], cm[r + 2], cm[r + 3], src[r][1], src[r + 1][1], src[r + 2][1], src[r + 3][1], g4[1]);
                  } else {
#pragma unroll
                    for (int r = 0; r < kBR; r += 4)
                      submax3x4_sv(cm[r], cm[r + 1], cm[r + 2], cm[r + 3], src[r][u], src[r + 1][u], src[r + 2][u], src[r + 3][u], g4[u],
                                   src[r][u + 1], src[r + 1][u + 1], src[r + 2][u + 1], src[r + 3][u + 1], g4[u + 1]);
                  }
                }
              }
#pragma unroll
              for (int r = 0; r < kBR; ++r) {
                const bool up = cm[r] >= m[r];
                ee[r] = up ? ninf : vmaxf(ee[r], cm[r]);
                cc[r] = up ? kc : cc[r];
                m[r] = up ? cm[r] : m[r];
              }
            }
            gbase -= (gbase == ctop) ? 1 : 4;
            if (gbase < 0) break;
            const int ng = gbase + 1 < 4 ? gbase + 1 : 4;
            todo = (1u << ng) - 1u;
            if (prune_del) {
              const float curv = nxt;
              if (gbase >= 4) nxt = gload(gbase - 4);
              float thr[kBR];
#pragma unroll
              for (int r = 0; r < kBR; ++r) thr[r] = (r < r_lo || r > r_hi) ? __builtin_inff() : prune_thr(m[r], ceps);
              // one comparison first: the chunk's maximum over the 16 rows (a DPP maximum down the 16 lanes that hold them) against
              // the LOWEST of the sixteen thresholds dominates the sixteen row tests; only a chunk that fails it gets them
              float thr_min = thr[0];
#pragma unroll
              for (int r = 1; r < kBR; ++r) thr_min = vminf(thr_min, thr[r]);
              float mxall = ((ln & 15) < r_lo || (ln & 15) > r_hi) ? ninf : curv;      // rows that are no targets do not count
              mxall = vmaxf(mxall, row_shr_f<1>(mxall));
              mxall = vmaxf(mxall, row_shr_f<2>(mxall));
              mxall = vmaxf(mxall, row_shr_f<4>(mxall));
              mxall = vmaxf(mxall, row_shr_f<8>(mxall));                              // lane 16 t + 15: chunk gbase - t
              todo = 0u;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                if (t < ng) {
                  const int c = gbase - t;
                  const float2 cmn = cminl[c];
                  const float dist = (float)(b - (c * kBC + kBC - 1) - 2);
                  const float g_lb = (TPOS ? fminr(cmn.x, gib) : gi_c) + (TPOS ? fminr(cmn.y, geb) : ge_c) * dist;
                  const float mxc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mxall), 16 * t + 15));
                  if (__ballot((mxc - g_lb < thr_min) || !bv) != ~0ull) {
                    bool ok = true;
#pragma unroll
                    for (int r = 0; r < kBR; ++r) {
                      const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, curv), 16 * t + r));
                      ok = ok && (mx - g_lb < thr[r]);
                    }
                    if (__ballot(ok || !bv) != ~0ull) todo |= 1u << t;
                  }
                }
              }
              n_tested_d += (unsigned)ng;
              n_skip_d += (unsigned)(ng - __builtin_popcount(todo));
            }
          }
#pragma unroll
          for (int r = 0; r < kBR; ++r) { fdm[r * SP + sc] = m[r]; fde[r * SP + sc] = ee[r]; fdc[r * SP + sc] = cc[r]; }
        }
        // ============ tile prologue: row a0-1 of the tile's columns (and column b0-1) into the local buffer ==========
        {
          float* pv = WF ? rlw : ((a0 - 1) & 1) ? rowloc1 : rowloc0;
          const float* src = rowsF + (size_t)((a0 - 1) & (RING - 1)) * PT;
          float v = ninf;
          if (a0 >= 2 && bv) v = aload(&src[bc]);
          pv[1 + lt] = v;
          if constexpr (WF) {
            // column kbase of rows a0-1 .. a0+15 (rows of this block: written by this wave one step ago; row a0-1: by the wave
            // that owns the block above, at least one step ago)
            if (lt <= kBR) {
              const int row = a0 - 1 + lt;
              const float ev = (cb >= 1 && row >= 1 && row <= nQ - 1) ? aload(&rowsF[(size_t)(row & (RING - 1)) * PT + kbase]) : ninf;
              edgeL[wave][lt] = ev;
              if (lt == 0) pv[0] = ev;
            }
          } else {
            if (tid == 0) pv[0] = (a0 >= 2 && cb >= 1) ? aload(&src[kbase]) : ninf;
          }
        }
        __builtin_amdgcn_s_waitcnt(0);
        if constexpr (WF) { __threadfence_block(); __builtin_amdgcn_wave_barrier(); } else { __syncthreads(); }

        // ============ the rows of the block inside this tile ==========================================================
        for (int a = a0; a <= a_end; ++a) {
          const int i = f.rq(a);
          const int r = a - a0;
          float* prevl = WF ? rlw + r * kWRow : ((a - 1) & 1) ? rowloc1 : rowloc0;
          float* curl = WF ? rlw + (r + 1) * kWRow : (a & 1) ? rowloc1 : rowloc0;
          const float* prev = prevl - kbase;            // prev[k] for absolute k in [kbase, kbase + kTLoc)
          float* rowF = rowsF + (size_t)(a & (RING - 1)) * PT;
          if (wave_on) {
            const bool valid = bv;
            const int bb = valid ? bc : 1;
            const int sb = WF ? lt : bb;
            const int jj = f.rt(bb);
            const float sim = dev_sim(e, i, jj);
            float opt; uint32_t optp;
            // what the pointer search below needs of a cell that a gap won (pcat 1: deletion, 2: far insertion)
            int pcat = 0, pk = 0, oa = a - 1, ob = bb - 1; bool pamb = false, pgen = false;
            float pgi = gi_c, pge = ge_c, pcn = 0.f;
            if (a == 1) {                                   // first row: dpmatrix.h:409-418
              float sv = 0.f;
              if (bb > 1) sv -= frame_del(e, f, 0, bb);
              sv += sim;
              opt = clip0(sv, LOCAL); optp = origin;
            } else if (bb == 1) {                           // first column: :421-426
              float sv = 0.f;
              sv -= frame_ins(e, f, 0, a, 0, 1);
              sv += sim;
              opt = clip0(sv, LOCAL); optp = origin;
            } else {
              // ---- deletion scan: far-left state, then the tile's own columns from LDS ------------------------------
              ScanState<1> s;
              if (TPOS) { const float2 t = tg[bb]; s.gib[0] = t.x; s.geb[0] = t.y; } else { s.gib[0] = 0.f; s.geb[0] = 0.f; }
              s.fd[0] = (float)(bb - kbase - 2);
              s.cm[0] = ninf;
              if (cb >= 1) { s.m[0] = aload(&fdm[r * SP + sb]); s.e[0] = aload(&fde[r * SP + sb]); s.cidx[0] = aloadi(&fdc[r * SP + sb]); }
              else { s.m[0] = ninf; s.e[0] = ninf; s.cidx[0] = 0; }
              const int tail = WF ? kbase : kbase + 64 * wave;
              if constexpr (TAB) {
                // the tile's own source columns: one table value per candidate, same 32-column chunks and (max, e, chunk) bookkeeping
                for (int kc = kbase; kc < tail + 64; kc += kBC) {
                  float cmx = ninf;
#pragma unroll 8
                  for (int u = 0; u < kBC; ++u) {
                    const int k = kc + u;
                    float d = prev[k] - del_at(k, bb);
                    d = (k <= bb - 2) ? d : ninf;
                    cmx = vmaxf(cmx, d);
                  }
                  const bool up = cmx > s.m[0];
                  s.e[0] = up ? s.m[0] : s.e[0];
                  s.cidx[0] = up ? kc : s.cidx[0];
                  s.m[0] = up ? cmx : s.m[0];
                }
              } else {
                scan_range<0, 1, TPOS, false>(s, prev, tg, kbase, tail, gi_c, ge_c);
                scan_range<0, 1, TPOS, true, true>(s, prev, tg, tail, tail + 64, gi_c, ge_c);
              }
              const float dm = s.m[0], de = s.e[0]; const int dc = s.cidx[0];
              // ---- insertions ---------------------------------------------------------------------------------------
              const size_t colb = (size_t)f.rt(bb - 1);
              const int kn0 = (a0 - 1 > 1) ? a0 - 1 : 1;
              float xn[kBR];
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                if constexpr (WF) xn[u] = (kn0 + u <= a - 2) ? rlw[(kn0 + u - (a0 - 1)) * kWRow + lt] : ninf;   // column b-1 = entry lt of the row's copy
                else xn[u] = (kn0 + u <= a - 2) ? aload(&H[(size_t)f.rq(kn0 + u) * ld + colb]) : ninf;
              }
              float mf = ninf, ef = ninf; int cf = 0;
              if (a0 >= 3) { mf = aload(&scr_m[r * SP + sb]); ef = aload(&scr_e[r * SP + sb]); cf = aloadi(&scr_c[r * SP + sb]); }
              opt = clip0(prev[bb - 1] + sim, LOCAL);          // match, :447-451
              int cat = 0;
              const float sd = clip0(dm + sim, LOCAL);
              if (sd > opt) { opt = sd; cat = 1; }
              const float sfar = clip0(mf + sim, LOCAL);
              float snear = ninf; int knear = 0;
              float gi = gi_c, ge = ge_c, cn = 0.f;
              if (TPOS) { const float2 t0 = tg[bb - 1], t1 = tg[bb]; gi = fminr(t0.x, t1.x); ge = fminr(t0.y, t1.y); }
              if (TAB) { const int lo = rev ? bb : bb - 1; const float2 t = tg[lo]; gi = t.x; ge = t.y; cn = tcnl[lo]; }
              auto ins_gap = [&](int n) -> float { float g = gi + ge * (float)n; if (TAB) g = g + cn; return g; };
#pragma unroll
              for (int u = 0; u < kBR; ++u) {
                if (kn0 + u <= a - 2) {
                  float sv = xn[u];
                  sv -= ins_gap(a - (kn0 + u) - 2);
                  sv += sim;
                  sv = clip0(sv, LOCAL);
                  if (sv > snear) { snear = sv; knear = kn0 + u; }
                }
              }
              const float si = (snear > sfar) ? snear : sfar;
              if (si > opt) { opt = si; cat = (snear > sfar) ? 3 : 2; }
              pgen = true; pgi = gi; pge = ge; pcn = cn;
              if (cat == 1) {
                pcat = 1; pamb = clip0(de + sim, LOCAL) == opt;
                pk = pamb ? 1 : (dc > 1 ? dc : 1);
              } else if (cat == 2) {
                pcat = 2; pamb = clip0(ef + sim, LOCAL) == opt;
                pk = pamb ? 1 : (cf > 1 ? cf : 1);
              } else if (cat == 3) {
                oa = knear; ob = bb - 1;
              }
            }
            // ---- which candidate: the reference keeps the FIRST k whose literal value equals the cell's (dpmatrix.h:447-486).  The
            // scans left (chunk, tie flag); inside the chunk the wave searches together: a cell's 32 (deletion) or 16 (insertion)
            // candidates are evaluated by as many lanes at once — two (four) cells per round, the loads of up to four (two) rounds
            // in flight together — instead of every lane walking its own chunk while the other 60 wait.  Same arithmetic, value by
            // value; cells with a tie (or, never seen, without a hit in their chunk) fall back to the literal walk.
            {
              const int lane = tid & 63;
              const float* prevF = rowsF + (size_t)((a - 1) & (RING - 1)) * PT;   // row a-1, every column, through L2
              const float gbi = TPOS ? tg[bb].x : 0.f, gbe = TPOS ? tg[bb].y : 0.f;
              bool pend = pcat != 0;
              auto fetch_i = [&](int sidx, int v) -> int { return __builtin_amdgcn_ds_bpermute(sidx, v); };
              auto fetch_f = [&](int sidx, float v) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sidx, __builtin_bit_cast(int, v))); };
              unsigned long long m1 = __ballot(pcat == 1 && !pamb);
              while (m1) {
                int La[4], Lb[4]; float pv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  La[q] = -1; Lb[q] = -1;
                  if (m1) { La[q] = __builtin_ctzll(m1); m1 &= m1 - 1; }
                  if (m1) { Lb[q] = __builtin_ctzll(m1); m1 &= m1 - 1; }
                  const int sl = lane < 32 ? La[q] : Lb[q];
                  const int sidx = (sl < 0 ? lane : sl) << 2;
                  const int kk = fetch_i(sidx, pk) + (lane & 31), bS = fetch_i(sidx, bb);
                  const bool live = sl >= 0 && kk >= 1 && kk <= bS - 2;
                  pv[q] = ninf;
                  if (live) pv[q] = (kk >= kbase) ? prev[kk] : aload(&prevF[kk]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (La[q] < 0) break;                            // wave-uniform
                  const int sl = lane < 32 ? La[q] : Lb[q];
                  const int sidx = (sl < 0 ? lane : sl) << 2;
                  const int kk = fetch_i(sidx, pk) + (lane & 31), bS = fetch_i(sidx, bb);
                  const bool live = sl >= 0 && kk >= 1 && kk <= bS - 2;
                  const float optS = fetch_f(sidx, opt), simS = fetch_f(sidx, sim);
                  const float2 tk = tg[live ? kk : 1];
                  float g = gi_c + ge_c * (float)(bS - kk - 2);
                  if (TPOS) g = fminr(tk.x, fetch_f(sidx, gbi)) + fminr(tk.y, fetch_f(sidx, gbe)) * (float)(bS - kk - 2);
                  if (TAB) g = live ? del_at(kk, bS) : 0.f;
                  float sv = pv[q];
                  sv -= g;
                  sv += simS;
                  sv = clip0(sv, LOCAL);
                  const unsigned long long hit = __ballot(live && sv == optS);
                  if (lane == La[q] && (unsigned)hit) { ob = pk + __builtin_ctz((unsigned)hit); pend = false; }
                  if (lane == Lb[q] && (unsigned)(hit >> 32)) { ob = pk + __builtin_ctz((unsigned)(hit >> 32)); pend = false; }
                }
              }
              unsigned long long m2 = __ballot(pcat == 2 && !pamb);
              while (m2) {
                int Lq[2][4]; float xv[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                  for (int w4 = 0; w4 < 4; ++w4) { Lq[q][w4] = -1; if (m2) { Lq[q][w4] = __builtin_ctzll(m2); m2 &= m2 - 1; } }
                  const int sl = lane < 16 ? Lq[q][0] : lane < 32 ? Lq[q][1] : lane < 48 ? Lq[q][2] : Lq[q][3];
                  const int sidx = (sl < 0 ? lane : sl) << 2;
                  const int kk = fetch_i(sidx, pk) + (lane & 15), bS = fetch_i(sidx, bb);
                  const bool live = sl >= 0 && kk <= a0 - 2;
                  xv[q] = ninf;
                  if (live) xv[q] = aload(&H[(size_t)f.rq(kk) * ld + (size_t)f.rt(bS - 1)]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  if (Lq[q][0] < 0) break;                         // wave-uniform
                  const int sl = lane < 16 ? Lq[q][0] : lane < 32 ? Lq[q][1] : lane < 48 ? Lq[q][2] : Lq[q][3];
                  const int sidx = (sl < 0 ? lane : sl) << 2;
                  const int kk = fetch_i(sidx, pk) + (lane & 15);
                  const bool live = sl >= 0 && kk <= a0 - 2;
                  const float optS = fetch_f(sidx, opt), simS = fetch_f(sidx, sim);
                  float g = fetch_f(sidx, pgi) + fetch_f(sidx, pge) * (float)(a - kk - 2);
                  if (TAB) g = g + fetch_f(sidx, pcn);
                  float sv = xv[q];
                  sv -= g;
                  sv += simS;
                  sv = clip0(sv, LOCAL);
                  const unsigned long long hit = __ballot(live && sv == optS);
#pragma unroll
                  for (int w4 = 0; w4 < 4; ++w4) {
                    const unsigned hh = (unsigned)(hit >> (16 * w4)) & 0xFFFFu;
                    if (lane == Lq[q][w4] && hh) { oa = pk + __builtin_ctz(hh); ob = bb - 1; pend = false; }
                  }
                }
              }
              if (pend && pcat == 1) {                              // literal walk (ties)
                auto lit = [&](int kk, float pvv) -> float {
                  const float2 tk = tg[kk];
                  float g = (TPOS ? fminr(tk.x, gbi) : gi_c) + (TPOS ? fminr(tk.y, gbe) : ge_c) * (float)(bb - kk - 2);
                  if (TAB) g = del_at(kk, bb);
                  float sv = pvv;
                  sv -= g;
                  sv += sim;
                  return clip0(sv, LOCAL);
                };
                int k = pk;
                for (; k < bb - 2; ++k) {
                  const float pvv = (k >= kbase) ? prev[k] : aload(&prevF[k]);
                  if (lit(k, pvv) == opt) break;
                }
                oa = a - 1; ob = k;
              } else if (pend && pcat == 2) {
                int k = pk;
                for (; k < a0 - 2; ++k) {
                  float sv = aload(&H[(size_t)f.rq(k) * ld + (size_t)f.rt(bb - 1)]);
                  float g = pgi + pge * (float)(a - k - 2);
                  if (TAB) g = g + pcn;
                  sv -= g;
                  sv += sim;
                  sv = clip0(sv, LOCAL);
                  if (sv == opt) break;
                }
                oa = k; ob = bb - 1;
              }
              if (pgen) optp = pack_ptr(f.rq(oa), f.rt(ob));
            }
            if (valid) {
              H[(size_t)i * ld + jj] = opt; P[(size_t)i * ld + jj] = optp;
              rowF[bc] = opt;
              curl[1 + lt] = opt;
              colmax = vmaxf(colmax, opt);
              const uint32_t pos = ((uint32_t)a << 16) | (uint32_t)bc;
              if (opt > lmax || (opt == lmax && pos < lpos)) { lmax = opt; lpos = pos; }   // tiles are not visited in row-major order
            } else {
              curl[1 + lt] = ninf;
            }
            if constexpr (WF) {
              // the row's maxima over this tile's two 32-column chunks (the skip tests of later far-left scans)
              if (prune) {
                float rv = valid ? opt : ninf;
#pragma unroll
                for (int o = 1; o <= 16; o <<= 1) rv = vmaxf(rv, __shfl_xor(rv, o));
                if ((lt & 31) == 0) delmaxF[(size_t)(kbase / 32 + (lt >> 5)) * RING + (a & (RING - 1))] = rv;
              }
            }
          } else {
            curl[1 + lt] = ninf;
          }
          if constexpr (WF) {
            if (lt == 0) curl[0] = edgeL[wave][r + 1];         // column kbase of row a (finished one step ago)
            // row a is in LDS for this wave's next rows (LDS executes a wave's accesses in order); nothing inside the tile reads
            // it back from L2, so its stores drain behind the next rows
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
          } else {
            if (tid == 0) curl[0] = (cb >= 1) ? aload(&rowF[kbase]) : ninf;     // column b0-1 of row a (finished in the previous tile)
            __threadfence_block();
            __syncthreads();
          }
        }
        // ============ tile epilogue: the maxima later far scans test their chunks against ==============================
        if (prune) {
          if (bv) insmaxF[(size_t)((a0 - 1) >> 4) * PT + bc] = colmax;          // this column over the rows of the block
          if constexpr (!WF) {
            // rows a0 .. a_end x this tile's 8 chunks [kbase + 32 j, kbase + 32 j + 31]: 16 threads per row, 2 per chunk
            const int r = tid >> 4, j = (tid & 15) >> 1, half = tid & 1;
            const int a = a0 + r;
            float mxv = ninf;
            if (a <= a_end) {
              const float* rowp = rowsF + (size_t)(a & (RING - 1)) * PT;
              const int k0 = kbase + 32 * j + 16 * half;
              float v[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) v[u] = (k0 + u >= 1 && k0 + u <= nT - 1) ? aload(&rowp[k0 + u]) : ninf;
#pragma unroll
              for (int u = 0; u < 16; ++u) mxv = vmaxf(mxv, v[u]);
            }
            mxv = vmaxf(mxv, __shfl_xor(mxv, 1));
            if (half == 0 && a <= a_end) delmaxF[(size_t)(kbase / 32 + j) * RING + (a & (RING - 1))] = mxv;
            __threadfence_block();
            __syncthreads();
          }
        }
      }
    };
    if constexpr (!WF) {
      for (int a0 = 1; a0 <= nQ - 1; a0 += kBR) {
        set_prio(a0);
        for (int cb = 0; cb < ntiles; ++cb) do_tile(a0, cb);
      }
    } else {
      // Wavefront over (row block, 64-column tile): wave w owns the row blocks w, w+4, w+8, ... and sweeps each left to right.  Its
      // tile (A, c) needs tile (A-1, c) — row a0-1 and the maxima of the block above — so it starts its j-th tile when the wave
      // that owns the block above has finished ITS j-th tile; wave 0 follows wave 3's previous block the same way (that needs
      // >= 4 tiles per sweep: narrower templates get empty ones).  Inside a tile nothing is shared between waves — the 64 columns'
      // own sources are this wave's previous rows, in LDS — so there is no workgroup barrier at all: a wave publishes the number
      // of tiles it has finished in LDS (after its stores have reached L2) and the next one polls that word.  Tile times differ
      // (which far chunks are scanned depends on the data); a wave may run up to a sweep ahead of its follower, which absorbs that.
      // No deadlock: tile j of wave w waits for a tile with a smaller (block, column) only, and every wave of the workgroup is resident.
      const int NB = (nQ - 1 + kBR - 1) / kBR;
      const int sweep = ntiles < 4 ? 4 : ntiles;
      const int my_tiles = (NB > wave ? (NB - wave + 3) / 4 : 0) * sweep;
      const int src = (wave + 3) & 3;                      // the wave that owns the block above
      for (int j = 0; j < my_tiles; ++j) {
        const int g = j / sweep, c = j - g * sweep;
        const int a0 = 1 + kBR * (wave + 4 * g);
        const int need = wave ? j + 1 : j - sweep + 1;     // tiles the wave above must have finished
        if (need > 0) {
          // (bounded: a wait of seconds can only be a defect — the pair's maximum is poisoned instead of the GPU hanging)
          int spins = 0;
          while (__hip_atomic_load(&wdone[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 24)) { lmax = __builtin_nanf(""); lpos = 0u; break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (c == 0) set_prio(a0);
        if (c < ntiles) do_tile(a0, c);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the tile's stores have been acknowledged by L2
        if (lt == 0) __hip_atomic_store(&wdone[wave], j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (dbg && (threadIdx.x & 63) == 0) {
    if (dbg[4] == 2) {                                  // exact_debug 2: how long the waves ran (sum, longest, ~shortest, count; 1024 ticks)
      const unsigned long long dur = (__builtin_amdgcn_s_memtime() - t_begin) >> 10;
      atomicAdd(&dbg[0], dur); atomicMax(&dbg[1], dur); atomicMax(&dbg[2], ~dur); atomicAdd(&dbg[3], 1ull);
      if (smax_arr && threadIdx.x == 0) const_cast<float*>(smax_arr)[blockIdx.x] = (float)dur;   // (read at the start; the host lists the slowest pairs)
    } else {
      atomicAdd(&dbg[0], (unsigned long long)n_tested_d); atomicAdd(&dbg[1], (unsigned long long)n_skip_d);
      atomicAdd(&dbg[2], (unsigned long long)n_tested_i); atomicAdd(&dbg[3], (unsigned long long)n_skip_i);
    }
  }
  float m = lmax; uint32_t p = lpos;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    float om = __shfl_xor(m, o); uint32_t op = (uint32_t)__shfl_xor((int)p, o);
    bool take = om > m || (om == m && op < p);
    m = take ? om : m; p = take ? op : p;
  }
  if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = m; red_p[threadIdx.x >> 6] = p; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kTW / 64; ++w) {
      bool take = red_v[w] > m || (red_v[w] == m && red_p[w] < p);
      if (take) { m = red_v[w]; p = red_p[w]; }
    }
    uint32_t rp = 0xFFFFFFFFu;
    if (p != 0xFFFFFFFFu && m > 0.f) rp = ((uint32_t)f.rq((int)(p >> 16)) << 16) | (uint32_t)f.rt((int)(p & 0xFFFFu));
    res[blockIdx.x].part_max = m;
    res[blockIdx.x].part_pos = rp;
  }
}

template <int NS>
static int launch_ns(aln_batch* b, const EvalDev& proto, bool tpos, bool sub, float* scratch) {
  aln_ctx* ctx = b->ctx;
  constexpr int PT = NS * 256 + kBPad;
  const size_t lds = ((size_t)PT * 4 + (size_t)3 * NS * kBT) * sizeof(float);
  const int rev = (int)(b->direction == ALN_REV);
#define ALN_XLAUNCH(TP, LC)                                                                                                   \
  hipLaunchKernelGGL((dp_exact_blocked_kernel<NS, TP, LC>), dim3(b->n_pairs), dim3(kBT), lds, ctx->stream, b->d_pairs, proto,   \
                     sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr,                      \
                     tpos ? b->d_tge : nullptr, b->d_H, b->d_P, sub ? nullptr : b->d_S, b->d_res, rev, scratch)
  if (tpos) { if (b->islocal) ALN_XLAUNCH(true, true); else ALN_XLAUNCH(true, false); }
  else { if (b->islocal) ALN_XLAUNCH(false, true); else ALN_XLAUNCH(false, false); }
#undef ALN_XLAUNCH
  ALN_HIP_CHECK(ctx, hipGetLastError());
  return ALN_OK;
}

// bytes of scratch one pair needs for a given slot count
static size_t blocked_scratch_floats(int ns) { return (size_t)3 * kBR * (ns * 256 + kBPad); }

bool dp_exact_blocked_legal(const aln_batch* b) {
  int mx = 0;
  for (const PairDesc& d : b->h_pairs) { const int nT = d.t1 - d.t0; if (nT - 1 > mx) mx = nT - 1; }
  return mx <= 16 * 256 && b->gapdev.model != ALN_GAP_TABLES;   // a plugin's fully tabulated gap functions run in the literal kernel
}

// max |S| over a pair's resident similarity plane (the rounding margin of the chunk-skipping tests)
__global__ __launch_bounds__(256) void plane_absmax_kernel(const PairDesc* __restrict__ pairs, const float* __restrict__ Sbase,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  const PairDesc pd = pairs[blockIdx.x];
  const float* S = Sbase + pd.plane_off;
  const size_t n = (size_t)pd.Q * pd.ld;
  float mx = 0.f;
  for (size_t k = threadIdx.x; k < n; k += 256) mx = fmaxf(mx, fabsf(S[k]));
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// tabR[x][y] = tab[T-1-y][T-1-x]: the deletion table of a template as a REVERSE build's frame sees it
__global__ void flip_transpose_kernel(const float* __restrict__ tab, float* __restrict__ tabR, int T) {
  __shared__ float tile[16][17];
  const int x0 = blockIdx.y * 16, y0 = blockIdx.x * 16;          // output tile: rows x0.., columns y0..
  // source element of output (x, y) is tab[T-1-y][T-1-x]: read with the source's column index (T-1-x) running over threadIdx.x
  const int sy = T - 1 - (y0 + threadIdx.y), sx = T - 1 - (x0 + threadIdx.x);
  if (sy >= 0 && sx >= 0) tile[threadIdx.y][threadIdx.x] = tab[(size_t)sy * T + sx];
  __syncthreads();
  const int x = x0 + threadIdx.y, y = y0 + threadIdx.x;
  if (x < T && y < T) tabR[(size_t)x * T + y] = tile[threadIdx.x][threadIdx.y];
}

int launch_dp_exact_blocked(aln_batch* b) {
  aln_ctx* ctx = b->ctx;
  int mx = 1;
  for (const PairDesc& d : b->h_pairs) { const int nT = d.t1 - d.t0; if (nT - 1 > mx) mx = nT - 1; }
  const int ns = mx <= 256 ? 1 : mx <= 512 ? 2 : mx <= 1024 ? 4 : 8;
  // templates wider than two tiles: the tiled kernel shares the far-left deletion scans between 16 rows
  const bool gn2 = b->gapdev.model == ALN_GAP_DEL_TABLE_INS_TPOS;
  const bool tiled = gn2 || (mx > 2 * kTW && (ctx->hints.exact_tiles || mx > 8 * kTW));   // the slot kernel ends at 8 x 256 columns and knows no tables
  const int ptt = (mx + 1 <= 4 * kTW ? 4 : mx + 1 <= 8 * kTW ? 8 : mx + 1 <= 12 * kTW ? 12 : 16) * kTW + kBPad;          // row pitch of the scratch rows: a compile-time constant of the kernel
  // tiled kernel: + one row of chunk maxima + one row of column maxima per 16-row block (the skip tests of the far scans) + max|S| per pair
  const int q_blocks = (b->maxQ + kBR - 1) / kBR + 1;
  // the wavefront form keeps 17 rows of every wave's tile in LDS; beyond 64 KB per workgroup (the widest templates with Gn2Eval's tables) the 256-column form runs
  const bool wavefront = ctx->hints.exact_wavefront != 0 && ((size_t)2 * ptt + 4 * kWHist + (gn2 ? ptt : 0)) * sizeof(float) <= 65536;
  const int ring = wavefront ? kWRing : kTRing;
  const size_t tiled_floats = (size_t)(6 * kBR + ring + 2 + ring / 32 + q_blocks) * ptt;     // per pair, as the kernel lays them out
  const size_t need = tiled ? tiled_floats * (size_t)b->n_pairs + (size_t)b->n_pairs + 64
                            : blocked_scratch_floats(ns) * (size_t)b->n_pairs;
  if (b->xscratch_floats < need) {
    if (b->d_xscratch) { ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); hipFree(b->d_xscratch); b->d_xscratch = nullptr; b->xscratch_floats = 0; }
    if (hipMalloc((void**)&b->d_xscratch, need * 4) != hipSuccess) { ctx->last_error = "hipMalloc (far-insertion scratch)"; return ALN_E_NOMEM; }
    b->xscratch_floats = need;
  }
  // untouched cells read score 0 / pointer (-1,-1) (dpmatrix.cpp:17-25): the kernel only writes computed cells
  ALN_HIP_CHECK(ctx, hipMemsetAsync(b->d_H, 0, (size_t)b->plane_elems * 4, ctx->stream));
  ALN_HIP_CHECK(ctx, hipMemsetAsync(b->d_P, 0xFF, (size_t)b->plane_elems * 4, ctx->stream));
  EvalDev proto = {};
  proto.model = b->gapdev.model;
  proto.align_type = b->gapdev.align_type;
  proto.gi = b->gapdev.gi; proto.ge = b->gapdev.ge;
  proto.sim_kind = (b->sim_kind == ALN_SIM_SUBMATRIX) ? ALN_SIM_SUBMATRIX : ALN_SIM_MATRIX;
  proto.tablef = b->d_tablef;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model == ALN_GAP_AFFINE_TPOS_MIN;
  int rc;
  if (tiled) {
    const size_t lds = ((size_t)2 * ptt + (wavefront ? 4 * kWHist : 2 * kTLoc) + (gn2 ? ptt : 0)) * sizeof(float);
    const int rev = (int)(b->direction == ALN_REV);
    // Gn2Eval's model: per pair the frame-ordered deletion table.  Forward builds read the caller's tables; reverse builds their
    // flipped transposes tabR[x][y] = tab[T-1-y][T-1-x] (made here, once per upload), so that a thread's column b stays the
    // fastest-running index in both.
    const float* delF = nullptr;
    if (gn2) {
      const int ns_t = (int)b->t_offsets.size() - 1;
      std::vector<int64_t> toff(ns_t);
      int64_t total = 0;
      for (int q = 0; q < ns_t; ++q) { const int64_t T = b->t_offsets[q + 1] - b->t_offsets[q]; toff[q] = total; total += T * T; }
      if (rev && !b->deltabR_valid) {
        if (!b->d_deltabR) ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_deltabR, (size_t)std::max<int64_t>(total, 1) * 4));
        for (int q = 0; q < ns_t; ++q) {
          const int T = (int)(b->t_offsets[q + 1] - b->t_offsets[q]);
          hipLaunchKernelGGL(flip_transpose_kernel, dim3((T + 15) / 16, (T + 15) / 16), dim3(16, 16), 0, ctx->stream,
                             b->d_deltab + toff[q], b->d_deltabR + toff[q], T);
        }
        ALN_HIP_CHECK(ctx, hipGetLastError());
        b->deltabR_valid = true;
      }
      delF = rev ? b->d_deltabR : b->d_deltab;
      if (!b->d_pair_deloff) {
        std::vector<int64_t> poff(b->n_pairs);
        for (int q = 0; q < b->n_pairs; ++q) poff[q] = toff[b->h_pairs[q].t_seq];
        ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_pair_deloff, (size_t)std::max(b->n_pairs, 1) * 8));
        ALN_HIP_CHECK(ctx, hipMemcpy(b->d_pair_deloff, poff.data(), (size_t)b->n_pairs * 8, hipMemcpyHostToDevice));
      }
      proto.tcn = b->d_tcn; proto.deltab = b->d_deltab; proto.deltab_off = b->d_deltab_off;
    }
    const int alt_prio = ctx->hints.exact_alt_prio;      // aln_ctx_set_hint "exact_alt_prio"
    // Skipping far chunks by bounds needs gaps that grow with the distance: extension coefficients >= 0 (checked when they were
    // uploaded) — and max|S| for the rounding margin: the table's for codes + table, a reduction over the resident plane otherwise.
    // hint exact_prune: 1 = both far scans, 2 = the far-left deletions only, 3 = the far insertions only (development), 0 = off
    const int hp = ctx->hints.exact_prune;
    const int prune = !b->gap_ext_nonneg ? 0 : hp == 1 ? 3 : hp == 2 ? 1 : hp == 3 ? 2 : 0;
    float* d_smax = nullptr;
    float smax_const = 0.f;
    if (sub) { for (float v : b->h_table) smax_const = std::max(smax_const, std::fabs(v)); }
    else if (prune && b->sabs_valid && ctx->hints.exact_debug != 2) {
      d_smax = b->d_sabs;                                  // left by hmap2_apply_kernel with the plane (exact_debug 2 borrows the array: own copy then)
    } else if (prune) {
      d_smax = b->d_xscratch + tiled_floats * (size_t)b->n_pairs;
      hipLaunchKernelGGL(plane_absmax_kernel, dim3(b->n_pairs), dim3(256), 0, ctx->stream, b->d_pairs, b->d_S, d_smax);
      ALN_HIP_CHECK(ctx, hipGetLastError());
    }
    // the launch's progress word (exact_alt_prio 2), in the scratch's tail beside the debug counters
    unsigned int* d_gprog = reinterpret_cast<unsigned int*>(b->d_xscratch + (need - 32));
    ALN_HIP_CHECK(ctx, hipMemsetAsync(d_gprog, 0, 4, ctx->stream));
    unsigned long long* d_dbg = nullptr;
    if (ctx->hints.exact_debug) {
      d_dbg = reinterpret_cast<unsigned long long*>(b->d_xscratch + ((need - 16) & ~(size_t)1));
      const unsigned long long init[5] = {0, 0, 0, 0, (unsigned long long)ctx->hints.exact_debug};
      ALN_HIP_CHECK(ctx, hipMemcpyAsync(d_dbg, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    }
#define ALN_TLAUNCH(PTC, GM_, LC)                                                                                                \
    if (wavefront) ALN_TLAUNCH_W(PTC, GM_, LC, true); else ALN_TLAUNCH_W(PTC, GM_, LC, false)
#define ALN_TLAUNCH_W(PTC, GM_, LC, WF_)                                                                                         \
    hipLaunchKernelGGL((dp_exact_tiled_kernel<PTC, GM_, LC, WF_>), dim3(b->n_pairs), dim3(kTW), lds, ctx->stream, b->d_pairs, proto,    \
                       sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, (tpos || gn2) ? b->d_tgi : nullptr,               \
                       (tpos || gn2) ? b->d_tge : nullptr, b->d_H, b->d_P, sub ? nullptr : b->d_S, b->d_res, rev, b->d_xscratch,   \
                       alt_prio, delF, b->d_pair_deloff, prune, q_blocks, d_smax, smax_const, d_dbg, d_gprog)
#define ALN_TLAUNCH_P(PTC)                                                                                                       \
    do { if (gn2) { if (b->islocal) { ALN_TLAUNCH(PTC, 2, true); } else { ALN_TLAUNCH(PTC, 2, false); } }                        \
         else if (tpos) { if (b->islocal) { ALN_TLAUNCH(PTC, 1, true); } else { ALN_TLAUNCH(PTC, 1, false); } }                  \
         else { if (b->islocal) { ALN_TLAUNCH(PTC, 0, true); } else { ALN_TLAUNCH(PTC, 0, false); } } } while (0)
    if (ptt == 4 * kTW + kBPad) ALN_TLAUNCH_P(4 * kTW + kBPad);
    else if (ptt == 8 * kTW + kBPad) ALN_TLAUNCH_P(8 * kTW + kBPad);
    else if (ptt == 12 * kTW + kBPad) ALN_TLAUNCH_P(12 * kTW + kBPad);
    else ALN_TLAUNCH_P(16 * kTW + kBPad);
#undef ALN_TLAUNCH_P
#undef ALN_TLAUNCH_W
#undef ALN_TLAUNCH
    ALN_HIP_CHECK(ctx, hipGetLastError());
    if (d_dbg) {                                         // chunks tested / skipped, far-left deletions then far insertions (per wave)
      ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->exact_stats, d_dbg, 32, hipMemcpyDeviceToHost, ctx->stream));
      ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->hints.exact_debug == 2 && d_smax) {
        std::vector<float> dur(b->n_pairs);
        ALN_HIP_CHECK(ctx, hipMemcpy(dur.data(), d_smax, (size_t)b->n_pairs * 4, hipMemcpyDeviceToHost));
        std::vector<int> idx(b->n_pairs);
        for (int q = 0; q < b->n_pairs; ++q) idx[q] = q;
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return dur[x] > dur[y]; });
        fprintf(stderr, "[exact_debug 2] slowest pairs (wave 0, 1024 ticks):");
        for (int q = 0; q < b->n_pairs && q < 12; ++q) fprintf(stderr, " %d:%.0f", idx[q], dur[idx[q]]);
        fprintf(stderr, "; median %.0f\n", dur[idx[b->n_pairs / 2]]);
      }
    }
    b->kernel_name = std::string("dp_exact_tiled_kernel<") + (gn2 ? "gn2tab," : tpos ? "tpos," : "const,") + (b->islocal ? "local" : "global") +
                     (b->direction == ALN_REV ? ",rev" : ",fwd") + (wavefront ? ",wavefront>" : ">");
    return ALN_OK;
  }
  switch (ns) {
    case 1: rc = launch_ns<1>(b, proto, tpos, sub, b->d_xscratch); break;
    case 2: rc = launch_ns<2>(b, proto, tpos, sub, b->d_xscratch); break;
    case 4: rc = launch_ns<4>(b, proto, tpos, sub, b->d_xscratch); break;
    default: rc = launch_ns<8>(b, proto, tpos, sub, b->d_xscratch); break;
  }
  if (rc) return rc;
  b->kernel_name = std::string("dp_exact_blocked_kernel<") + (tpos ? "tpos," : "const,") + (b->islocal ? "local" : "global") +
                   (b->direction == ALN_REV ? ",rev>" : ",fwd>");
  return ALN_OK;
}

}  // namespace aln
