// score_only.hip — all-vs-all local alignment SCORES, no planes (BASELINE config 5) (gfx950).
//
// The score Optimal reports for a local build is find_max over the matrix (optimal.h:90-93, :108-124): the
// maximum cell of the DPMatrix::build_forw_local_dpm_nonlinear_gaps recurrence (dpmatrix.h:538-689).  When nobody
// needs cells or pointers nothing per-cell has to touch HBM: one wave per (query, template) pair sweeps the rows
// with the whole state in VGPRs — the same collapsed recurrence as dp_affine_tag.hip, without tags:
//   E(j) by a DPP max-plus prefix scan of A(k) = D[i-1][k] + ge k, F by a per-column running max of D[k][c] + ge k,
//   best = max3(match, E, F) + S, clipped at 0, running maximum per lane.
// Algorithmic bytes per pair: |q| + |t| residue bytes in, 4 bytes out (~0 B/cell) — the kernel is
// bound by VALU issue (about 6 half-rate + 8 full-rate instructions per cell), not by HBM.
// Grid: x = template index, y = query index inside the caller's row block; templates are replicated on every GPU,
// query rows are what ranks shard.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "aln_internal.h"

namespace aln {

constexpr int kNegS = -(1 << 28);

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int sdpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int wave_incl_max_s(int v) {
  const int ident = (int)0x80000000;
  v = max(v, sdpp<0x111>(ident, v));
  v = max(v, sdpp<0x112>(ident, v));
  v = max(v, sdpp<0x114>(ident, v));
  v = max(v, sdpp<0x118>(ident, v));
  v = max(v, sdpp<0x142, 0xA>(ident, v));
  v = max(v, sdpp<0x143, 0xC>(ident, v));
  return v;
}

struct ScoreArgs {
  const uint8_t* qcodes; const int64_t* qoff;   // query pool, offsets (n_q + 1)
  const uint8_t* tcodes; const int64_t* toff;   // template pool
  const int32_t* table32;                       // 32 x 32
  const int32_t* tsel;                          // blockIdx.x -> template index (templates are launched by length class)
  const int32_t* qsel;                          // packed kernel: query rows of the slab sorted by length (neighbours share a wave)
  float* scores;                                // rows x n_t
  int q_begin, n_t;
  int gi, ge;
};

template <int R>
__global__ __launch_bounds__(64) void score_local_kernel(ScoreArgs a) {
  __shared__ int tab[32 * 32];
  const int lane = threadIdx.x;
  for (int k = lane; k < 32 * 32; k += 64) tab[k] = a.table32[k];
  __syncthreads();
  const int ti = a.tsel[blockIdx.x], qi = a.q_begin + blockIdx.y;
  const uint8_t* __restrict__ qc = a.qcodes + a.qoff[qi];
  const uint8_t* __restrict__ tc = a.tcodes + a.toff[ti];
  const int Q = (int)(a.qoff[qi + 1] - a.qoff[qi]), T = (int)(a.toff[ti + 1] - a.toff[ti]);
  const int gi = a.gi, ge = a.ge;
  const int cb = 4 * lane;
  const int gime = gi - ge;

  int code4[R][4], gec[R][4], ekc[R][4], inm[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int c = cb + 256 * r + x;
      int code = kCodeTail;
      if (c < T) code = tc[c];
      code4[r][x] = code * 4;
      gec[r][x] = ge * c;
      ekc[r][x] = ge * c + gime;                                      // E(c+1) = prefix max - ekc
      inm[r][x] = ((unsigned)(c - 1) < (unsigned)(T - 2)) ? -1 : 0;   // interior column: scores are >= 0, so "& mask" zeroes the rest
    }
  int d[R][4], gmx[R][4], cv[R], ak[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cv[r] = kNegS;
#pragma unroll
    for (int x = 0; x < 4; ++x) { d[r][x] = 0; gmx[r][x] = kNegS; }
  }
  int lmax = 0;
  auto tab_at = [&](int qrow, int c4) -> int {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(tab) + qrow + c4);
  };
  // prefix-scan preparation on the row held in d[] + running maximum
  auto finish_row = [&]() {
    int sk = kNegS;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int tk = kNegS;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int A = d[r][x] + gec[r][x];
        if (r == 0 && x == 0) A = (lane == 0) ? kNegS : A;   // column 0 is never a source
        ak[r][x] = A;
        tk = max(tk, A);
      }
      lmax = max(max(lmax, d[r][0]), d[r][1]);               // two v_max3 per group
      lmax = max(max(lmax, d[r][2]), d[r][3]);
      const int ik = wave_incl_max_s(tk);
      const int ek = sdpp<0x138>(kNegS, ik);
      cv[r] = max(sk, ek);
      sk = max(sk, __builtin_amdgcn_readlane(ik, 63));
    }
  };
  if (Q >= 3) {
    // row 1 (dpmatrix.h:579-590): local mode -> end gaps are free: clip(S[1][c])
    const int qrow = (int)qc[1] * 128;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int h = max(tab_at(qrow, code4[r][x]), 0);
        d[r][x] = h & inm[r][x];
        (void)c;
      }
    finish_row();
  }
  int qcode_next = (Q >= 4) ? (int)qc[2] : 0;
  for (int i = 2; i <= Q - 2; ++i) {                        // dpmatrix.h:607-649
    const int qrow = qcode_next * 128;
    if (i + 1 <= Q - 2) qcode_next = (int)qc[i + 1];
    const int roff = gi + ge * (i - 2);
    const int rowB = ge * (i - 1);
    int bk[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int pv = cv[r];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int m = d[r][x];
        const int A = ak[r][x];
        const int e = pv - ekc[r][x];
        const int f = gmx[r][x] - roff;
        bk[r][x] = max(max(m, e), f);
        pv = max(pv, A);
        gmx[r][x] = max(gmx[r][x], m + rowB);
      }
    }
    int prev_k = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int uk = sdpp<0x138>(0, bk[r][3]);
      if (r > 0) uk = (lane == 0) ? prev_k : uk;
      prev_k = __builtin_amdgcn_readlane(bk[r][3], 63);
      const bool masked = (r == 0) || (256 * (r + 1) > T - 1);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int s = tab_at(qrow, code4[r][x]);
        int h = max(((x == 0) ? uk : bk[r][x - 1]) + s, 0);
        if (r == 0 && x == 1) h = (c == 1) ? max(s, 0) : h;  // column 1 (lane 0 only): free insertion from the origin (:593-599)
        if (masked) h &= inm[r][x];                          // columns 0 and >= T-1 stay 0
        d[r][x] = h;
      }
    }
    finish_row();
  }
  int m = lmax;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
  if (lane == 0) a.scores[(size_t)blockIdx.y * a.n_t + ti] = (float)m;
}


// ---- the four non-local align types: the score Optimal reports is the FINAL cell's (optimal.h:56-74) ---------------------------
// Same row sweep without the clip: values may be negative, so columns outside the interior are kept at "minus infinity"
// instead of being masked to 0; row 1 and column 1 pay (or not: free end gaps, aasubalib.h:34-49,60-75) the gap from the origin
// (dpmatrix.h:409-426); the final cell (dpmatrix.h:505-534) is the best of the last interior cell, a deletion from the last
// interior row and an insertion from the last interior column, each free or priced by the align type.
template <int R>
__global__ __launch_bounds__(64) void score_global_kernel(ScoreArgs a, int free_del, int free_ins) {
  __shared__ int tab[32 * 32];
  const int lane = threadIdx.x;
  for (int k = lane; k < 32 * 32; k += 64) tab[k] = a.table32[k];
  __syncthreads();
  const int ti = a.tsel[blockIdx.x], qi = a.q_begin + blockIdx.y;
  const uint8_t* __restrict__ qc = a.qcodes + a.qoff[qi];
  const uint8_t* __restrict__ tc = a.tcodes + a.toff[ti];
  const int Q = (int)(a.qoff[qi + 1] - a.qoff[qi]), T = (int)(a.toff[ti + 1] - a.toff[ti]);
  const int gi = a.gi, ge = a.ge;
  float* out = &a.scores[(size_t)blockIdx.y * a.n_t + ti];
  // degenerate shortcuts (dpmatrix.h:375-390): no interior row or column -> one gap from the origin, never clipped
  if (Q == 2 || T == 2) {
    int cost = 0;
    if (Q == 2) { const int len = T - 2; cost = (len < 1 || free_del) ? 0 : gi + ge * (len - 1); }
    else { const int len = Q - 2; cost = (len < 1 || free_ins) ? 0 : gi + ge * (len - 1); }
    if (lane == 0) *out = (float)(-cost);
    return;
  }
  const int cb = 4 * lane;
  const int gime = gi - ge;
  const int cl = T - 2;                                            // last interior column; its (wave-uniform) slot and lane
  const int rs = cl / 256, xs = cl & 3, ls = (cl & 255) >> 2;

  int code4[R][4], gec[R][4], ekc[R][4]; bool in[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int c = cb + 256 * r + x;
      int code = kCodeTail;
      if (c < T) code = tc[c];
      code4[r][x] = code * 4;
      gec[r][x] = ge * c;
      ekc[r][x] = ge * c + gime;
      in[r][x] = (unsigned)(c - 1) < (unsigned)(T - 2);
    }
  int d[R][4], gmx[R][4], cv[R], ak[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cv[r] = kNegS;
#pragma unroll
    for (int x = 0; x < 4; ++x) { d[r][x] = kNegS; gmx[r][x] = kNegS; }
  }
  int clast = kNegS;                                               // max over rows of D[k][T-2] (free insertions into the final cell)
  auto tab_at = [&](int qrow, int c4) -> int {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(tab) + qrow + c4);
  };
  auto pick = [&](const int (&v)[R][4]) -> int {                   // this lane's value in slot (rs, xs)
    int o = kNegS;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) o = (r == rs && x == xs) ? v[r][x] : o;
    return o;
  };
  auto finish_row = [&]() {
    int sk = kNegS;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int tk = kNegS;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int A = d[r][x] + gec[r][x];                       // non-interior cells hold "minus infinity": never a source
        ak[r][x] = A;
        tk = max(tk, A);
      }
      const int ik = wave_incl_max_s(tk);
      const int ek = sdpp<0x138>(kNegS, ik);
      cv[r] = max(sk, ek);
      sk = max(sk, __builtin_amdgcn_readlane(ik, 63));
    }
    const int v = pick(d);
    clast = max(clast, lane == ls ? v : kNegS);
  };
  {
    // row 1 (dpmatrix.h:409-418): one deletion from the origin, free if the template's head gap is
    const int qrow = (int)qc[1] * 128;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int cost = (c >= 2 && !free_del) ? gi + ge * (c - 2) : 0;
        d[r][x] = in[r][x] ? tab_at(qrow, code4[r][x]) - cost : kNegS;
      }
    finish_row();
  }
  int qcode_next = (Q >= 4) ? (int)qc[2] : 0;
  for (int i = 2; i <= Q - 2; ++i) {                               // dpmatrix.h:447-486
    const int qrow = qcode_next * 128;
    if (i + 1 <= Q - 2) qcode_next = (int)qc[i + 1];
    const int roff = gi + ge * (i - 2);
    const int rowB = ge * (i - 1);
    const int col1 = free_ins ? 0 : roff;                          // column 1: one insertion from the origin (:421-426)
    int bk[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int pv = cv[r];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int m = d[r][x];
        const int A = ak[r][x];
        const int e = pv - ekc[r][x];
        const int f = gmx[r][x] - roff;
        bk[r][x] = max(max(m, e), f);
        pv = max(pv, A);
        gmx[r][x] = max(gmx[r][x], m + rowB);
      }
    }
    int prev_k = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int uk = sdpp<0x138>(0, bk[r][3]);
      if (r > 0) uk = (lane == 0) ? prev_k : uk;
      prev_k = __builtin_amdgcn_readlane(bk[r][3], 63);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int s = tab_at(qrow, code4[r][x]);
        int h = ((x == 0) ? uk : bk[r][x - 1]) + s;
        if (r == 0 && x == 1) h = (c == 1) ? s - col1 : h;
        d[r][x] = in[r][x] ? h : kNegS;
      }
    }
    finish_row();
  }
  // ---- the final cell (dpmatrix.h:505-534): row Q-2 is in d[], gmx holds rows <= Q-3, clast every row of column T-2 -----
  int best = (lane == ls) ? pick(d) : kNegS;                       // match: D[Q-2][T-2] (the final cell's similarity is 0)
  {
    int dl = kNegS;                                                // deletion from (Q-2, k), k = 1 .. T-2 (k = T-2 costs nothing)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int len = T - 2 - c;
        const int cost = (len < 1 || free_del) ? 0 : gi + ge * (len - 1);
        dl = max(dl, in[r][x] ? d[r][x] - cost : kNegS);
      }
    best = max(best, dl);
    int il;                                                        // insertion from (k, T-2), k = 1 .. Q-2
    if (free_ins) il = clast;
    else {
      const int g = pick(gmx);                                     // max over k <= Q-3 of D[k][T-2] + ge k
      il = (lane == ls && Q >= 4) ? g - (gi + ge * (Q - 3)) : kNegS;
    }
    best = max(best, il);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) best = max(best, __shfl_xor(best, o));
  if (lane == 0) *out = (float)best;
}


// ---- two queries per wave in packed 16-bit lanes --------------------------------------------------------------------
// v_max_i32 issues at half rate on gfx950 and so does v_pk_max_i16 — which does two.  When every intermediate fits in 15 bits
// (checked on the host) the low half of each register carries query A and the high half query B against the same template:
// the recurrence, the DPP prefix scans and the one-column shift act on both halves at once, the substitution score comes
// from a per-row table of packed pairs (tab[qA[i]][c], tab[qB[i]][c]) that 32 lanes rebuild in LDS for every row.
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(int v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ int as_i(s2 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ s2 dup2(int v) { return as_s2((v & 0xFFFF) | (v << 16)); }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
constexpr int kNeg16 = -12000;          // "minus infinity" of the packed kernel: below every real value, clear of int16 wrap-around

__device__ __forceinline__ s2 wave_incl_max_pk(s2 v) {
  const int ident = (int)0x80008000;
  v = pmax(v, as_s2(sdpp<0x111>(ident, as_i(v))));
  v = pmax(v, as_s2(sdpp<0x112>(ident, as_i(v))));
  v = pmax(v, as_s2(sdpp<0x114>(ident, as_i(v))));
  v = pmax(v, as_s2(sdpp<0x118>(ident, as_i(v))));
  v = pmax(v, as_s2(sdpp<0x142, 0xA>(ident, as_i(v))));
  v = pmax(v, as_s2(sdpp<0x143, 0xC>(ident, as_i(v))));
  return v;
}

template <int R>
__global__ __launch_bounds__(64) void score_local_pk_kernel(ScoreArgs a, int n_rows) {
  __shared__ int tab[32 * 32];
  __shared__ int prow[2][32];           // packed substitution row of the current query residues, double-buffered by row parity
  const int lane = threadIdx.x;
  for (int k = lane; k < 32 * 32; k += 64) tab[k] = a.table32[k];
  __syncthreads();
  const int ti = a.tsel[blockIdx.x];
  const int rowA = a.qsel[2 * blockIdx.y], rowB_ = a.qsel[(2 * blockIdx.y + 1 < n_rows) ? 2 * blockIdx.y + 1 : 2 * blockIdx.y];
  const int qiA = a.q_begin + rowA, qiB = a.q_begin + rowB_;
  const uint8_t* __restrict__ qcA = a.qcodes + a.qoff[qiA];
  const uint8_t* __restrict__ qcB = a.qcodes + a.qoff[qiB];
  const uint8_t* __restrict__ tc = a.tcodes + a.toff[ti];
  const int QA = (int)(a.qoff[qiA + 1] - a.qoff[qiA]), QB = (int)(a.qoff[qiB + 1] - a.qoff[qiB]);
  const int T = (int)(a.toff[ti + 1] - a.toff[ti]);
  const int Qm = QA > QB ? QA : QB, Qs = QA > QB ? QB : QA;
  const int gi = a.gi, ge = a.ge;
  const int cb = 4 * lane;
  const int gime = gi - ge;
  const s2 zero2 = dup2(0), neg2 = dup2(kNeg16);

  int code4[R][4], inm[R][4];
  s2 gec[R][4], ekc[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int c = cb + 256 * r + x;
      int code = kCodeTail;
      if (c < T) code = tc[c];
      code4[r][x] = code * 4;
      gec[r][x] = dup2(ge * c);
      ekc[r][x] = dup2(ge * c + gime);
      inm[r][x] = ((unsigned)(c - 1) < (unsigned)(T - 2)) ? -1 : 0;
    }
  s2 d[R][4], gmx[R][4], cv[R], ak[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cv[r] = neg2;
#pragma unroll
    for (int x = 0; x < 4; ++x) { d[r][x] = zero2; gmx[r][x] = neg2; ak[r][x] = neg2; }
  }
  s2 lmax = zero2, snap = zero2;
  // the packed substitution row of query row i: lanes 0..31 build it (residue indices clamped for the shorter query)
  auto build_row = [&](int i) {
    if (lane < 32) {
      const int ia = i < QA ? i : QA - 1, ib = i < QB ? i : QB - 1;
      const int va = tab[(int)qcA[ia] * 32 + lane], vb = tab[(int)qcB[ib] * 32 + lane];
      prow[i & 1][lane] = (va & 0xFFFF) | (vb << 16);
    }
    __syncthreads();
  };
  auto row_at = [&](int i, int c4) -> s2 {
    return as_s2(*reinterpret_cast<const int*>(reinterpret_cast<const char*>(prow[i & 1]) + c4));
  };
  auto finish_row = [&]() {
    s2 sk = neg2;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s2 tk = neg2;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        s2 A = d[r][x] + gec[r][x];
        if (r == 0 && x == 0) A = (lane == 0) ? neg2 : A;    // column 0 is never a source
        ak[r][x] = A;
        tk = pmax(tk, A);
      }
      lmax = pmax(pmax(lmax, pmax(d[r][0], d[r][1])), pmax(d[r][2], d[r][3]));
      const s2 ik = wave_incl_max_pk(tk);
      const s2 ek = as_s2(sdpp<0x138>(as_i(neg2), as_i(ik)));
      cv[r] = pmax(sk, ek);
      sk = pmax(sk, as_s2(__builtin_amdgcn_readlane(as_i(ik), 63)));
    }
  };
  if (Qs < 3) snap = zero2;                                  // a query without interior rows scores 0
  if (Qm >= 3) {
    build_row(1);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const s2 h = pmax(row_at(1, code4[r][x]), zero2);
        d[r][x] = as_s2(as_i(h) & inm[r][x]);
      }
    finish_row();
    if (Qs - 2 == 1) snap = lmax;
  }
  for (int i = 2; i <= Qm - 2; ++i) {
    build_row(i);
    const s2 roff = dup2(gi + ge * (i - 2));
    const s2 rowB = dup2(ge * (i - 1));
    s2 bk[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s2 pv = cv[r];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const s2 m = d[r][x];
        const s2 e = pv - ekc[r][x];
        const s2 f = gmx[r][x] - roff;
        bk[r][x] = pmax(pmax(m, e), f);
        pv = pmax(pv, ak[r][x]);
        gmx[r][x] = pmax(gmx[r][x], m + rowB);
      }
    }
    int prev_k = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int uk = sdpp<0x138>(0, as_i(bk[r][3]));
      if (r > 0) uk = (lane == 0) ? prev_k : uk;
      prev_k = __builtin_amdgcn_readlane(as_i(bk[r][3]), 63);
      const bool masked = (r == 0) || (256 * (r + 1) > T - 1);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const s2 sv = row_at(i, code4[r][x]);
        s2 h = pmax(((x == 0) ? as_s2(uk) : bk[r][x - 1]) + sv, zero2);
        if (r == 0 && x == 1) h = (c == 1) ? pmax(sv, zero2) : h;
        if (masked) h = as_s2(as_i(h) & inm[r][x]);
        d[r][x] = h;
      }
    }
    finish_row();
    if (i == Qs - 2) snap = lmax;                            // the shorter query ends here; later rows of its half are not its own
  }
  // the longer query's half of lmax, the shorter one's half of snap
  const int full = as_i(lmax), part = as_i(snap);
  int mA = (short)(((QA >= QB) ? full : part) & 0xFFFF);
  int mB = (short)((((QB >= QA) ? full : part) >> 16) & 0xFFFF);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { mA = max(mA, __shfl_xor(mA, o)); mB = max(mB, __shfl_xor(mB, o)); }
  if (lane == 0) {
    a.scores[(size_t)rowA * a.n_t + ti] = (float)mA;
    if (rowB_ != rowA) a.scores[(size_t)rowB_ * a.n_t + ti] = (float)mB;
  }
}

}  // namespace aln

using namespace aln;

// The general route: scores[(q - q_begin) * n_t + t] for the templates listed in `tlist`, through resident batches of full
// DP builds (aln_batch_dp picks the kernel: tagged keys, int32 rows, exact-order scans) + Optimal's score — what the reference does
// for every pair, kept for what the register-resident kernels above do not take: templates beyond 2048 columns, fractional tables
// or gaps.  Pairs are grouped so that one group's planes stay below ~12 GB.
static int score_through_batches(aln_ctx* ctx, const aln_seqs* queries, const aln_seqs* templates, const aln_submatrix* sub,
                                 const aln_gap* gap, int32_t q_begin, int32_t q_end, const std::vector<int32_t>& tlist, float* scores) {
  const int n_t = templates->n_seqs;
  const size_t budget = (size_t)12 << 30;
  aln_sim sim = aln_sim();
  sim.kind = ALN_SIM_SUBMATRIX;
  sim.sub = *sub;
  std::vector<int32_t> qi, tix;
  std::vector<float> sc;
  std::vector<int32_t> st;
  auto flush = [&]() -> int {
    if (qi.empty()) return ALN_OK;
    aln_batch* bb = nullptr;
    int rc = aln_batch_create(ctx, queries, templates, (int32_t)qi.size(), qi.data(), tix.data(), 0, &bb);
    if (rc == ALN_OK) rc = aln_batch_dp(bb, &sim, gap, ALN_FWD, ALN_DP_AUTO, 0);
    sc.resize(qi.size()); st.resize(qi.size());
    if (rc == ALN_OK) rc = aln_batch_optimal(bb, sc.data(), nullptr, nullptr, 0, st.data());
    if (bb) aln_batch_destroy(bb);
    if (rc != ALN_OK) return rc;
    for (size_t k = 0; k < qi.size(); ++k) {
      if (st[k] != 0) return st[k];
      scores[(size_t)(qi[k] - q_begin) * n_t + tix[k]] = sc[k];
    }
    qi.clear(); tix.clear();
    return ALN_OK;
  };
  size_t bytes = 0;
  for (int32_t t : tlist) {
    const size_t T = (size_t)(templates->offsets[t + 1] - templates->offsets[t]);
    for (int32_t q = q_begin; q < q_end; ++q) {
      const size_t Q = (size_t)(queries->offsets[q + 1] - queries->offsets[q]);
      const size_t need = Q * (T + 16) * 8;
      if (!qi.empty() && bytes + need > budget) { int rc = flush(); if (rc) return rc; bytes = 0; }
      qi.push_back(q); tix.push_back(t); bytes += need;
    }
  }
  return flush();
}

// The score Optimal reports for queries[q_begin .. q_end) against every template: scores[(q - q_begin) * n_t + t].
// Replaces (q_end - q_begin) x n_t constructions of DPMatrix(q, t, AASubstitutionEval, fwd, align_type) + Optimal(align_type):
// find_max for local alignments, the final cell's score for the four other align types.
extern "C" int aln_score_all_vs_all(aln_ctx* ctx, const aln_seqs* queries, const aln_seqs* templates, const aln_submatrix* sub,
                                    const aln_gap* gap, int32_t q_begin, int32_t q_end, float* scores) {
  if (!ctx || !queries || !templates || !sub || !gap || !scores) return ALN_E_ARG;
  if (q_begin < 0 || q_end > queries->n_seqs || q_begin > q_end) return ALN_E_ARG;
  if (gap->model != ALN_GAP_AFFINE_CONST || gap->align_type < 0 || gap->align_type > 4) return ALN_E_ARG;
  const bool local = gap->align_type == ALN_LOCAL;
  const int free_del = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_LOCAL_GLOBAL);
  const int free_ins = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_GLOBAL_LOCAL);
  if (!sub->alphabet || !sub->table || sub->n < 1 || sub->n > 30) return ALN_E_ARG;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const float gi = gap->gap_init, ge = gap->gap_extn;
  std::vector<int32_t> every_t((size_t)templates->n_seqs);
  for (int t = 0; t < templates->n_seqs; ++t) every_t[t] = t;
  if (q_begin == q_end || templates->n_seqs == 0) return ALN_OK;
  // fractional gaps or table values: full builds in the exact-order kernels (the reference's arithmetic), batch by batch
  if (!(gi == (float)(int)gi) || !(ge == (float)(int)ge) || gi < 0 || ge < 0)
    return score_through_batches(ctx, queries, templates, sub, gap, q_begin, q_end, every_t, scores);
  int idx[256];
  for (int i = 0; i < 256; ++i) idx[i] = -1;
  for (int i = 0; i < sub->n; ++i) idx[(unsigned char)sub->alphabet[i]] = i;
  int32_t ti[32 * 32];
  double maxs = 0;
  for (int i = 0; i < 32 * 32; ++i) ti[i] = 0;
  for (int i = 0; i < sub->n; ++i)
    for (int j = 0; j < sub->n; ++j) {
      float v = sub->table[i * sub->n + j];
      if (!(v == (float)(int)v)) return score_through_batches(ctx, queries, templates, sub, gap, q_begin, q_end, every_t, scores);
      ti[i * 32 + j] = (int32_t)v;
      maxs = std::max(maxs, fabs((double)v));
    }
  auto encode = [&](const aln_seqs* s, std::vector<uint8_t>& codes, int& maxlen) -> int {
    const int64_t total = s->offsets[s->n_seqs];
    codes.resize((size_t)total);
    for (int64_t k = 0; k < total; ++k) {
      unsigned char ch = (unsigned char)s->residues[k];
      int c = (ch == '^') ? kCodeHead : (ch == '$') ? kCodeTail : idx[ch];
      if (c < 0) return ALN_E_RESIDUE;
      codes[(size_t)k] = (uint8_t)c;
    }
    maxlen = 0;
    for (int i = 0; i < s->n_seqs; ++i) {
      int64_t len = s->offsets[i + 1] - s->offsets[i];
      if (len < 2) return ALN_E_ARG;
      maxlen = std::max<int>(maxlen, (int)len);
    }
    return ALN_OK;
  };
  std::vector<uint8_t> qc, tc;
  int maxQ = 0, maxT = 0, rc;
  if ((rc = encode(queries, qc, maxQ)) != ALN_OK) return rc;
  if ((rc = encode(templates, tc, maxT)) != ALN_OK) return rc;
  if (maxT > kMaxLen || maxQ > kMaxLen) return ALN_E_TOO_LONG;
  if ((maxs + ge) * ((double)maxQ + std::min(maxT, 2048)) + gi + maxs >= 8388608.0)
    return score_through_batches(ctx, queries, templates, sub, gap, q_begin, q_end, every_t, scores);
  const int rows = q_end - q_begin, n_t = templates->n_seqs;
  if (rows == 0 || n_t == 0) return ALN_OK;

  ScoreArgs a = {};
  uint8_t *dq = nullptr, *dt = nullptr; int64_t *dqo = nullptr, *dto = nullptr; int32_t* dtab = nullptr; float* dsc = nullptr; int32_t* dsel = nullptr; int32_t* dqsel = nullptr;
  auto cleanup = [&]() { hipFree(dq); hipFree(dt); hipFree(dqo); hipFree(dto); hipFree(dtab); hipFree(dsc); hipFree(dsel); hipFree(dqsel); };
#define STRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return ALN_E_HIP; } } while (0)
  STRY(hipMalloc((void**)&dq, qc.size())); STRY(hipMalloc((void**)&dt, tc.size()));
  STRY(hipMalloc((void**)&dqo, (size_t)(queries->n_seqs + 1) * 8)); STRY(hipMalloc((void**)&dto, (size_t)(n_t + 1) * 8));
  STRY(hipMalloc((void**)&dtab, sizeof ti)); STRY(hipMalloc((void**)&dsc, (size_t)rows * n_t * 4));
  STRY(hipMemcpyAsync(dq, qc.data(), qc.size(), hipMemcpyHostToDevice, ctx->stream));
  STRY(hipMemcpyAsync(dt, tc.data(), tc.size(), hipMemcpyHostToDevice, ctx->stream));
  STRY(hipMemcpyAsync(dqo, queries->offsets, (size_t)(queries->n_seqs + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  STRY(hipMemcpyAsync(dto, templates->offsets, (size_t)(n_t + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  STRY(hipMemcpyAsync(dtab, ti, sizeof ti, hipMemcpyHostToDevice, ctx->stream));
  a.qcodes = dq; a.qoff = dqo; a.tcodes = dt; a.toff = dto; a.table32 = dtab; a.scores = dsc;
  a.q_begin = q_begin; a.n_t = n_t; a.gi = (int)gi; a.ge = (int)ge;
  // Templates are launched by length class: a wave sweeps 256 R columns, so a template of T columns needs
  // R = ceil(T / 256) groups; one launch per class keeps short templates from paying for the longest one.
  std::vector<int32_t> order, long_t; std::vector<int> cls_begin(10, 0);
  {
    std::vector<std::vector<int32_t>> by(9);
    for (int t = 0; t < n_t; ++t) {
      const int T = (int)(templates->offsets[t + 1] - templates->offsets[t]);
      if (T > 2048) long_t.push_back(t);            // beyond the register-resident kernels: full builds below
      else by[(T + 255) / 256].push_back(t);
    }
    for (int r = 1; r <= 8; ++r) { cls_begin[r] = (int)order.size(); order.insert(order.end(), by[r].begin(), by[r].end()); }
    cls_begin[9] = (int)order.size();
  }
  STRY(hipMalloc((void**)&dsel, (size_t)n_t * 4));
  if (!order.empty()) STRY(hipMemcpyAsync(dsel, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  // packed 16-bit lanes (two queries per wave) when every intermediate provably fits: best local score <= maxs * min(Q,T),
  // A keys add ge * column, the "minus infinity" -12000 must stay below every real candidate and clear of wrap-around
  const int fastT = std::min(maxT, 2048);                // (longer templates do not run in these kernels)
  const double L = (double)std::max(maxQ, fastT), best = maxs * (double)std::min(maxQ, fastT);
  const bool packed = local && best + ge * L + maxs < 30000.0 && ge * L + gi + maxs < 8000.0 && maxs < 2048.0 && ctx->hints.score_packed;
  const dim3 block(64);
  // blockIdx.y is limited to 65535: walk the query rows in slabs.  The packed kernel pairs queries of similar length (a wave
  // runs to the longer one's last row): every slab's length order goes to the device ONCE, before the first launch, into its own
  // region of dqsel (slab starting at row r0 -> dqsel + r0), so no launch can see another slab's order (the kernels run
  // asynchronously on ctx->stream) and the host vector lives until the final synchronisation.
  std::vector<int32_t> qo_all;
  if (packed) {
    qo_all.resize((size_t)rows);
    for (int r0 = 0; r0 < rows; r0 += 32768) {
      const int nr = std::min(32768, rows - r0);
      int32_t* qo = qo_all.data() + r0;
      for (int k = 0; k < nr; ++k) qo[k] = k;
      std::stable_sort(qo, qo + nr, [&](int32_t x, int32_t y) {
        return queries->offsets[q_begin + r0 + x + 1] - queries->offsets[q_begin + r0 + x] <
               queries->offsets[q_begin + r0 + y + 1] - queries->offsets[q_begin + r0 + y];
      });
    }
    STRY(hipMalloc((void**)&dqsel, (size_t)rows * 4));
    STRY(hipMemcpyAsync(dqsel, qo_all.data(), (size_t)rows * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  for (int r0 = 0; r0 < rows; r0 += 32768) {
    const int nr = std::min(32768, rows - r0);
    for (int r = 1; r <= 8; ++r) {
      const int nc = cls_begin[r + 1] - cls_begin[r];
      if (nc == 0) continue;
      ScoreArgs s = a;
      s.q_begin = q_begin + r0;
      s.scores = dsc + (size_t)r0 * n_t;
      s.tsel = dsel + cls_begin[r];
      s.qsel = dqsel ? dqsel + r0 : nullptr;
      if (packed) {
        const dim3 grid(nc, (nr + 1) / 2);             // two query rows per wave
        switch (r) {
          case 1: hipLaunchKernelGGL(score_local_pk_kernel<1>, grid, block, 0, ctx->stream, s, nr); break;
          case 2: hipLaunchKernelGGL(score_local_pk_kernel<2>, grid, block, 0, ctx->stream, s, nr); break;
          case 3: hipLaunchKernelGGL(score_local_pk_kernel<3>, grid, block, 0, ctx->stream, s, nr); break;
          case 4: hipLaunchKernelGGL(score_local_pk_kernel<4>, grid, block, 0, ctx->stream, s, nr); break;
          case 5: hipLaunchKernelGGL(score_local_pk_kernel<5>, grid, block, 0, ctx->stream, s, nr); break;
          case 6: hipLaunchKernelGGL(score_local_pk_kernel<6>, grid, block, 0, ctx->stream, s, nr); break;
          case 7: hipLaunchKernelGGL(score_local_pk_kernel<7>, grid, block, 0, ctx->stream, s, nr); break;
          default: hipLaunchKernelGGL(score_local_pk_kernel<8>, grid, block, 0, ctx->stream, s, nr); break;
        }
      } else if (!local) {
        const dim3 grid(nc, nr);
        switch (r) {
          case 1: hipLaunchKernelGGL(score_global_kernel<1>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 2: hipLaunchKernelGGL(score_global_kernel<2>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 3: hipLaunchKernelGGL(score_global_kernel<3>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 4: hipLaunchKernelGGL(score_global_kernel<4>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 5: hipLaunchKernelGGL(score_global_kernel<5>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 6: hipLaunchKernelGGL(score_global_kernel<6>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          case 7: hipLaunchKernelGGL(score_global_kernel<7>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
          default: hipLaunchKernelGGL(score_global_kernel<8>, grid, block, 0, ctx->stream, s, free_del, free_ins); break;
        }
      } else {
      const dim3 grid(nc, nr);
      switch (r) {
        case 1: hipLaunchKernelGGL(score_local_kernel<1>, grid, block, 0, ctx->stream, s); break;
        case 2: hipLaunchKernelGGL(score_local_kernel<2>, grid, block, 0, ctx->stream, s); break;
        case 3: hipLaunchKernelGGL(score_local_kernel<3>, grid, block, 0, ctx->stream, s); break;
        case 4: hipLaunchKernelGGL(score_local_kernel<4>, grid, block, 0, ctx->stream, s); break;
        case 5: hipLaunchKernelGGL(score_local_kernel<5>, grid, block, 0, ctx->stream, s); break;
        case 6: hipLaunchKernelGGL(score_local_kernel<6>, grid, block, 0, ctx->stream, s); break;
        case 7: hipLaunchKernelGGL(score_local_kernel<7>, grid, block, 0, ctx->stream, s); break;
        default: hipLaunchKernelGGL(score_local_kernel<8>, grid, block, 0, ctx->stream, s); break;
      }
      }
      STRY(hipGetLastError());
    }
  }
  STRY(hipMemcpyAsync(scores, dsc, (size_t)rows * n_t * 4, hipMemcpyDeviceToHost, ctx->stream));
  STRY(hipStreamSynchronize(ctx->stream));
#undef STRY
  cleanup();
  if (!long_t.empty()) return score_through_batches(ctx, queries, templates, sub, gap, q_begin, q_end, long_t, scores);
  return ALN_OK;
}
