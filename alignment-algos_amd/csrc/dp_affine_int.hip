// dp_affine_int.hip — O(n^2) row-sweep DP for constant affine gaps with integer-valued scores (gfx950).
//
// Replaces DPMatrix::build_forw_dpm_nonlinear_gaps / build_forw_local_dpm_nonlinear_gaps
// (reference dpmatrix.h:356-536, :538-689) for AASubstitutionEval-style gaps (aasubalib.h:27-77)
// when every similarity and gap value is an integer small enough that the reference's fp32
// arithmetic is exact: then the O(n^3) predecessor scans collapse to
//   E_i(j) = max_{1<=k<=j-2} D[i-1][k] - gi - ge (j-2-k)   (deletions, row i-1)
//   F_j(i) = max_{1<=k<=i-2} D[k][j-1] - gi - ge (i-2-k)   (insertions, column j-1)
// with the reference's tie-breaking (match, then deletions k ascending, then insertions k
// ascending, replace only on strict '>') kept by carrying the EARLIEST arg-max.
//
// Mapping (row sweep, SURVEY.md A.5: a row depends only on finished rows):
//   * one workgroup of NW waves per pair; a wave owns 256*R consecutive columns; lane l owns the
//     16-byte groups {4(l+64r) .. +3}, r < R, so every row store is one fully coalesced 1 KiB
//     global_store_dwordx4 per group (fp32 score plane + packed pointer plane: 8 B/cell, HBM-write bound);
//   * D[i-1][.] and the per-column insertion state (running max of D[k][c]+ge*k and its first k) live in VGPRs;
//   * deletions: with A(k) = D[i-1][k] + ge*k, E_i(j) = prefmax_{k<=j-2} A(k) - gi - ge (j-2): a max-plus
//     prefix scan along the row done with DPP row_shr/row_bcast steps (value scan + "last strict record"
//     scan for the first arg-max), chained across a lane's R groups with scalar carries and across waves
//     through 16 bytes of LDS per wave and ONE barrier per row;
//   * the substitution row for the current query residue is a 32-entry LDS row (conflict-free gather);
//   * no MFMA: this is a scalar max-plus recurrence, not a contraction.
// Source column c produces target column c+1; (best, pointer) are shifted one column right with DPP
// wave_shr before the similarity of the target column is added, so stores stay 16-byte aligned.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "aln_internal.h"

namespace aln {

struct FastParams {
  int gi, ge;
  int free_del, free_ins;
};

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp_mov(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive max-scan over the 64 lanes of a wave; lanes without a source contribute `ident`
__device__ __forceinline__ int wave_incl_max(int v, int ident) {
  v = max(v, dpp_mov<0x111>(ident, v));          // row_shr:1
  v = max(v, dpp_mov<0x112>(ident, v));          // row_shr:2
  v = max(v, dpp_mov<0x114>(ident, v));          // row_shr:4
  v = max(v, dpp_mov<0x118>(ident, v));          // row_shr:8
  v = max(v, dpp_mov<0x142, 0xA>(ident, v));     // row_bcast:15 -> rows 1,3
  v = max(v, dpp_mov<0x143, 0xC>(ident, v));     // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ int wave_shr1(int v, int lane0) { return dpp_mov<0x138>(lane0, v); }   // wave_shr:1

// (v,a) <- later candidate (nv,na) only if strictly greater: the earlier arg wins ties (dpmatrix.h:463 "s > opt_s")
__device__ __forceinline__ void take_later(int& v, int& a, int nv, int na) {
  bool gt = nv > v;
  v = max(v, nv);
  a = gt ? na : a;
}

template <int NW, int R, bool LOCAL, bool SIMPLANE>
__global__ __launch_bounds__(64 * NW) void dp_affine_int_kernel(
    const PairDesc* __restrict__ pairs, const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
    const int32_t* __restrict__ table32, float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
    const float* __restrict__ Sbase, PairResult* __restrict__ res, FastParams prm) {
  __shared__ int tab[SIMPLANE ? 1 : 32 * 32];
  __shared__ int xch[2][NW][4];
  __shared__ int red[NW][2];

  const PairDesc pd = pairs[blockIdx.x];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int W0 = w * 256 * R;
  const int cb = W0 + 4 * lane;
  const int gi = prm.gi, ge = prm.ge;
  const int gime = gi - ge;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint32_t* __restrict__ P = Pbase + pd.plane_off;
  const float* __restrict__ S = SIMPLANE ? (Sbase + pd.plane_off) : nullptr;
  const uint8_t* __restrict__ qc = qcodes + pd.q_off;
  const uint8_t* __restrict__ tc = tcodes + pd.t_off;

  if (!SIMPLANE) {
    for (int k = threadIdx.x; k < 32 * 32; k += 64 * NW) tab[k] = table32[k];
    __syncthreads();
  }

  // ---- static per-column data ---------------------------------------------------------------
  int code4[R][4];
  bool inrange[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    inrange[r] = (cb + 256 * r) < ld;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      int c = cb + 256 * r + x;
      int code = kCodeTail;
      if (!SIMPLANE && c < T) code = tc[c];
      code4[r][x] = code * 4;
    }
  }
  const int CB = W0 + 256 * R;     // first column of the next wave = this wave's boundary target
  int codeB4 = kCodeTail * 4;
  if (NW > 1 && !SIMPLANE && CB < T) codeB4 = tc[CB] * 4;

  const int gecb = ge * cb;
  int d[R][4], gmx[R][4], gar[R][4];
  int cv[R], ca[R];
  uint32_t pf[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cv[r] = kNeg; ca[r] = -1;
#pragma unroll
    for (int x = 0; x < 4; ++x) { d[r][x] = 0; gmx[r][x] = kNeg; gar[r][x] = 0; pf[r][x] = kNullPtr; }
  }
  int lmax = 0; uint32_t lpos = 0;   // LOCAL: per-lane maximum over interior cells and its first row-major position
  int par = 0;

  auto store_row = [&](int i) {
    const size_t ro = (size_t)i * ld + cb;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (inrange[r]) {
        float4 hv = make_float4((float)d[r][0], (float)d[r][1], (float)d[r][2], (float)d[r][3]);
        uint4 pv = make_uint4(pf[r][0], pf[r][1], pf[r][2], pf[r][3]);
        *reinterpret_cast<float4*>(H + ro + 256 * r) = hv;
        *reinterpret_cast<uint4*>(P + ro + 256 * r) = pv;
      }
    }
  };
  auto tab_at = [&](int qrow, int c4) -> int {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(tab) + qrow + c4);
  };

  // Finish row i held in d[]/pf[] (complete except, for w > 0, the wave's first column which the previous wave
  // computed and passes as (hB,pB)): prefix-scan preparation for the next row, exchange, local-max tracking, store.
  auto finish_row = [&](int i, int hB, uint32_t pB) {
    // lane-exclusive prefix (A-space value, arg column) per group, EXCLUDING the wave's first column
    int sv = kNeg, sa = -1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int tv = kNeg, ta = -1;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int c = cb + 256 * r + x;
        int A = d[r][x] + gecb + ge * (256 * r + x);
        if (r == 0 && x == 0) A = (lane == 0) ? kNeg : A;   // column 0 is never a source; wave firsts are folded below
        take_later(tv, ta, A, c);
      }
      int iv = wave_incl_max(tv, kNeg);
      int ev = wave_shr1(iv, kNeg);
      int rec = (tv > ev) ? ta : -1;            // strict record => this lane holds the first arg-max so far
      int ia = wave_incl_max(rec, -1);
      int ea = wave_shr1(ia, -1);
      bool gt = ev > sv;                        // earlier groups (scalar carry) win ties
      cv[r] = max(sv, ev);
      ca[r] = gt ? ea : sa;
      int gv = __builtin_amdgcn_readlane(iv, 63), ga = __builtin_amdgcn_readlane(ia, 63);
      take_later(sv, sa, gv, ga);
    }
    if (NW > 1) {
      if (lane == 63) { xch[par][w][0] = sv; xch[par][w][1] = sa; xch[par][w][2] = hB; xch[par][w][3] = (int)pB; }
      // LDS-only barrier: a __syncthreads() would also wait (vmcnt(0)) for this row's global stores to be acknowledged
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (w > 0) {
        int fv = kNeg, fa = -1;                 // prefix over columns 1 .. W0-1
        for (int v = 0; v < w; ++v) {
          take_later(fv, fa, xch[par][v][0], xch[par][v][1]);
          if (v < w - 1) { int Cn = (v + 1) * 256 * R; take_later(fv, fa, xch[par][v][2] + ge * Cn, Cn); }
        }
        int h0 = xch[par][w - 1][2];
        uint32_t p0 = (uint32_t)xch[par][w - 1][3];
        if (lane == 0) { d[0][0] = h0; pf[0][0] = p0; }
        int f2v = fv, f2a = fa;
        take_later(f2v, f2a, h0 + ge * W0, W0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          bool gt = cv[r] > f2v;
          int nv = max(f2v, cv[r]);
          int na = gt ? ca[r] : f2a;
          if (r == 0) { nv = (lane == 0) ? fv : nv; na = (lane == 0) ? fa : na; }
          cv[r] = nv; ca[r] = na;
        }
      }
      par ^= 1;
    }
    if (LOCAL) {
      int rm = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x) rm = max(rm, d[r][x]);
      if (rm > lmax) {                          // rare: resolve the first column of this lane reaching the new maximum
        lmax = rm;
        int cfirst = 0x7FFFFFFF;
#pragma unroll
        for (int r = R - 1; r >= 0; --r)
#pragma unroll
          for (int x = 3; x >= 0; --x) cfirst = (d[r][x] == rm) ? (cb + 256 * r + x) : cfirst;
        lpos = ((uint32_t)i << 16) | (uint32_t)cfirst;
      }
    }
    store_row(i);
  };

  // ---- row 0, row 1 -----------------------------------------------------------------------------
  store_row(0);   // untouched cells: score 0, pointer (-1,-1)  (dpmatrix.cpp:17-25)
  if (Q >= 3) {
    // row 1 (dpmatrix.h:409-418 / :579-590): match at (1,1), otherwise one gap from the origin; pointer (0,0)
    const int qrow = SIMPLANE ? 0 : (int)qc[1] * 128;
    auto row1 = [&](int c, int s) -> int {
      int cost = (c >= 2 && !prm.free_del) ? gi + ge * (c - 2) : 0;
      int h = s - cost;
      if (LOCAL) h = max(h, 0);
      return ((unsigned)(c - 1) < (unsigned)(T - 2)) ? h : 0;
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int c = cb + 256 * r + x;
        int s = SIMPLANE ? ((c < T) ? (int)S[(size_t)ld + c] : 0) : tab_at(qrow, code4[r][x]);
        d[r][x] = row1(c, s);
        pf[r][x] = ((unsigned)(c - 1) < (unsigned)(T - 2)) ? 0u : kNullPtr;
      }
    }
    int hB = 0; uint32_t pB = kNullPtr;
    if (NW > 1) {
      int s = SIMPLANE ? ((CB < T) ? (int)S[(size_t)ld + CB] : 0) : tab_at(qrow, codeB4);
      hB = row1(CB, s);
      pB = ((unsigned)(CB - 1) < (unsigned)(T - 2)) ? 0u : kNullPtr;
    }
    finish_row(1, hB, pB);
  }

  // ---- interior rows 2 .. Q-2 (dpmatrix.h:447-486 / :607-649) -----------------------------------------
  int qcode_next = (!SIMPLANE && Q >= 4) ? (int)qc[2] : 0;
  for (int i = 2; i <= Q - 2; ++i) {
    const int qrow = qcode_next * 128;
    if (!SIMPLANE && i + 1 <= Q - 2) qcode_next = (int)qc[i + 1];
    const int roff = gi + ge * (i - 2);         // insertion cost offset in B-space
    const int rowB = ge * (i - 1);
    const int im1 = i - 1;
    const uint32_t im1s = (uint32_t)im1 << 16;

    int bst[R][4]; uint32_t ptt[R][4];
    // cell phase: source column c -> (best, pointer) of target column c+1
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int pv = cv[r], pa = ca[r];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        const int gec = gecb + ge * (256 * r + x);
        const int m = d[r][x];
        int A = m + gec;
        if (r == 0 && x == 0) A = (cb == 0) ? kNeg : A;   // column 0 is never a source (dpmatrix.h:459 starts at t0+1)
        const int e = pv - gec - gime;          // E(c+1) = prefmax_{k<=c-1} A(k) - gi - ge (c-1)
        const int f = gmx[r][x] - roff;         // F = max_k D[k][c] + ge k - gi - ge (i-2)
        const bool de = e > m;
        int best = max(m, e);
        const bool df = f > best;
        best = max(best, f);
        const int pq = df ? gar[r][x] : im1;
        const int pt = (de && !df) ? pa : c;
        bst[r][x] = best;
        ptt[r][x] = ((uint32_t)pq << 16) | (uint32_t)pt;
        take_later(pv, pa, A, c);
      }
    }
    // boundary target (first column of the next wave), finished by this wave's lane 63
    int hB = 0; uint32_t pB = kNullPtr;
    if (NW > 1) {
      int s = SIMPLANE ? ((CB < T) ? (int)S[(size_t)i * ld + CB] : 0) : tab_at(qrow, codeB4);
      int h = bst[R - 1][3] + s;
      uint32_t p = ptt[R - 1][3];
      if (LOCAL) { bool pos = h > 0; h = pos ? h : 0; p = pos ? p : (im1s | (uint32_t)(CB - 1)); }
      bool in = CB <= T - 2;
      hB = in ? h : 0;
      pB = in ? p : kNullPtr;
    }
    // vertical state update with row i-1 (k = i-1 becomes a candidate for row i+1)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int B = d[r][x] + rowB;
        bool rec = B > gmx[r][x];
        gmx[r][x] = max(gmx[r][x], B);
        gar[r][x] = rec ? im1 : gar[r][x];
      }
    // shift one column right, add the target column's similarity, clip, mask
    int prev_b = 0; uint32_t prev_p = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int ub = wave_shr1(bst[r][3], 0);
      int up = wave_shr1((int)ptt[r][3], 0);
      if (r > 0) {
        ub = (lane == 0) ? prev_b : ub;
        up = (lane == 0) ? (int)prev_p : up;
      }
      prev_b = __builtin_amdgcn_readlane(bst[r][3], 63);
      prev_p = (uint32_t)__builtin_amdgcn_readlane((int)ptt[r][3], 63);
      const bool masked = (r == 0 && W0 == 0) || (W0 + 256 * (r + 1) > T - 1);   // wave-uniform
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int c = cb + 256 * r + x;
        int b = (x == 0) ? ub : bst[r][x - 1];
        uint32_t p = (x == 0) ? (uint32_t)up : ptt[r][x - 1];
        int s = SIMPLANE ? ((c < T) ? (int)S[(size_t)i * ld + c] : 0) : tab_at(qrow, code4[r][x]);
        int h = b + s;
        if (LOCAL) { bool pos = h > 0; h = pos ? h : 0; p = pos ? p : (im1s | (uint32_t)(c - 1)); }
        if (masked) {
          // column 1: one insertion from the origin (dpmatrix.h:421-426 / :593-599), pointer (0,0)
          int h1 = s - (prm.free_ins ? 0 : roff);
          if (LOCAL) h1 = max(h1, 0);
          bool is1 = c == 1;
          h = is1 ? h1 : h; p = is1 ? 0u : p;
          bool in = (unsigned)(c - 1) < (unsigned)(T - 2);
          h = in ? h : 0; p = in ? p : kNullPtr;
        }
        d[r][x] = h; pf[r][x] = p;
      }
    }
    finish_row(i, hB, pB);
  }

  // ---- last row: untouched except the corner, which dp_corner_kernel writes ------------------------
  if (Q >= 2) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < 4; ++x) { d[r][x] = 0; pf[r][x] = kNullPtr; }
    store_row(Q - 1);
  }

  // ---- find_max partial (optimal.h:108-124): value and first row-major position over interior cells ----
  if (LOCAL) {
    int m = lmax;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    uint32_t p = (lmax == m && m > 0) ? lpos : 0xFFFFFFFFu;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) p = min(p, (uint32_t)__shfl_xor((int)p, o));
    if (NW > 1) {
      if (lane == 0) { red[w][0] = m; red[w][1] = (int)p; }
      __syncthreads();
      int gm = 0;
      for (int v = 0; v < NW; ++v) gm = max(gm, red[v][0]);
      uint32_t gp = 0xFFFFFFFFu;
      for (int v = 0; v < NW; ++v) if (red[v][0] == gm) gp = min(gp, (uint32_t)red[v][1]);
      m = gm; p = gp;
    }
    if (threadIdx.x == 0) { res[blockIdx.x].part_max = (float)m; res[blockIdx.x].part_pos = p; }
  } else {
    if (threadIdx.x == 0) { res[blockIdx.x].part_max = 0.f; res[blockIdx.x].part_pos = 0xFFFFFFFFu; }
  }
}

// ---- host side --------------------------------------------------------------------------------------

// Proof that the reference's fp32 arithmetic is exact and the integer kernel cannot overflow:
// every similarity and gap value an integer, and every intermediate magnitude below 2^24.
bool fast_path_legal(const aln_batch* b, const float* table, int n, const aln_gap* gap, bool simplane_integral) {
  if (gap->model != ALN_GAP_AFFINE_CONST) return false;
  if (b->maxld > 8192) return false;            // the row-sweep kernels hold a row in registers: 4 waves x 8 groups x 256 columns at most
  float gi = gap->gap_init, ge = gap->gap_extn;
  if (!(gi == (float)(int)gi) || !(ge == (float)(int)ge)) return false;
  if (gi < 0 || ge < 0 || gi > 65536.f || ge > 4096.f) return false;
  double maxs = 0;
  if (table) {
    for (int k = 0; k < n * n; ++k) {
      float v = table[k];
      if (!(v == (float)(int)v)) return false;
      if (fabs((double)v) > maxs) maxs = fabs((double)v);
    }
  } else {
    if (!simplane_integral) return false;
    maxs = 4096;   // bound enforced by the caller's scan of the planes
  }
  double span = (double)b->maxQ + (double)b->maxT;
  double bound = (maxs + ge) * span + gi + maxs;
  return bound < 8388608.0;   // 2^23: leaves a factor two of headroom below fp32's 2^24 integer range
}

template <int NW, int R>
static int launch_variant(aln_batch* b, bool simplane, const FastParams& prm) {
  dim3 grid(b->n_pairs), block(64 * NW);
  hipStream_t st = b->ctx->stream;
  const bool local = b->islocal;
#define ALN_LAUNCH(LOC, SIMP)                                                                                   \
  hipLaunchKernelGGL((dp_affine_int_kernel<NW, R, LOC, SIMP>), grid, block, 0, st, b->d_pairs, b->d_qcodes,     \
                     b->d_tcodes, b->d_table32, b->d_H, b->d_P, b->d_S, b->d_res, prm)
  if (local) { if (simplane) ALN_LAUNCH(true, true); else ALN_LAUNCH(true, false); }
  else       { if (simplane) ALN_LAUNCH(false, true); else ALN_LAUNCH(false, false); }
#undef ALN_LAUNCH
  char nm[96];
  snprintf(nm, sizeof nm, "dp_affine_int_kernel<NW=%d,R=%d,%s,%s>", NW, R, local ? "local" : "global", simplane ? "simplane" : "submatrix");
  b->kernel_name = nm;
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

int launch_dp_affine_int(aln_batch* b, bool use_simplane) {
  FastParams prm;
  prm.gi = (int)b->gap.gap_init;
  prm.ge = (int)b->gap.gap_extn;
  prm.free_del = b->gapdev.free_del;
  prm.free_ins = b->gapdev.free_ins;
  const int ld = b->maxld;
  // variant choice: columns covered = 256 * R * NW >= ld.  Prefer several waves per pair (VALU issue needs >= 2 waves
  // per SIMD; one barrier per row is cheap) — overridable for tuning with ALN_DP_VARIANT="NW,R".
  int nw = b->ctx->hints.dp_nw, r = b->ctx->hints.dp_r;
  if (nw == 0) {
    if (ld <= 256) { nw = 1; r = 1; }
    else if (ld <= 512) { nw = 2; r = 1; }
    else if (ld <= 1024) { nw = 4; r = 1; }
    else if (ld <= 2048) { nw = 4; r = 2; }
    else if (ld <= 4096) { nw = 4; r = 4; }
    else if (ld <= 8192) { nw = 4; r = 8; }
    else return ALN_E_TOO_LONG;
  }
  if (256 * nw * r < ld) return ALN_E_TOO_LONG;
#define ALN_V(NW_, R_) if (nw == NW_ && r == R_) return launch_variant<NW_, R_>(b, use_simplane, prm)
  ALN_V(1, 1); ALN_V(1, 2); ALN_V(1, 4); ALN_V(1, 8);
  ALN_V(2, 1); ALN_V(2, 2); ALN_V(2, 4);
  ALN_V(4, 1); ALN_V(4, 2); ALN_V(4, 4); ALN_V(4, 8);
  ALN_V(8, 1);
#undef ALN_V
  return ALN_E_ARG;
}

}  // namespace aln
