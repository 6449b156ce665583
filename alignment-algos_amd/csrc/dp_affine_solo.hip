// dp_affine_solo.hip — the tagged-key row sweep with ONE wave per pair (gfx950), templates of up to 2048 columns.
//
// Same recurrence, keys, planes and results as dp_affine_tag.hip (reference dpmatrix.h:356-689 collapsed per SURVEY.md A.6).
// What differs is who waits for whom.  In dp_affine_tag a pair is two waves on two SIMDs that meet at a workgroup barrier in
// every row, each sharing its SIMD with a wave of another pair: PMC (profiles/r02_lone_stall_pmc.json) shows waves executing
// 55 % of their life, a quarter of it spent in s_waitcnt / s_barrier, and every experiment that disturbed the lock step of the
// two waves (skewed exchange, segment queue) lost.  Only LATER columns depend on EARLIER ones (deletion scans run left to right,
// the boundary cell of one column strip is the first column of the next), so nothing forces the two strips to run at the same
// time: here one wave owns the whole pair and visits its (up to four) 512-column strips one after the other in blocks of RB rows —
// strip 0 for rows i0 .. i0+RB-1 (leaving per row the prefix key, the boundary cell and its pointer in LDS), then strip 1 for the
// same rows, and so on.  No barrier, no cross-wave exchange; a wave never waits for another one, and stalls of one wave are
// covered by whatever other pairs' waves share its SIMD.  Every strip's row state (previous row, insertion maxima, scan carry,
// packed residue offsets: 19 VGPRs) stays in registers; the per-column key constants are recomputed at every switch.
//
// Occupancy: one wave per pair, so a launch needs >= 2048 pairs in flight to put two waves on every SIMD; callers with 1024-pair
// batches keep two or three launches in flight on as many streams (bench.py does).  Planes are written exactly as dp_affine_tag
// does (uint16 score + uint16 pointer word in local builds), one 1-KB store per plane, strip and row.
#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "dp_tag_common.h"

namespace aln {

// OCC: waves per SIMD the register allocation is made for (2: ~200 VGPRs, 3: 168)
template <bool LOCAL, bool H16, int KBT, int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void dp_affine_solo_kernel(
    const PairDesc* __restrict__ pairs, const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
    const int32_t* __restrict__ table32, float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
    PairResult* __restrict__ res, TagParams prm) {
  constexpr int TB = 11, X = 8;
  typedef TagBits<TB> tag;
  constexpr int TAGMAX = tag::TAGMAX, P_MATCH = tag::P_MATCH, P_DEL = tag::P_DEL, P_INS = tag::P_INS;
  constexpr int ZKEY = tag::ZKEY, ORIGIN_DEL = tag::ORIGIN_DEL, ORIGIN_INS = tag::ORIGIN_INS;
  constexpr int KB = (KBT == 16) ? 16 : TB + 2;
  constexpr int LOW = (1 << KB) - 1;
  constexpr int NEGK = (KBT == 16) ? -(1 << 29) : tag::NEGK;
  static_assert(KBT == 13 || (KBT == 16 && LOCAL && H16), "the 16-bit key layout needs non-negative 15-bit scores");
  constexpr int SW = 64 * X;            // columns of one strip (512): one lane-contiguous stretch of a row, one scan
  constexpr int NS = 4;                 // strips: up to 2048 columns
  constexpr int RB = 16;                // rows a strip runs before the next strip follows
  __shared__ int tab[32 * 32];          // substitution scores << KB
  __shared__ uint8_t qcs[1 << TB];      // the query's residue codes
  // per row of the block and strip: prefix key over every column up to the strip's last, boundary cell (= first column of the
  // next strip), its pointer word
  __shared__ __attribute__((aligned(16))) int ring[RB][NS][4];

  const PairDesc pd = pairs[blockIdx.x];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int lane = threadIdx.x;
  const int gi = prm.gi, ge = prm.ge;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint16_t* __restrict__ P = reinterpret_cast<uint16_t*>(Pbase) + pd.plane_off;
  uint16_t* const H16p = reinterpret_cast<uint16_t*>(Hbase) + pd.plane_off;
  const uint8_t* __restrict__ qc = qcodes + pd.q_off;
  const uint8_t* __restrict__ tc = tcodes + pd.t_off;

  for (int k = lane; k < 32 * 32; k += 64) tab[k] = table32[k] * (1 << KB);
  for (int k = lane; k < Q; k += 64) qcs[k] = qc[k];
  __syncthreads();

  // ---- per-strip row state, resident in registers for every strip -------------------------------------------------------
  int dkS[NS][X];         // D[i-1][c] << KB
  int gmxS[NS][X];        // running max over k of key(D[k][c] + ge*k, insertion, k)
  int cvkS[NS];           // lane-exclusive prefix key of the row in dk (A-space)
  uint32_t c4S[NS][2];    // byte offsets (code * 4 <= 124) of the columns' residues in a table row, four per register
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    cvkS[s] = NEGK;
    c4S[s][0] = c4S[s][1] = 0;
#pragma unroll
    for (int x = 0; x < X; ++x) {
      const int c = s * SW + X * lane + x;
      int code = kCodeTail;
      if (c < T) code = tc[c];
      c4S[s][x >> 2] |= (uint32_t)(code * 4) << (8 * (x & 3));
      dkS[s][x] = 0; gmxS[s][x] = NEGK;
    }
  }
  int lmax = 0; uint32_t lpos = 0;

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto tab_at = [&](int qrow, int c4) -> int {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(tab) + qrow + c4);
  };
  const int lane_row4 = (lane & 31) * 4;

  // rows i0 .. i1 (1 <= i0, i1 <= Q-2) of strip S
  const int T_k = T, ld_k = ld, Q_k = Q, gi_k = gi, ge_k = ge, fd_k = prm.free_del, fi_k = prm.free_ins;
  auto phase = [&](auto S_, int i0, int i1) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    asm volatile("" ::: "memory");            // the ring is written in one phase and read in the next: keep the accesses in program order
    // Opaque copies of the wave-uniform inputs: everything derived from them (masks, plane descriptors, shifted gap constants)
    // is then computed at the start of a phase — once per 16 rows — instead of being hoisted out of the row-block loop for all
    // four strips at once, which costs more scalar registers than the wave has.
    int T = T_k, ld = ld_k, Q = Q_k, gi = gi_k, ge = ge_k, free_del = fd_k, free_ins = fi_k;
    asm volatile("" : "+s"(T), "+s"(ld), "+s"(Q), "+s"(gi), "+s"(ge), "+s"(free_del), "+s"(free_ins));
    int (&dk)[X] = dkS[S];
    int (&gmx)[X] = gmxS[S];
    int& cvk = cvkS[S];
    const int W0 = S * SW;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));          // opaque: the column constants below must be REcomputed here, not hoisted out of the
    const int cb = W0 + X * lane_o;           // row-block loop into 24 more resident registers per strip
    const int vo16 = cb * 2;
    const int CB = W0 + SW;                    // boundary target: first column of the next strip
    int codeB4 = kCodeTail * 4;
    if (S + 1 < NS && CB < T) codeB4 = tc[CB] * 4;
    // constants of this strip's columns, recomputed at every switch (a few scalar-operand ops each, against 24 resident registers)
    int GK[X], EK[X], code4[X];
#pragma unroll
    for (int x = 0; x < X; ++x) {
      const int c = cb + x;
      GK[x] = ((ge * c) * (1 << KB)) | P_DEL | (TAGMAX - (c & TAGMAX));
      EK[x] = (ge * c + gi - ge) * (1 << KB);
      code4[x] = (int)((c4S[S][x >> 2] >> (8 * (x & 3))) & 0xFFu);
    }
    int ak[X];
#pragma unroll
    for (int x = 0; x < X; ++x) ak[x] = dk[x] + GK[x];
    uint32_t pf[X];

    auto store_row = [&](int i) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(P + (size_t)i * ld, 0, ld * 2, 0x00020000);
      uint32_t pw[X / 2], hw[X / 2];
#pragma unroll
      for (int x = 0; x < X; x += 2) pw[x / 2] = __builtin_amdgcn_perm(pf[x + 1], pf[x], 0x05040100u);
      if (H16) {
        const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(H16p + (size_t)i * ld, 0, ld * 2, 0x00020000);
#pragma unroll
        for (int x = 0; x < X; x += 2)
          hw[x / 2] = (KBT == 16) ? __builtin_amdgcn_perm((uint32_t)dk[x + 1], (uint32_t)dk[x], 0x07060302u)
                                  : (((uint32_t)dk[x] >> KB) | (((uint32_t)dk[x + 1] >> KB) << 16));
        const u32x4 v = {hw[0], hw[1], hw[2], hw[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsH, vo16, 0, 0);
      } else if (cb < ld) {
        const float sc = 1.0f / (float)(1 << KB);        // exact: values are multiples of 2^KB
        const size_t ro = (size_t)i * ld + cb;
#pragma unroll
        for (int x = 0; x < X; x += 4)
          *reinterpret_cast<float4*>(H + ro + x) =
              make_float4((float)dk[x] * sc, (float)dk[x + 1] * sc, (float)dk[x + 2] * sc, (float)dk[x + 3] * sc);
      }
      const u32x4 v = {pw[0], pw[1], pw[2], pw[3]};
      __builtin_amdgcn_raw_buffer_store_b128(v, rsP, vo16, 0, 0);
    };

    // finish the row held in dk[]/pf[]: scan preparation for the next row, hand-over between the strips, find_max, store
    auto finish_row = [&](int i, int dB, uint32_t pB) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < X; ++x) ak[x] = dk[x] + GK[x];
      const int a0 = (lane == 0) ? NEGK : ak[0];            // column 0 is never a source; a later strip's first column is folded below
      int t0 = max(a0, ak[1]), t1 = max(ak[2], ak[3]);
      t0 = max(max(t0, ak[4]), ak[5]); t1 = max(max(t1, ak[6]), ak[7]);
      const int ik = wave_incl_max_key(max(t0, t1));
      cvk = tdpp<0x138>(NEGK, ik);                          // wave_shr:1 -> exclusive
      int tot = __builtin_amdgcn_readlane(ik, 63);          // prefix key over this strip's own columns (without its first)
      const int slot = (i - i0) & (RB - 1);
      if (S > 0) {
        const int4 t = *reinterpret_cast<const int4*>(ring[slot][S - 1]);   // this wave wrote it a phase ago: program order suffices
        const int fk = t.x, d0 = t.y; const uint32_t p0 = (uint32_t)t.z;
        if (lane == 0) { dk[0] = d0; pf[0] = p0; ak[0] = d0 + GK[0]; }
        const int f2 = max(fk, d0 + (((ge * W0) * (1 << KB)) | P_DEL | (TAGMAX - W0)));
        const int nv = max(f2, cvk);
        cvk = (lane == 0) ? fk : nv;
        tot = max(tot, f2);
      }
      if (S + 1 < NS) {
        if (lane == 63) { int* w = ring[slot][S]; w[0] = tot; w[1] = dB; w[2] = (int)pB; }
      }
      if (LOCAL) {
        int rm = 0;
#pragma unroll
        for (int x = 0; x < X; x += 2) rm = max(max(rm, dk[x]), dk[x + 1]);
        if (rm > lmax || (rm == lmax && rm > 0)) {          // strips are not visited in row-major order: keep the first position
          int cfirst = 0x7FFFFFFF;
#pragma unroll
          for (int x = X - 1; x >= 0; --x) cfirst = (dk[x] == rm) ? (cb + x) : cfirst;
          const uint32_t pos = ((uint32_t)i << 16) | (uint32_t)cfirst;
          if (rm > lmax || pos < lpos) { lmax = rm; lpos = pos; }
        }
      }
      store_row(i);
    };

    // similarity pipeline (see dp_affine_tag.hip): residue code of row i+2, table row of row i+1, per-cell values of row i
    int code_n1 = (int)qcs[min(i0 + 1, Q - 1)];
    int rowv_next = tab_at((int)qcs[min(i0, Q - 1)] * 128, lane_row4);
    for (int i = i0; i <= i1; ++i) {
      const int rowv = rowv_next;
      rowv_next = tab_at(code_n1 * 128, lane_row4);
      code_n1 = (int)qcs[min(i + 2, Q - 1)];
      int sv[X];
#pragma unroll
      for (int x = 0; x < X; ++x) sv[x] = __builtin_amdgcn_ds_bpermute(code4[x], rowv);
      const int svB = (S + 1 < NS) ? __builtin_amdgcn_ds_bpermute(codeB4, rowv) : 0;
      int dB = 0; uint32_t pB = kNullPtr;
      if (i == 1) {
        // row 1 (dpmatrix.h:409-418 / :579-590): match at (1,1), otherwise one deletion from the origin -> pointer (0,0)
        auto row1 = [&](int c, int sK, int& dkv, uint32_t& pv) __attribute__((always_inline)) {
          const int cost = (c >= 2 && !free_del) ? ((gi + ge * (c - 2)) * (1 << KB)) : 0;
          int v = sK - cost;
          if (LOCAL) v = max(v, 0);
          const bool in = (unsigned)(c - 1) < (unsigned)(T - 2);
          dkv = in ? v : 0;
          pv = in ? (uint32_t)(c == 1 ? P_MATCH : ORIGIN_DEL) : kNullPtr;
        };
#pragma unroll
        for (int x = 0; x < X; ++x) row1(cb + x, sv[x], dk[x], pf[x]);
        if (S + 1 < NS) row1(CB, svB, dB, pB);
        finish_row(1, dB, pB);
        continue;
      }
      // ---- interior rows (dpmatrix.h:447-486 / :607-649) ------------------------------------------------------------------
      const int FK = (gi + ge * (i - 2)) * (1 << KB);
      const int RK = ((ge * (i - 1)) * (1 << KB)) | P_INS | (TAGMAX - (i - 1));
      const int colK = free_ins ? 0 : FK;
      int bk[X];
      {
        int pv = cvk;
#pragma unroll
        for (int x = 0; x < X; ++x) {
          const int m = dk[x];
          int A = ak[x];
          if (x == 0) A = (cb == 0) ? NEGK : A;              // column 0 is never a source
          const int e = pv - EK[x];
          const int f = gmx[x] - FK;
          bk[x] = max(max(m | P_MATCH, e), f);
          pv = max(pv, A);
        }
      }
      if (S + 1 < NS) {                                        // boundary target: first column of the next strip
        int kh = bk[X - 1] + svB;
        if (LOCAL) kh = max(kh, ZKEY);
        const bool in = CB <= T - 2;
        dB = in ? (kh & ~LOW) : 0;
        pB = in ? (uint32_t)(kh & LOW) : kNullPtr;
      }
#pragma unroll
      for (int x = 0; x < X; ++x) {
        gmx[x] = max(gmx[x], dk[x] + RK);
        asm volatile("" : "+v"(gmx[x]));
      }
      {
        // shift one column right (lane 0's own first column comes through the ring, strip 0's column 0 is a don't-care)
        const int uk = __builtin_amdgcn_update_dpp(0, bk[X - 1], 0x138, 0xF, 0xF, true);
        const bool masked = (S == 0) || (W0 + SW > T - 1);     // wave-uniform
        int sK1 = 0;
#pragma unroll
        for (int x = 0; x < X; ++x) {
          const int sK = sv[x];
          if (x == 1) sK1 = sK;
          int kh = ((x == 0) ? uk : bk[x - 1]) + sK;
          if (LOCAL) kh = max(kh, ZKEY);
          dk[x] = kh & ~LOW;
          pf[x] = (KBT == 16) ? (uint32_t)kh : (uint32_t)(kh & LOW);
        }
        if (masked) {
          asm volatile("" ::: "memory");
          if (S == 0) {                                        // column 1 (dpmatrix.h:421-426 / :593-599), pointer (0,0)
            int v1 = sK1 - colK;
            if (LOCAL) v1 = max(v1, 0);
            const bool is1 = cb == 0;
            dk[1] = is1 ? v1 : dk[1]; pf[1] = is1 ? (uint32_t)ORIGIN_INS : pf[1];
          }
#pragma unroll
          for (int x = 0; x < X; ++x) {
            const int c = cb + x;
            const int inm = ((unsigned)(c - 1) < (unsigned)(T - 2)) ? -1 : 0;
            dk[x] &= inm;                                      // columns 0 and >= T-1: score 0, null pointer
            pf[x] |= ~(uint32_t)inm;
          }
        }
      }
      finish_row(i, dB, pB);
    }
    // untouched rows of this strip (row 0 and the last row: score 0, null pointer; the corner comes from dp_corner_kernel)
    if (i0 == 1 || i1 >= Q - 2) {
      int keep[X];
#pragma unroll
      for (int x = 0; x < X; ++x) { keep[x] = dk[x]; dk[x] = 0; pf[x] = kNullPtr; }
      if (i0 == 1) store_row(0);
      if (i1 >= Q - 2) store_row(Q - 1);
#pragma unroll
      for (int x = 0; x < X; ++x) dk[x] = keep[x];
    }
  };

  const int nstrips = (ld + SW - 1) / SW;                      // strips that hold columns of this pair (the others would store nothing)
  auto block = [&](int i0, int i1) __attribute__((always_inline)) {
    phase(std::integral_constant<int, 0>(), i0, i1);
    if (nstrips > 1) phase(std::integral_constant<int, 1>(), i0, i1);
    if (nstrips > 2) phase(std::integral_constant<int, 2>(), i0, i1);
    if (nstrips > 3) phase(std::integral_constant<int, 3>(), i0, i1);
  };
  if (Q >= 3) {
    for (int i0 = 1; i0 <= Q - 2; i0 += RB) block(i0, min(i0 + RB - 1, Q - 2));
  } else {
    block(1, 0);                                               // no interior row: only the untouched rows 0 and Q-1 exist
  }

  // ---- find_max partial (optimal.h:108-124): value and first row-major position over interior cells -----------------------
  if (LOCAL) {
    int m = lmax;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    uint32_t p = (lmax == m && m > 0) ? lpos : 0xFFFFFFFFu;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) p = min(p, (uint32_t)__shfl_xor((int)p, o));
    if (lane == 0) { res[blockIdx.x].part_max = (float)(m >> KB); res[blockIdx.x].part_pos = p; }
  } else {
    if (lane == 0) { res[blockIdx.x].part_max = 0.f; res[blockIdx.x].part_pos = 0xFFFFFFFFu; }
  }
}

// host side -------------------------------------------------------------------------------------------------------------
bool tag_key16_legal(const aln_batch* b);   // dp_affine_tag.hip

bool dp_affine_solo_legal(const aln_batch* b) {
  return b->ptr_mode == 1 && b->maxld <= 2048 && b->maxQ <= 2048;
}

int launch_dp_affine_solo(aln_batch* b) {
  TagParams prm = {};
  prm.gi = (int)b->gap.gap_init;
  prm.ge = (int)b->gap.gap_extn;
  prm.free_del = b->gapdev.free_del;
  prm.free_ins = b->gapdev.free_ins;
  prm.n_pairs = b->n_pairs;
  dim3 grid(b->n_pairs), block(64);
  hipStream_t st = b->ctx->stream;
  const bool k16 = b->islocal && b->h_mode == 1 && tag_key16_legal(b) && b->ctx->hints.key16;
  const bool occ3 = b->ctx->hints.tag_solo != 2;           // hint value 2: allocate for two waves per SIMD; otherwise three
#define ALN_SOLO(LOC_, H16_, KB_)                                                                                              \
  do { if (occ3) hipLaunchKernelGGL((dp_affine_solo_kernel<LOC_, H16_, KB_, 3>), grid, block, 0, st, b->d_pairs, b->d_qcodes,  \
                                    b->d_tcodes, b->d_table32, b->d_H, b->d_P, b->d_res, prm);                                  \
       else hipLaunchKernelGGL((dp_affine_solo_kernel<LOC_, H16_, KB_, 2>), grid, block, 0, st, b->d_pairs, b->d_qcodes,       \
                               b->d_tcodes, b->d_table32, b->d_H, b->d_P, b->d_res, prm); } while (0)
  if (k16) ALN_SOLO(true, true, 16);
  else if (b->islocal && b->h_mode == 1) ALN_SOLO(true, true, 13);
  else if (b->islocal) ALN_SOLO(true, false, 13);
  else ALN_SOLO(false, false, 13);
#undef ALN_SOLO
  char nm[96];
  snprintf(nm, sizeof nm, "dp_affine_solo_kernel<%s%s%s,occ%d>", b->islocal ? "local" : "global", b->h_mode == 1 ? ",h16" : "", k16 ? ",key16" : "",
           occ3 ? 3 : 2);
  b->kernel_name = nm;
  b->tag_segmented = false;
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

}  // namespace aln
