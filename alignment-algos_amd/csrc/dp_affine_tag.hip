// dp_affine_tag.hip — tagged-key O(n^2) row-sweep DP for constant affine gaps, integer scores, Q,T <= 4096 (gfx950).
//
// Same recurrence, mapping and tie-breaking as dp_affine_int.hip (reference dpmatrix.h:356-689 collapsed per
// SURVEY.md A.6), rebuilt around what the gfx950 VALU actually issues at full rate.  Measured on MI355X
// (tools/valu_rate2.hip, 4 waves/SIMD): v_add/v_sub/v_and/v_or/v_ashr/v_mov/v_add_f32/v_mul_f32 issue every
// ~2.2 cycles; v_max, v_cmp, v_cndmask, every 3-operand VOP3, v_cvt and DPP ops every ~3.8.  The first kernel
// spent two thirds of its issue slots on v_cmp + v_cndmask pairs that only carried arg-max bookkeeping, and was
// VALU-issue bound (profiles/r01_a_*).  Here the bookkeeping rides in the low bits of the values:
//
//     key = value << 13 | prio << 11 | tag          (value: 19 signed bits, |value| < 2^16 proven on the host)
//       match      prio 3, tag ignored                      -> predecessor (i-1, j-1)
//       deletion   prio 2, tag = 2047 - k  (row i-1, col k) -> earliest k wins a tie, as "s > opt_s" demands
//       insertion  prio 1, tag = 2047 - k  (row k, col j-1)
//
// so "first strictly greater candidate in the order match, deletions k^, insertions k^" (dpmatrix.h:453-480) is
// ONE v_max3_i32 over three keys, a prefix maximum with its first arg-max is ONE v_max_i32 (or one fused
// v_max_i32_dpp step across lanes), the local-mode clip "s = max(0,s) keeps the match pointer" is one v_max against
// the key (0, match), and the traceback pointer of a cell is simply the key's low 13 bits (plane mode 1,
// decoded by aln_device.h::decode_ptr).  Gap constants and the substitution table are pre-shifted by 13, so all
// additions are plain full-rate adds that leave the tags alone.
//
// Per cell: ~18 VALU instructions (was ~63).  The pointer word needs TB+2 <= 14 bits, so the pointer plane is written as uint16
// (0xFFFF = untouched), and local builds write uint16 scores too: 4 bytes per cell reach HBM (6 with fp32 scores) instead of the
// 8 of an fp32 + 32-bit layout.  What bounds the kernel, and what was tried on it: DESIGN.md 4.1.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "aln_internal.h"
#include "dp_tag_common.h"

namespace aln {

__device__ __forceinline__ void setprio_dyn(int p) {     // s_setprio takes an immediate
  switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// SEGQ: workgroups take (pair, row segment) items from the queue instead of building pair blockIdx.x from first to last row.
template <int NW, int R, bool LOCAL, bool H16, int KBT, int X, bool SEGQ, int TB, int OCC>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(R * X == 16 ? OCC : 1, R * X == 16 ? OCC : 8))) void dp_affine_tag_kernel(
    const PairDesc* __restrict__ pairs, const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
    const int32_t* __restrict__ table32, float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
    PairResult* __restrict__ res, TagParams prm) {
  typedef TagBits<TB> tag;
  constexpr int TAGMAX = tag::TAGMAX, P_MATCH = tag::P_MATCH, P_DEL = tag::P_DEL, P_INS = tag::P_INS;
  constexpr int ZKEY = tag::ZKEY, ORIGIN_DEL = tag::ORIGIN_DEL, ORIGIN_INS = tag::ORIGIN_INS;
  constexpr int KB = (KBT == 16) ? 16 : TB + 2;                // KBT: 13 = "value right above the tag bits", 16 = score in the high half
  constexpr int LOW = (1 << KB) - 1;
  constexpr int NEGK = (KBT == 16) ? -(1 << 29) : tag::NEGK;   // value -8192 at KB = 16
  static_assert(KBT == 13 || (KBT == 16 && LOCAL && H16), "the 16-bit key layout needs non-negative 15-bit scores");
  static_assert(TB == 11 || TB == 12, "11 or 12 tag bits");
  static_assert(X == 4 || X == 8, "a lane owns 4 or 8 consecutive columns of each group");
  constexpr int GW = 64 * X;            // columns of one group (one lane-contiguous stretch of a row)
  __shared__ int tab[32 * 32];          // substitution scores << KB
  __shared__ uint8_t qcs[1 << TB];      // the query's residue codes (Q <= 2^TB): one LDS byte per row instead of a global load
  constexpr int RING = 16;              // exchange slots: one per row, reused every 16 rows
  __shared__ __attribute__((aligned(16))) int xch[RING][NW][4];
  __shared__ int red[NW][2];

  // ---- which pair, which rows ------------------------------------------------------------------------------------
  __shared__ int s_item;
  int pair_id = blockIdx.x, seg = 0;
  if constexpr (SEGQ) {
    if (threadIdx.x == 0) {
      const int t = __hip_atomic_fetch_add(&prm.queue[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;   // counters start at -1
      int item;
      if (t < prm.n_pairs) item = t * 8;                                   // first segment of pair t: nothing to wait for
      else {
        const int* slot = &prm.queue[16 + (t - prm.n_pairs)];
        long spins = 0;
        while ((item = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
          __builtin_amdgcn_s_sleep(32);
          if (++spins > (1L << 24)) {                                      // ~ seconds: something is broken; give up loudly, never hang
            __hip_atomic_store(&prm.queue[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item = -2;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      s_item = item;
    }
    __syncthreads();
    const int item = s_item;
    if (item < 0) return;
    pair_id = item >> 3; seg = item & 7;
  }
  const PairDesc pd = pairs[pair_id];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int n_seg = SEGQ ? seg_count(Q, prm.ksegs) : 1;
  const int i_begin = seg_bound(Q, seg, n_seg), i_end = seg_bound(Q, seg + 1, n_seg);   // interior rows [i_begin, i_end)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the exchange loop and the boundary tests stay scalar
  const int W0 = w * GW * R;
  const int cb = W0 + X * lane;
  const int gi = prm.gi, ge = prm.ge;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint16_t* __restrict__ P = reinterpret_cast<uint16_t*>(Pbase) + pd.plane_off;   // 16-bit pointer words (mode 1)
  const uint8_t* __restrict__ qc = qcodes + pd.q_off;
  const uint8_t* __restrict__ tc = tcodes + pd.t_off;

  for (int k = threadIdx.x; k < 32 * 32; k += 64 * NW) tab[k] = table32[k] * (1 << KB);
  for (int k = threadIdx.x; k < Q; k += 64 * NW) qcs[k] = qc[k];
  __syncthreads();

  // ---- static per-column constants -----------------------------------------------------------------
  int inm[R][X];        // -1 for interior columns 1 .. T-2, else 0
  int code4[R][X];      // byte offset of the column's residue in a table row
  int GK[R][X];         // ((ge*c) << 13 | P_DEL | (2047 - c)) - P_MATCH: dk + GK = key of A(c) = D + ge*c as a deletion source
  int EK[R][X];         // (ge*c + gi - ge) << 13: E(c+1) = prefmax - EK
  bool inrange[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    inrange[r] = (cb + GW * r) < ld;
#pragma unroll
    for (int x = 0; x < X; ++x) {
      const int c = cb + GW * r + x;
      int code = kCodeTail;
      if (c < T) code = tc[c];
      code4[r][x] = code * 4;
      inm[r][x] = ((unsigned)(c - 1) < (unsigned)(T - 2)) ? -1 : 0;
      GK[r][x] = (((ge * c) * (1 << KB)) | P_DEL | (TAGMAX - (c & TAGMAX))) - P_MATCH;   // (dk carries the match bits: see below)
      EK[r][x] = (ge * c + gi - ge) * (1 << KB);
    }
  }
  const int CB = W0 + GW * R;     // first column of the next wave = this wave's boundary target
  int codeB4 = kCodeTail * 4;
  if (NW > 1 && CB < T) codeB4 = tc[CB] * 4;

  int dk[R][X];         // D[i-1][c] << 13 | P_MATCH: the row's scores, already dressed as match candidates of the next row (the
                        // split of a finished key is one v_and_or_b32 either way; the next row's `m | P_MATCH` per cell is gone)
  int gmx[R][X];        // running max over k of key(D[k][c] + ge*k, insertion, 2047-k)
  int cvk[R];           // lane-exclusive prefix key of the row in dk (A-space), per group
  int ak[R][X];         // A-space keys of the row in dk: dk + GK (column 0 / wave firsts handled where they are used)
  uint32_t pf[R][X];    // pointer words of the row being finished
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cvk[r] = NEGK;
#pragma unroll
    for (int x = 0; x < X; ++x) { dk[r][x] = P_MATCH; gmx[r][x] = NEGK; pf[r][x] = kNullPtr; }
  }
  int lmax = P_MATCH; uint32_t lpos = 0;
  // Skewed exchange.  Only later waves depend on earlier ones (deletion scans run left to right, the boundary cell of wave v is
  // the first column of wave v+1; nothing flows back), so wave w may run behind wave v < w by any number of rows.  With
  // lag = L > 0 wave w handles row (it - L*w) in iteration `it`: what wave v wrote for that row (iteration row + L*v) is at least L
  // iterations old, and one barrier every L iterations separates every write from its reads and every read from the slot's reuse
  // (slot = row mod 16; L * (NW-1) + L <= 16).  The LDS write -> wait -> barrier -> read -> wait chain that sat inside every row
  // is gone: the reads are issued at the top of an iteration from a slot that has long been written.  lag = 0 keeps the
  // synchronous form (write, barrier, read in every row); row 1 always uses it.
  const int lag = (NW > 1) ? prm.lag : 0;

  // 16-bit planes are stored through buffer descriptors that cover ONE ROW (base = the row, num_records = 2*ld bytes),
  // rebuilt per row with scalar instructions: lanes of a partial last group (columns >= ld) are dropped by the memory
  // pipeline's range check -- no exec-mask branch around the stores.  (The scalar-offset operand cannot carry the row:
  // gfx950 includes it in the range check, measured with tools/scratch/buf_test.hip.)
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  uint16_t* const H16p = reinterpret_cast<uint16_t*>(Hbase) + pd.plane_off;
  const int vo16 = cb * 2;              // this lane's byte offset inside a 16-bit row
  auto store_words = [&](const uint32_t (&wd)[X / 2], __amdgpu_buffer_rsrc_t rs, int off) {
    if constexpr (X == 4) { const u32x2 v = {wd[0], wd[1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0); }
    else { const u32x4 v = {wd[0], wd[1], wd[2], wd[3]}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0); }
  };
  auto store_row = [&](int i) {
    const size_t ro = (size_t)i * ld + cb;
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(P + (size_t)i * ld, 0, ld * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(H16p + (size_t)i * ld, 0, ld * 2, 0x00020000);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // low halves of two pointer words (at KB = 16: of two whole keys): one v_perm_b32 per pair of cells
      uint32_t pw[X / 2], hw[X / 2];
#pragma unroll
      for (int x = 0; x < X; x += 2) pw[x / 2] = __builtin_amdgcn_perm(pf[r][x + 1], pf[r][x], 0x05040100u);
      if (H16) {                              // local: 0 <= score < 2^16 -> uint16 plane (2 B/cell)
#pragma unroll
        for (int x = 0; x < X; x += 2)
          hw[x / 2] = (KBT == 16) ? __builtin_amdgcn_perm((uint32_t)dk[r][x + 1], (uint32_t)dk[r][x], 0x07060302u)   // the scores ARE the high halves
                                  : (((uint32_t)dk[r][x] >> KB) | (((uint32_t)dk[r][x + 1] >> KB) << 16));
        store_words(hw, rsH, vo16 + 2 * GW * r);
      } else if (inrange[r]) {
        const float sc = 1.0f / (float)(1 << KB);     // exact: values are multiples of 2^KB
#pragma unroll
        for (int x = 0; x < X; x += 4)
          *reinterpret_cast<float4*>(H + ro + GW * r + x) =
              make_float4((float)(dk[r][x] - P_MATCH) * sc, (float)(dk[r][x + 1] - P_MATCH) * sc, (float)(dk[r][x + 2] - P_MATCH) * sc,
                          (float)(dk[r][x + 3] - P_MATCH) * sc);
      }
      store_words(pw, rsP, vo16 + 2 * GW * r);
    }
  };
  auto tab_at = [&](int qrow, int c4) -> int {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(tab) + qrow + c4);
  };

  // Finish the row held in dk[]/pf[] (complete except, for w > 0, the wave's first column which the previous wave
  // computed and passes as (dB,pB)): prefix-scan preparation for the next row, exchange, local-max tracking, store.
  auto finish_row = [&](int i, int dB, uint32_t pB, bool sync, const int4 (&xin)[NW > 1 ? NW - 1 : 1]) {
    int sk = NEGK;     // scalar carry: prefix key over this wave's earlier groups
    int ik[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int x = 0; x < X; ++x) ak[r][x] = dk[r][x] + GK[r][x];
      int a0 = ak[r][0];
      if (r == 0) a0 = (lane == 0) ? NEGK : a0;             // column 0 is never a source; wave firsts are folded below
      int t0 = max(a0, ak[r][1]), t1 = max(ak[r][2], ak[r][3]);
      if constexpr (X == 8) { t0 = max(max(t0, ak[r][4]), ak[r][5]); t1 = max(max(t1, ak[r][6]), ak[r][7]); }   // v_max3
      ik[r] = max(t0, t1);
    }
    wave_incl_max_keys<R>(ik);   // the R scans in lock-step: each DPP stage's hazard slots hold the other groups
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ek = tdpp<0x138>(NEGK, ik[r]);              // wave_shr:1 -> exclusive
      cvk[r] = max(sk, ek);
      sk = max(sk, __builtin_amdgcn_readlane(ik[r], 63));
    }
    if (NW > 1) {
      const int slot = i & (RING - 1);
      if (lane == 63) { xch[slot][w][0] = sk; xch[slot][w][1] = dB; xch[slot][w][2] = (int)pB; }
      // LDS-only barrier: a __syncthreads() would also wait (vmcnt(0)) for this row's global stores to be acknowledged
      if (sync) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (w > 0) {
        int fk = NEGK;                                      // prefix over columns 1 .. W0-1
        int d0 = P_MATCH; uint32_t p0 = kNullPtr;
#pragma unroll
        for (int v = 0; v < NW - 1; ++v) {
          if (v < w) {                                      // wave-uniform; one ds_read_b128 per earlier wave
            const int4 t = sync ? *reinterpret_cast<const int4*>(&xch[slot][v][0]) : xin[v];
            fk = max(fk, t.x);
            if (v < w - 1) {
              const int Cn = (v + 1) * GW * R;
              fk = max(fk, t.y + ((((ge * Cn) * (1 << KB)) | P_DEL | (TAGMAX - Cn)) - P_MATCH));
            } else {
              d0 = t.y; p0 = (uint32_t)t.z;
            }
          }
        }
        if (lane == 0) { dk[0][0] = d0; pf[0][0] = p0; ak[0][0] = d0 + GK[0][0]; }
        const int f2 = max(fk, d0 + ((((ge * W0) * (1 << KB)) | P_DEL | (TAGMAX - W0)) - P_MATCH));
#pragma unroll
        for (int r = 0; r < R; ++r) {
          int nv = max(f2, cvk[r]);
          if (r == 0) nv = (lane == 0) ? fk : nv;
          cvk[r] = nv;
        }
      }
    }
    if (LOCAL) {
      int rm = P_MATCH;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int x = 0; x < X; x += 2) rm = max(max(rm, dk[r][x]), dk[r][x + 1]);   // v_max3 per two cells
      if (rm > lmax) {                                      // rare: resolve the first column of this lane at the new maximum
        lmax = rm;
        int cfirst = 0x7FFFFFFF;
#pragma unroll
        for (int r = R - 1; r >= 0; --r)
#pragma unroll
          for (int x = X - 1; x >= 0; --x) cfirst = (dk[r][x] == rm) ? (cb + GW * r + x) : cfirst;
        lpos = ((uint32_t)i << 16) | (uint32_t)cfirst;
      }
    }
    store_row(i);
  };

  // ---- hand-off slots of the segment queue: 9 x 16 bytes per thread (dk[16], gmx[16], cvk[R], lmax, lpos), chunk-major ------
  constexpr int kStateChunks = (2 * R * X + R + 2 + 3) / 4;
  constexpr int kStateBytes = kStateChunks * 16 * 64 * NW;
  // (the slots' address is read from the queue header where it is needed, the item is re-read from LDS at the end: nothing of the
  // queue stays in registers across the row loop except the queue pointer)
  auto state_rsrc = [&](int pr, int sgm) {
    char* base = *reinterpret_cast<char* volatile*>(&prm.queue[4]);
    return __builtin_amdgcn_make_buffer_rsrc(base + ((size_t)pr * kSegs + sgm) * kStateBytes, 0, kStateBytes, 0x00020000);
  };
  auto state_words = [&](uint32_t (&wv)[kStateChunks * 4], bool save) {
    int n = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) { if (save) wv[n] = (uint32_t)dk[r][x]; else dk[r][x] = (int)wv[n]; ++n; }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) { if (save) wv[n] = (uint32_t)gmx[r][x]; else gmx[r][x] = (int)wv[n]; ++n; }
#pragma unroll
    for (int r = 0; r < R; ++r) { if (save) wv[n] = (uint32_t)cvk[r]; else cvk[r] = (int)wv[n]; ++n; }
    if (save) { wv[n] = (uint32_t)lmax; wv[n + 1] = lpos; } else { lmax = (int)wv[n]; lpos = wv[n + 1]; }
  };

  // ---- row 0, row 1 (first segment) — or the state the previous segment left ---------------------------------------
  if (SEGQ && seg > 0) {
    const __amdgpu_buffer_rsrc_t rs = state_rsrc(pair_id, seg - 1);
    uint32_t wv[kStateChunks * 4] = {};
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (c * 64 * NW + (int)threadIdx.x) * 16, 0, 16);   // aux 16 = sc1
      wv[4 * c] = v.x; wv[4 * c + 1] = v.y; wv[4 * c + 2] = v.z; wv[4 * c + 3] = v.w;
    }
    state_words(wv, false);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) ak[r][x] = dk[r][x] + GK[r][x];
  }
  if (seg == 0) store_row(0);   // untouched cells: score 0, null pointer (dpmatrix.cpp:17-25)
  if (seg == 0 && Q >= 3) {
    // row 1 (dpmatrix.h:409-418 / :579-590): match at (1,1), otherwise one deletion from the origin -> pointer (0,0)
    const int qrow = (int)qcs[1] * 128;
    auto row1 = [&](int c, int sK, int& dkv, uint32_t& pv) {
      const int cost = (c >= 2 && !prm.free_del) ? ((gi + ge * (c - 2)) * (1 << KB)) : 0;
      int v = sK - cost;
      if (LOCAL) v = max(v, 0);
      const bool in = (unsigned)(c - 1) < (unsigned)(T - 2);
      dkv = (in ? v : 0) | P_MATCH;
      pv = in ? (uint32_t)(c == 1 ? P_MATCH : ORIGIN_DEL) : kNullPtr;
    };
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) row1(cb + GW * r + x, tab_at(qrow, code4[r][x]), dk[r][x], pf[r][x]);
    int dB = P_MATCH; uint32_t pB = kNullPtr;
    if (NW > 1) row1(CB, tab_at(qrow, codeB4), dB, pB);
    const int4 none[NW > 1 ? NW - 1 : 1] = {};
    finish_row(1, dB, pB, true, none);
  }

  // ---- interior rows 2 .. Q-2 (dpmatrix.h:447-486 / :607-649) ---------------------------------------------
  // Similarity values travel through a three-stage pipeline so that no LDS round trip sits on a row's critical path:
  // the residue code of row i+2, the table row of row i+1 (one entry per lane) and the per-cell values of row i
  // (ds_bpermute with fixed per-lane addresses = code * 4, so no address arithmetic per cell) are all fetched at the top
  // of iteration i and consumed one stage later.
  const int lane_row4 = (lane & 31) * 4;
  const int hwslot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_ID.WAVE_ID: this wave's slot on its SIMD
  int code_n1 = (int)qcs[min(i_begin + 1, Q - 1)];                             // residue of row i+1 (clamped: only rows <= Q-2 are consumed)
  int rowv_next = tab_at((int)qcs[min(i_begin, Q - 1)] * 128, lane_row4);       // table row of row i
  const int it_end = i_end - 1 + lag * (NW - 1);
  for (int it = i_begin; it <= it_end; ++it) {
    const int i = it - lag * w;                                                  // the row this wave handles now
    const bool sync = lag == 0;
    if (i >= i_begin && i < i_end) {
    int4 xin[NW > 1 ? NW - 1 : 1];
    if (NW > 1 && !sync && w > 0) {
#pragma unroll
      for (int v = 0; v < NW - 1; ++v) if (v < w) xin[v] = *reinterpret_cast<const int4*>(&xch[i & (RING - 1)][v][0]);
    }
    // The SIMD's arbiter favours its older wave: of the 4 pairs of a CU the first finishes after 2.4 ms, the last after
    // 3.3 ms, and the SIMD idles behind the early finishers.  Alternating the user priority row by row (by the parity of
    // the wave's hardware slot) evens that out: -3...-6 % on a lone launch.  Launches that overlap on several streams
    // (bench.py) fill those gaps better and lose with it; launch_dp_affine_tag decides (one context alive -> on).
    const int par = (i ^ hwslot) & 1;
    if (prm.alt_prio & 0x100) {               // experiment: explicit levels, bits 0-1 / 4-5 = the row's bulk for even / odd parity
      setprio_dyn(par ? (prm.alt_prio >> 4) & 3 : prm.alt_prio & 3);
    } else if (prm.alt_prio == 1) {
      if (par) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    } else if (prm.alt_prio == 2) {           // experiment: the bulk of a row at low priority, its barrier-coupled end at high
      __builtin_amdgcn_s_setprio(0);
    } else if (prm.alt_prio == 3) {           // experiment: both
      if (par) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    const int rowv = rowv_next;
    rowv_next = tab_at(code_n1 * 128, lane_row4);
    code_n1 = (int)qcs[min(i + 2, Q - 1)];                                      // (clamped: only rows <= Q-2 are consumed)
    const int FK = (gi + ge * (i - 2)) * (1 << KB);                                   // F = gmx - FK
    const int RK = (((ge * (i - 1)) * (1 << KB)) | P_INS | (TAGMAX - (i - 1))) - P_MATCH;   // dk + RK = key(D[i-1][c] + ge (i-1), insertion from row i-1)
    const int colK = prm.free_ins ? 0 : FK;                                     // column 1: one insertion from the origin

    int bk[R][X];
    int sv[R][X];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) sv[r][x] = __builtin_amdgcn_ds_bpermute(code4[r][x], rowv);
    const int svB = (NW > 1) ? __builtin_amdgcn_ds_bpermute(codeB4, rowv) : 0;
    // cell phase: source column c -> best key of target column c+1 (before the target's similarity)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int pv = cvk[r];
#pragma unroll
      for (int x = 0; x < X; ++x) {
        const int m = dk[r][x];
        int A = ak[r][x];
        if (r == 0 && x == 0) A = (cb == 0) ? NEGK : A;     // column 0 is never a source (dpmatrix.h:459 starts at t0+1)
        const int e = pv - EK[r][x];
        const int f = gmx[r][x] - FK;
        bk[r][x] = max(max(m, e), f);                       // v_max3_i32: match > deletion > insertion on equal values
        pv = max(pv, A);
      }
    }
    // boundary target (first column of the next wave), finished by this wave's lane 63
    int dB = P_MATCH; uint32_t pB = kNullPtr;
    if (NW > 1) {
      int kh = bk[R - 1][X - 1] + svB;
      if (LOCAL) kh = max(kh, ZKEY);
      const bool in = CB <= T - 2;
      dB = in ? ((kh & ~LOW) | P_MATCH) : P_MATCH;
      pB = in ? (uint32_t)(kh & LOW) : kNullPtr;
    }
    // vertical state: row i-1 becomes an insertion source for row i+1
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) {
        gmx[r][x] = max(gmx[r][x], dk[r][x] + RK);
        asm volatile("" : "+v"(gmx[r][x]));                  // pin the update here: sunk to the loop latch it keeps row i-1 alive next to
      }                                                      // row i and costs a register copy per cell
    // shift one column right, add the target column's similarity, clip, split into score and pointer word
    int prev_k = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // wave_shr:1; lane 0 takes the previous group's last column (its "old" operand), group 0's lane 0 is a don't-care
      const int uk = (r == 0) ? __builtin_amdgcn_update_dpp(0, bk[r][X - 1], 0x138, 0xF, 0xF, true) : tdpp<0x138>(prev_k, bk[r][X - 1]);
      prev_k = __builtin_amdgcn_readlane(bk[r][X - 1], 63);
      const bool masked = (r == 0 && W0 == 0) || (W0 + GW * (r + 1) > T - 1);   // wave-uniform
      int sK1 = 0;
#pragma unroll
      for (int x = 0; x < X; ++x) {
        const int sK = sv[r][x];
        if (r == 0 && x == 1) sK1 = sK;
        int kh = ((x == 0) ? uk : bk[r][x - 1]) + sK;
        if (LOCAL) kh = max(kh, ZKEY);
        dk[r][x] = (kh & ~LOW) | P_MATCH;                    // v_and_or_b32
        pf[r][x] = (KBT == 16) ? (uint32_t)kh : (uint32_t)(kh & LOW);   // KB = 16: the store takes the low half of the whole key
      }
      if (masked) {                                          // one scalar branch per group; only the groups holding column 0/1 or columns >= T-1 pay
        asm volatile("" ::: "memory");                       // (if-converted, this block costs two v_cndmask per cell on every group)
        if (r == 0) {                                        // column 1 (dpmatrix.h:421-426 / :593-599), pointer (0,0): lane 0 of wave 0
          int v1 = sK1 - colK;
          if (LOCAL) v1 = max(v1, 0);
          const bool is1 = cb == 0;
          dk[0][1] = is1 ? (v1 | P_MATCH) : dk[0][1]; pf[0][1] = is1 ? (uint32_t)ORIGIN_INS : pf[0][1];
        }
#pragma unroll
        for (int x = 0; x < X; ++x) {
          dk[r][x] = (dk[r][x] & inm[r][x]) | P_MATCH;       // columns 0 and >= T-1: score 0, null pointer
          pf[r][x] |= ~(uint32_t)inm[r][x];
        }
      }
    }
    if (prm.alt_prio & 0x100) setprio_dyn(par ? (prm.alt_prio >> 6) & 3 : (prm.alt_prio >> 2) & 3);   // bits 2-3 / 6-7 = the row's end
    else if (prm.alt_prio == 2) __builtin_amdgcn_s_setprio(3);
    else if (prm.alt_prio == 3) { if (par) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
    finish_row(i, dB, pB, sync, xin);
    }
    if (NW > 1 && !sync && (it & (lag - 1)) == lag - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- not the pair's last segment: leave the state for whoever takes the next one, publish the item -------------------
  int pair_out = pair_id;
  if constexpr (SEGQ) {
    const int item = *reinterpret_cast<volatile int*>(&s_item);
    pair_out = item >> 3;
    const int seg_out = item & 7;
    if (seg_out + 1 < seg_count(Q, prm.ksegs)) {
    const __amdgpu_buffer_rsrc_t rs = state_rsrc(pair_out, seg_out);
    uint32_t wv[kStateChunks * 4] = {};
    state_words(wv, true);
#pragma unroll
    for (int c = 0; c < kStateChunks; ++c) {
      const u32x4 v = {wv[4 * c], wv[4 * c + 1], wv[4 * c + 2], wv[4 * c + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (c * 64 * NW + (int)threadIdx.x) * 16, 0, 16);             // aux 16 = sc1 (write-through)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its stores (the plane rows too) ...
    __syncthreads();                                       // ... before ONE lane publishes
    if (threadIdx.x == 0) {
      const int idx = __hip_atomic_fetch_add(&prm.queue[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      __hip_atomic_store(&prm.queue[16 + idx], item + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
    }
  }

  // ---- last row: untouched except the corner, which dp_corner_kernel writes --------------------------------
  if (Q >= 2) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int x = 0; x < X; ++x) { dk[r][x] = P_MATCH; pf[r][x] = kNullPtr; }
    store_row(Q - 1);
  }

  // ---- find_max partial (optimal.h:108-124): value and first row-major position over interior cells ----------
  if (LOCAL) {
    int m = lmax;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    uint32_t p = (lmax == m && (m >> KB) > 0) ? lpos : 0xFFFFFFFFu;
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
    if (threadIdx.x == 0) { res[pair_out].part_max = (float)(m >> KB); res[pair_out].part_pos = p; }
  } else {
    if (threadIdx.x == 0) { res[pair_out].part_max = 0.f; res[pair_out].part_pos = 0xFFFFFFFFu; }
  }
}

// ---- host side ------------------------------------------------------------------------------------------

// Tagged keys: integer values, and every intermediate inside the key's value field.
//   sequences up to 2048 (11 tag bits, 19 value bits): the crude bound (max|S| + ge)(Q + T) + gi + max|S| < 2^16;
//   up to 4096 (12 tag bits, 18 value bits, "minus infinity" at -114688): |D| <= max|S| min(Q,T) + gi + ge max(Q,T) (a score is a
//   sum of at most min(Q,T) similarities minus gaps, and at least the score of the path "diagonal, then one gap"); source keys
//   add ge * column or ge * row, candidates subtract gi + ge * length once more -> max|S| min + 2 gi + 3 ge max + max|S| < 100000,
//   and gi + ge * length < 16000 so that "minus infinity" minus a gap constant stays inside the field.
int tag_path_bits(const aln_batch* b, const float* table, int n, const aln_gap* gap) {
  if (!table || gap->model != ALN_GAP_AFFINE_CONST) return 0;
  if (b->maxQ > 4096 || b->maxT > 4096) return 0;
  const float gi = gap->gap_init, ge = gap->gap_extn;
  if (!(gi == (float)(int)gi) || !(ge == (float)(int)ge) || gi < 0 || ge < 0) return 0;
  double maxs = 0;
  for (int k = 0; k < n * n; ++k) {
    const float v = table[k];
    if (!(v == (float)(int)v)) return 0;
    if (fabs((double)v) > maxs) maxs = fabs((double)v);
  }
  const double mn = (double)std::min(b->maxQ, b->maxT), mx = (double)std::max(b->maxQ, b->maxT);
  if (b->maxQ <= 2048 && b->maxT <= 2048 && b->ctx->hints.tag_bits != 12)
    return (maxs + ge) * ((double)b->maxQ + (double)b->maxT) + gi + maxs < 65536.0 ? 11 : 0;
  return (maxs * mn + 2 * gi + 3 * ge * mx + maxs < 100000.0 && gi + ge * mx < 16000.0) ? 12 : 0;
}
bool tag_path_legal(const aln_batch* b, const float* table, int n, const aln_gap* gap) { return tag_path_bits(b, table, n, gap) != 0; }
// uint16 score planes of local tagged builds: every score is in [0, max|S| min(Q,T)]
bool tag_h16_legal(const aln_batch* b) {
  double maxs = 0;
  for (float v : b->h_table) maxs = std::max(maxs, fabs((double)v));
  return maxs * (double)std::min(b->maxQ, b->maxT) < 65536.0;
}

// The 16-bit key layout (score = high half of the key): local builds only (scores >= 0), best score + the A-space offset
// ge * column below 2^15, and the most negative real candidate -(gi + ge * length + max|S|) above the kernel's -8192.
bool tag_key16_legal(const aln_batch* b) {
  double maxs = 0;
  for (float v : b->h_table) maxs = std::max(maxs, fabs((double)v));
  const double gi = b->gap.gap_init, ge = b->gap.gap_extn;
  const double L = (double)std::max(b->maxQ, b->maxT), best = maxs * (double)std::min(b->maxQ, b->maxT);
  return best + ge * L + maxs < 32767.0 && gi + ge * L + maxs < 8000.0;
}

template <int NW, int R, int X, int TB>
static int launch_tag_variant(aln_batch* b, const TagParams& prm_in) {
  TagParams prm = prm_in;
  while (prm.lag * NW > 16) prm.lag >>= 1;                 // the ring has 16 slots: lag * (NW-1) + lag <= 16
  dim3 grid(b->n_pairs), block(64 * NW);
  hipStream_t st = b->ctx->stream;
  // segment queue (see the kernel): hint "tag_segments" = K: long pairs are cut into K segments (2 .. 8) when the batch alone fills
  // the GPU (>= 512 pairs); -K: whenever pairs are long; 0 / 1: off
  const int want = b->ctx->hints.tag_segments;
  prm.n_pairs = b->n_pairs; prm.queue = nullptr;
  prm.ksegs = std::min(std::abs(want), kSegs);
  bool segq = false;
  if (prm.ksegs >= 1 && (want < 0 || (b->n_pairs >= 512 && R * X == 16))) {      // (1: the queue kernel with whole pairs — diagnosis only)
    long items = 0;
    for (const PairDesc& d : b->h_pairs) items += seg_count(d.Q, prm.ksegs);
    if (items > b->n_pairs || prm.ksegs == 1) {
      constexpr size_t kStateBytes = (size_t)((2 * R * X + R + 2 + 3) / 4) * 16 * 64 * NW;
      const size_t qbytes = ((size_t)(16 + std::max<long>(items - b->n_pairs, 4)) * 4 + 15) & ~(size_t)15;
      const size_t sbytes = (size_t)b->n_pairs * kSegs * kStateBytes;
      if (b->tagq_bytes < qbytes || b->tagstate_bytes < sbytes) {
        ALN_HIP_CHECK(b->ctx, hipStreamSynchronize(st));
        hipFree(b->d_tagq); hipFree(b->d_tagstate); b->d_tagq = nullptr; b->d_tagstate = nullptr; b->tagq_bytes = b->tagstate_bytes = 0;
        ALN_HIP_CHECK(b->ctx, hipMalloc((void**)&b->d_tagq, qbytes));
        ALN_HIP_CHECK(b->ctx, hipMalloc((void**)&b->d_tagstate, sbytes));
        b->tagq_bytes = qbytes; b->tagstate_bytes = sbytes;
        ALN_HIP_CHECK(b->ctx, hipMemcpy(b->d_tagq + 4, &b->d_tagstate, 8, hipMemcpyHostToDevice));   // header words 4..5, once
      }
      ALN_HIP_CHECK(b->ctx, hipMemsetAsync(b->d_tagq, 0xFF, 16, st));                                 // counters -1, error word -1
      ALN_HIP_CHECK(b->ctx, hipMemsetAsync(b->d_tagq + 16, 0xFF, qbytes - 64, st));                  // every slot "empty"
      prm.queue = b->d_tagq;
      grid = dim3((unsigned)items);
      segq = true;
    }
  }
  b->tag_segmented = segq;
  const bool k16 = b->islocal && b->h_mode == 1 && tag_key16_legal(b) && b->ctx->hints.key16;
  // 16 cells per lane (155 VGPRs): two or three waves per SIMD.  Three pay when enough waves are in flight to give EVERY SIMD three
  // (launches of several contexts overlapping: -8 % per step in bench.py; or a lone launch of 1536 pairs); a lone 1024-pair launch
  // (2048 waves on 1024 SIMDs) is spread unevenly by the dispatcher then, 3 on some SIMDs and 1 on others (+20 %).  Context hint
  // "tag_occupancy": 2, 3, or 0 = by the size of this launch.
  const int occ_hint = b->ctx->hints.tag_occupancy;
  // by the size of the launch: waves per SIMD it brings (1024 SIMDs), rounds of 2 (3.2 ms each, measured on config 2) against rounds
  // of 3 (4.5 ms): 1024 pairs -> 2, 1536 -> 3 (4.6 vs 6.3 ms), 2048 -> 2 (6.9 vs 7.5 ms)
  const long wps = ((long)b->n_pairs * NW + 1023) / 1024;
  const bool auto3 = ((wps + 2) / 3) * 45 < ((wps + 1) / 2) * 32;
  const bool occ3 = R * X == 16 && !segq && (occ_hint == 3 || (occ_hint == 0 && auto3));
#define ALN_TAG_LAUNCH_O(LOC_, H16_, KB_, SQ_, OCC_)                                                                                 \
  hipLaunchKernelGGL((dp_affine_tag_kernel<NW, R, LOC_, H16_, KB_, X, SQ_, TB, OCC_>), grid, block, 0, st, b->d_pairs, b->d_qcodes, b->d_tcodes, \
                     b->d_table32, b->d_H, b->d_P, b->d_res, prm)
#define ALN_TAG_LAUNCH(LOC_, H16_, KB_, SQ_) do {                                                                                    \
    if constexpr (R * X == 16 && !(SQ_)) { if (occ3) ALN_TAG_LAUNCH_O(LOC_, H16_, KB_, SQ_, 3); else ALN_TAG_LAUNCH_O(LOC_, H16_, KB_, SQ_, 2); } \
    else ALN_TAG_LAUNCH_O(LOC_, H16_, KB_, SQ_, 2);                                                                                  \
  } while (0)
#define ALN_TAG_LAUNCH_Q(LOC_, H16_, KB_) do { if (segq) ALN_TAG_LAUNCH(LOC_, H16_, KB_, true); else ALN_TAG_LAUNCH(LOC_, H16_, KB_, false); } while (0)
  if (k16) ALN_TAG_LAUNCH_Q(true, true, 16);
  else if (b->islocal && b->h_mode == 1) ALN_TAG_LAUNCH_Q(true, true, 13);
  else if (b->islocal) ALN_TAG_LAUNCH_Q(true, false, 13);
  else ALN_TAG_LAUNCH_Q(false, false, 13);
#undef ALN_TAG_LAUNCH_Q
#undef ALN_TAG_LAUNCH
#undef ALN_TAG_LAUNCH_O
  char nm[96];
  snprintf(nm, sizeof nm, "dp_affine_tag_kernel<NW=%d,R=%d,%s%s%s%s%s%s>%s", NW, R, X == 8 ? "X=8," : "", b->islocal ? "local" : "global",
           b->h_mode == 1 ? ",h16" : "", k16 ? ",key16" : "", TB == 12 ? ",tag12" : "", occ3 ? ",occ3" : "", segq ? "+segq" : "");
  b->kernel_name = nm;
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

int launch_dp_affine_tag(aln_batch* b) {
  TagParams prm;
  prm.gi = (int)b->gap.gap_init;
  prm.ge = (int)b->gap.gap_extn;
  prm.free_del = b->gapdev.free_del;
  prm.free_ins = b->gapdev.free_ins;
  // row-alternating wave priority: a per-context hint (aln_ctx_set_hint "tag_alt_prio"): it pays while launches follow each other
  // on one stream (the arbiter's favouritism costs ~6 %) and loses when the caller overlaps launches of several contexts
  prm.alt_prio = b->ctx->hints.tag_alt_prio;
  prm.lag = b->ctx->hints.tag_lag;
  if (prm.lag < 0 || prm.lag > 4 || (prm.lag & (prm.lag - 1))) prm.lag = 0;      // 0, 1, 2 or 4: lag * (NW-1) + lag <= 16 slots for NW <= 4
  const int ld = b->maxld;
  // variant = waves per pair, groups per lane, consecutive columns a lane owns in a group (ALN_DP_VARIANT="NW,R[,X]")
  int nw = b->ctx->hints.dp_nw, r = b->ctx->hints.dp_r, x = b->ctx->hints.dp_x ? b->ctx->hints.dp_x : 4;
  // sequences beyond 2048 need 12 tag bits (pointer dialect 2): one instantiation, 4 waves x 2 groups x 8 columns = 4096 columns
  if (b->ptr_mode == 2) return launch_tag_variant<4, 2, 8, 12>(b, prm);
  if (nw == 0) {
    if (ld <= 256) { nw = 1; r = 1; }
    else if (ld <= 512) { nw = 2; r = 1; }
    else if (ld <= 1024) { nw = 4; r = 1; }
    else { nw = 2; r = 2; x = 8; }   // 2 waves x 2 groups x 8 columns per lane: half the row scans of (2,4,4)
  }
  if (64 * x * nw * r < ld) return ALN_E_TOO_LONG;
#define ALN_V(NW_, R_, X_) if (nw == NW_ && r == R_ && x == X_) return launch_tag_variant<NW_, R_, X_, 11>(b, prm)
  ALN_V(1, 1, 4); ALN_V(1, 2, 4); ALN_V(1, 4, 4); ALN_V(1, 8, 4);
  ALN_V(2, 1, 4); ALN_V(2, 2, 4); ALN_V(2, 4, 4);
  ALN_V(4, 1, 4); ALN_V(4, 2, 4);
  ALN_V(8, 1, 4);
  ALN_V(1, 4, 8); ALN_V(2, 1, 8); ALN_V(2, 2, 8); ALN_V(4, 1, 8);
#undef ALN_V
  return ALN_E_ARG;
}

}  // namespace aln
