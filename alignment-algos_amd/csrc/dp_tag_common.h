// dp_tag_common.h — what the tagged-key row-sweep kernels share (dp_affine_tag.hip: several waves per pair; dp_affine_solo.hip:
// one wave per pair): the key layout, the launch parameters, the DPP prefix-maximum scans.
#pragma once
#include "aln_internal.h"

namespace aln {

// Key layout for TB tag bits (TB = 11: sequences up to 2048, TB = 12: up to 4096; pointer-word dialects 1 and 2 of
// aln_device.h::decode_ptr):  key = value << (TB+2) | prio << TB | tag,  tag = (2^TB - 1) - k.
template <int TB>
struct TagBits {
  static constexpr int TAGMAX = (1 << TB) - 1;
  static constexpr int P_MATCH = 3 << TB, P_DEL = 2 << TB, P_INS = 1 << TB;
  // "minus infinity": below every real value, with headroom for one subtraction of a gap constant.
  // TB 11 (value << 13): -2^17, real values stay inside +-2^16.  TB 12 (value << 14, range +-2^17): -114688, real values inside +-100000.
  static constexpr int NEGK = (TB == 11) ? -(1 << 30) : -(7 << 28);
  static constexpr int ZKEY = P_MATCH;               // (value 0, match): the clip of local alignments
  static constexpr int ORIGIN_DEL = P_DEL | TAGMAX;  // pointer (i-1, 0): cells of row 1 come from the origin by one deletion
  static constexpr int ORIGIN_INS = P_INS | TAGMAX;  // pointer (0, j-1): cells of column 1
};

struct TagParams {
  int gi, ge;
  int free_del, free_ins;
  int alt_prio;
  int lag;        // 0: the waves of a pair exchange their row state synchronously (write, barrier, read inside every row);
                  // L = 2^k >= 1: wave w runs L*w rows behind wave 0 and reads what the earlier waves left in a ring of
                  // exchange slots, one workgroup barrier every L rows (see "skewed exchange" below)
  // ---- segment queue (see "Segment queue" in the kernel) ----
  int n_pairs;
  int ksegs;      // segments a long pair is cut into (2 .. 8)
  int* queue;     // SEGQ kernels: [0] ticket counter, [1] push counter, [2] error word — all 0xFFFFFFFF before a launch;
                  // [4..5] address of the hand-off slots ((pair * kSegs + segment) * kStateBytes), written once;
                  // [16 ...] item slots, 0xFFFFFFFF before a launch
};

// Segment queue.  With one workgroup per pair a batch of 1024 pairs x 2 waves fills the 1024 SIMDs exactly once: the SIMD arbiter
// favours its older wave, so pairs finish between 2.4 and 3.3 ms, XCDs differ by 6-7 %, and everything that finishes early
// idles.  In segment mode a pair's interior rows are cut into kSegs consecutive segments of decreasing length (6:5:4:3:2:1) and a
// launch has one workgroup per (pair, segment).  A workgroup takes a TICKET when it starts (atomic counter).  Tickets below
// n_pairs mean "first segment of pair <ticket>"; every later ticket waits for the next item in a queue that finishing workgroups
// feed: whoever completes segment s of a pair stores the pair's row state (previous row, per-column insertion maxima, scan
// carries, running find_max) to a hand-off slot and pushes (pair, s+1).  So a freed SIMD slot continues whichever pair became
// ready first, on whichever XCD has room, until the whole batch is done; only the last, short segments form a tail.
// No deadlock: an item is only ever held by a workgroup that has started, it never waits once it has its item, and the k-th
// waiting ticket needs only the k-th push, which the k-th completion of a non-final segment delivers.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16, form R1): the state goes out
// with write-through (sc1) 16-byte stores, every wave drains them (s_waitcnt vmcnt(0)), a workgroup barrier, then ONE lane
// publishes the item with an agent-scope atomic store; the consumer polls that one word (relaxed, agent scope), executes one
// agent-scope acquire, and the workgroup reads the state with sc1 loads behind a barrier.  Every hand-off slot is written once
// and read once per launch (slot = pair x segment), so no line is rewritten while a stale copy could sit in another XCD's L2.
constexpr int kSegs = 8;                               // most segments a pair can have (item = pair * 8 + segment)
constexpr int kSegMinRows = 512;                       // pairs with fewer rows are one segment
__host__ __device__ inline int seg_count(int Q, int ksegs) { return Q >= kSegMinRows ? ksegs : 1; }
// first interior row of segment s (s = 0 .. n): interior rows are 2 .. Q-2; segment lengths in the ratio n : n-1 : ... : 1
__host__ __device__ inline int seg_bound(int Q, int s, int n) {
  if (n == 1) return s == 0 ? 2 : Q - 1;
  if (s >= n) return Q - 1;
  const int rows = Q - 3 > 0 ? Q - 3 : 0;
  const long cum = (long)s * n - (long)s * (s - 1) / 2, total = (long)n * (n + 1) / 2;
  return 2 + (int)((rows * cum) / total);
}

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int tdpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
// inclusive max-scan over the wave; identity INT_MIN lets every step fuse into one v_max_i32_dpp
__device__ __forceinline__ int wave_incl_max_key(int v) {
  const int ident = (int)0x80000000;
  v = max(v, tdpp<0x111>(ident, v));
  v = max(v, tdpp<0x112>(ident, v));
  v = max(v, tdpp<0x114>(ident, v));
  v = max(v, tdpp<0x118>(ident, v));
  v = max(v, tdpp<0x142, 0xA>(ident, v));
  v = max(v, tdpp<0x143, 0xC>(ident, v));
  return v;
}
// the same scan over R independent values, stage by stage
template <int R>
__device__ __forceinline__ void wave_incl_max_keys(int (&v)[R]) {
  const int ident = (int)0x80000000;
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x111>(ident, v[r]));
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x112>(ident, v[r]));
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x114>(ident, v[r]));
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x118>(ident, v[r]));
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x142, 0xA>(ident, v[r]));
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = max(v[r], tdpp<0x143, 0xC>(ident, v[r]));
}

// KBT = 13: value in bits 13..31 (|value| < 2^18).  KBT = 16 (local builds with 16-bit planes whose values provably fit 15
// bits): the score is the key's high half and the pointer word its low half, so two cells pack into one plane word with a
// single v_perm_b32 each and the pointer never has to be extracted.
// 16 cells per lane need ~170 VGPRs, which would let the dispatcher place up to 3 waves on a SIMD.  A batch of 1024 pairs x 2 waves
// is exactly 2 waves per SIMD, but the dispatcher does not spread them evenly on its own (3 on some SIMDs, 1 on others: measured
// +20 % kernel time, and it varies with unrelated code changes).  amdgpu_waves_per_eu(2,2) makes the compiler allocate for
// exactly two waves per SIMD (it rounds the VGPR allocation up so that a third cannot be placed), which caps every SIMD at 2.

}  // namespace aln
