// Device-side helpers shared by the exact-arithmetic kernels (corner cell, exact DP, enumeration).
// All float arithmetic here follows the reference's operation order; the library is compiled with
// -ffp-contract=off so no mul+add is fused.
#pragma once
#include "aln_internal.h"

namespace aln {

__device__ __forceinline__ float clip0(float s, bool local) {   // std::max(0.f, s) of dpmatrix.h:580
  return local ? ((0.f < s) ? s : 0.f) : s;
}
__device__ __forceinline__ float fminr(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b)

// Everything a kernel needs to evaluate similarity / deletion / insertion for one pair.
struct EvalDev {
  int Q, T;
  int model, align_type;
  float gi, ge;
  const float* tgi;     // template-position gap arrays of THIS pair's template (AFFINE_TPOS_MIN / DEL_TABLE_INS_TPOS) or nullptr
  const float* tge;
  const float* tcn;     // DEL_TABLE_INS_TPOS: Gn2Eval's v_cn; in the kernel-argument prototype these three are POOL bases
  const float* deltab;  // DEL_TABLE_INS_TPOS: this template's T x T deletion table (prototype: base of all tables)
  const int64_t* deltab_off;   // prototype only: first element of template sequence s's table
  const float* instab;  // TABLES: this pair's three T x Q insertion planes (prototype: base of all pairs' planes)
  // similarity
  int sim_kind;                 // ALN_SIM_SUBMATRIX: codes + table; else plane
  const uint8_t* qc; const uint8_t* tc;
  const float* tablef;          // 32 x 32
  const float* S; int ld;       // plane
};

// Evaluator::deletion — aasubalib.h:27-51 / hmap2_eval.h:41-67
__device__ __forceinline__ float dev_deletion(const EvalDev& e, int t1, int t2) {
  const bool free_end = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
  if (e.model == ALN_GAP_AFFINE_CONST) {
    int len = t2 - t1 - 1;
    if (len < 1) return 0.f;
    if (free_end && (t1 == 0 || t2 == e.T - 1)) return 0.f;
    return e.gi + e.ge * (float)(len - 1);
  } else if (e.model == ALN_GAP_DEL_TABLE_INS_TPOS || e.model == ALN_GAP_TABLES) {
    return e.deltab[(size_t)t1 * e.T + t2];     // gn2_eval.h:100-130 / any deletion(t1,t2), materialised by the caller
  } else {
    int dist = t2 - t1;
    if (dist < 2) return 0.f;
    float gi = fminr(e.tgi[t1], e.tgi[t2]);
    float ge = fminr(e.tge[t1], e.tge[t2]);
    if (free_end && (t1 == 0 || t2 == e.T - 1)) return 0.f;
    return gi + ge * (float)(dist - 2);
  }
}
// Evaluator::insertion — aasubalib.h:53-77 / hmap2_eval.h:69-95 (coefficients from TEMPLATE positions t1,t2)
__device__ __forceinline__ float dev_insertion(const EvalDev& e, int q1, int q2, int t1, int t2) {
  const bool free_end = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
  if (e.model == ALN_GAP_AFFINE_CONST) {
    int len = q2 - q1 - 1;
    if (len < 1) return 0.f;
    if (free_end && (q1 == 0 || q2 == e.Q - 1)) return 0.f;
    return e.gi + e.ge * (float)(len - 1);
  } else if (e.model == ALN_GAP_TABLES) {               // the evaluator's own insertion(), tabulated: interior / head / tail
    const size_t QT = (size_t)e.Q * e.T;
    if (q1 == 0) return e.instab[QT + (size_t)t1 * e.Q + q2];
    if (q2 == e.Q - 1) return e.instab[2 * QT + (size_t)t1 * e.Q + q1];
    return e.instab[(size_t)t1 * e.Q + (q2 - q1)];
  } else if (e.model == ALN_GAP_DEL_TABLE_INS_TPOS) {   // gn2_eval.h:132-165: coefficients of t1 only, plus the contact-number term
    int dist = q2 - q1;
    if (dist < 2) return 0.f;
    float gp = e.tgi[t1] + e.tge[t1] * (float)(dist - 2);
    gp = gp + e.tcn[t1];
    if (free_end && (q1 == 0 || q2 == e.Q - 1)) return 0.f;
    return gp;
  } else {
    int dist = q2 - q1;
    if (dist < 2) return 0.f;
    float gi = fminr(e.tgi[t1], e.tgi[t2]);
    float ge = fminr(e.tge[t1], e.tge[t2]);
    if (free_end && (q1 == 0 || q2 == e.Q - 1)) return 0.f;
    return gi + ge * (float)(dist - 2);
  }
}
// per-pair view of the table-model arrays (the prototype carries pool bases)
__device__ __forceinline__ void bind_table_model(EvalDev& e, const EvalDev& proto, const PairDesc& pd) {
  e.tcn = proto.tcn ? proto.tcn + pd.t_off : nullptr;
  e.deltab = proto.deltab ? proto.deltab + proto.deltab_off[pd.t_seq] : nullptr;
  e.instab = proto.instab ? proto.instab + pd.ins_off : nullptr;
}
// DPMatrix::getSim — SimilarityMatrix (simmatrix.h:51-72): zero borders, Evaluator::similarity inside
__device__ __forceinline__ float dev_sim(const EvalDev& e, int i, int j) {
  if (e.sim_kind == ALN_SIM_SUBMATRIX) {
    if (i <= 0 || j <= 0 || i >= e.Q - 1 || j >= e.T - 1) return 0.f;
    return e.tablef[(int)e.qc[i] * 32 + (int)e.tc[j]];
  }
  return e.S[(size_t)i * e.ld + j];
}

// A build rectangle and direction: frame coordinates (a,b) run from the origin (0,0) = (q0,t0) forward /
// (q1,t1) reverse to the final cell (nQ,nT); a reverse build is the forward programme in the mirrored frame.
struct Frame {
  int q0, q1, t0, t1, rev;
  __device__ __forceinline__ int nQ() const { return q1 - q0; }
  __device__ __forceinline__ int nT() const { return t1 - t0; }
  __device__ __forceinline__ int rq(int a) const { return rev ? q1 - a : q0 + a; }
  __device__ __forceinline__ int rt(int b) const { return rev ? t1 - b : t0 + b; }
};

// gap costs between two frame positions (lo < hi in the frame); the evaluator sees real, ordered positions
__device__ __forceinline__ float frame_del(const EvalDev& e, const Frame& f, int blo, int bhi) {
  int x = f.rt(blo), y = f.rt(bhi);
  return dev_deletion(e, x < y ? x : y, x < y ? y : x);
}
__device__ __forceinline__ float frame_ins(const EvalDev& e, const Frame& f, int alo, int ahi, int blo, int bhi) {
  int x = f.rq(alo), y = f.rq(ahi), u = f.rt(blo), v = f.rt(bhi);
  return dev_insertion(e, x < y ? x : y, x < y ? y : x, u < v ? u : v, u < v ? v : u);
}

__device__ __forceinline__ uint32_t pack_ptr(int pq, int pt) { return ((uint32_t)pq << 16) | ((uint32_t)pt & 0xFFFFu); }

// Pointer words of the P plane.  mode 0 (dp_affine_int, dp_exact): prev_q << 16 | prev_t.  mode 1 (dp_affine_tag):
// the low 13 bits of the winning key: prio << 11 | tag, prio 3 = match -> (i-1,j-1), 2 = deletion -> (i-1, 2047-tag),
// 1 = insertion -> (2047-tag, j-1).  mode 2: the same with 12 tag bits (prio << 12 | tag, 4095 - tag; sequences up to 4096).
// 0xFFFFFFFF = untouched cell (DPCell::null, null) in all.
__host__ __device__ __forceinline__ void decode_ptr(uint32_t w, int mode, int i, int j, int& pq, int& pt) {
  if (w == 0xFFFFFFFFu) { pq = -1; pt = -1; return; }
  if (mode == 0) { pq = (int)(w >> 16); pt = (int)(w & 0xFFFFu); if (pq == 0xFFFF) pq = -1; if (pt == 0xFFFF) pt = -1; return; }
  const int tb = mode == 2 ? 12 : 11, tmax = (1 << tb) - 1;          // dialect 2: 12 tag bits (sequences up to 4096)
  const int prio = (int)((w >> tb) & 3u), k = tmax - (int)(w & (uint32_t)tmax);
  if (prio == 3) { pq = i - 1; pt = j - 1; }
  else if (prio == 2) { pq = i - 1; pt = k; }
  else { pq = k; pt = j - 1; }
}
// In mode 1 the plane holds 16-bit words (13 used; 0xFFFF = untouched) at the same ELEMENT offsets as the fp32 score
// plane, i.e. it occupies the first half of the bytes reserved for a 32-bit plane: 6 instead of 8 bytes per cell.
__host__ __device__ __forceinline__ uint32_t load_ptr_word(const uint32_t* Pbase, int64_t plane_off, int ld, int i, int j, int mode) {
  const size_t e = (size_t)plane_off + (size_t)i * ld + j;
  if (mode == 0) return Pbase[e];
  const uint16_t w = reinterpret_cast<const uint16_t*>(Pbase)[e];
  return w == 0xFFFFu ? 0xFFFFFFFFu : (uint32_t)w;
}
__host__ __device__ __forceinline__ void store_ptr_word(uint32_t* Pbase, int64_t plane_off, int ld, int i, int j, int mode, uint32_t w) {
  const size_t e = (size_t)plane_off + (size_t)i * ld + j;
  if (mode == 0) Pbase[e] = w;
  else reinterpret_cast<uint16_t*>(Pbase)[e] = (uint16_t)(w & 0xFFFFu);
}
// Score plane: hmode 0 = fp32; hmode 1 = uint16 at the same element offsets (local builds of the tagged kernel: every
// score is an integer in [0, 65535]), 2 instead of 4 bytes per cell.
__host__ __device__ __forceinline__ float load_score(const float* Hbase, int64_t plane_off, int ld, int i, int j, int hmode) {
  const size_t e = (size_t)plane_off + (size_t)i * ld + j;
  if (hmode == 0) return Hbase[e];
  return (float)reinterpret_cast<const uint16_t*>(Hbase)[e];
}
__host__ __device__ __forceinline__ void store_score(float* Hbase, int64_t plane_off, int ld, int i, int j, int hmode, float v) {
  const size_t e = (size_t)plane_off + (size_t)i * ld + j;
  if (hmode == 0) Hbase[e] = v;
  else reinterpret_cast<uint16_t*>(Hbase)[e] = (uint16_t)(int)v;
}
__host__ __device__ __forceinline__ uint32_t encode_ptr(int mode, int i, int j, int pq, int pt) {
  if (mode == 0) return ((uint32_t)pq << 16) | ((uint32_t)pt & 0xFFFFu);
  const int tb = mode == 2 ? 12 : 11, tmax = (1 << tb) - 1;
  if (pq == i - 1 && pt == j - 1) return 3u << tb;
  if (pq == i - 1) return (2u << tb) | (uint32_t)(tmax - pt);
  return (1u << tb) | (uint32_t)(tmax - pq);
}

}  // namespace aln
