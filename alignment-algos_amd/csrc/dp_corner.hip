// dp_corner.hip — the final cell of a forward build and the end of find_max (gfx950).
//
// Reference: dpmatrix.h:505-534 / :655-687 (the cell (q1,t1) scans the whole last interior row and
// column with the evaluator's end-gap rules), the two degenerate shortcuts :375-390 / :558-573, and
// Optimal::find_max's seed rule (optimal.h:108-124).  One wave per pair; O(Q+T) reads of the finished
// score plane, literal fp32 arithmetic (s = D; s -= g; s += S; clip), so it serves the integer fast
// path and the exact path alike.  Candidate order = match, deletions k ascending, insertions k
// ascending; "replace on strict >" == the first candidate reaching the maximum.
#include "aln_device.h"

namespace aln {

__global__ __launch_bounds__(64) void dp_corner_kernel(const PairDesc* __restrict__ pairs, EvalDev proto,
                                                       const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                       const float* __restrict__ tgi, const float* __restrict__ tge,
                                                       float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
                                                       const float* __restrict__ Sbase, PairResult* __restrict__ res,
                                                       int islocal, int full_build) {
  const PairDesc pd = pairs[blockIdx.x];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  float* H = Hbase + pd.plane_off;
  uint32_t* P = Pbase + pd.plane_off;
  const int ld = pd.ld, lane = threadIdx.x;
  const int q0 = pd.q0, q1 = pd.q1, t0 = pd.t0, t1 = pd.t1;
  const bool local = islocal != 0;
  const float s_initial = (lane == 0) ? 0.f : 0.f;   // build()/build_subdpm() zero the origin (dpmatrix.h:306, :333)
  float corner;
  uint32_t cptr;
  if (q1 <= q0 || t1 <= t0) { if (lane == 0) res[blockIdx.x].status = ALN_E_BOUNDS; return; }
  if (q1 == q0 + 1) {                 // dpmatrix.h:375-381 / :558-564 — no clip
    float s = s_initial;
    s -= dev_deletion(e, t0, t1);
    s += dev_sim(e, q1, t1);
    corner = s; cptr = pack_ptr(q0, t0);
  } else if (t1 == t0 + 1) {          // :384-390 / :567-573
    float s = s_initial;
    s -= dev_insertion(e, q0, q1, t0, t1);
    s += dev_sim(e, q1, t1);
    corner = s; cptr = pack_ptr(q0, t0);
  } else {
    const float sc = dev_sim(e, q1, t1);
    const int ndel = t1 - 1 - t0;     // k = t0+1 .. t1-1
    const int nins = q1 - 1 - q0;     // k = q0+1 .. q1-1
    const int ncand = 1 + ndel + nins;
    float bs = 0.f; int bi = 0x7FFFFFFF;
    bool have = false;
    for (int idx = lane; idx < ncand; idx += 64) {
      float s;
      if (idx == 0) {
        s = H[(size_t)(q1 - 1) * ld + (t1 - 1)] + sc;
      } else if (idx <= ndel) {
        int k = t0 + idx;
        s = H[(size_t)(q1 - 1) * ld + k];
        s -= dev_deletion(e, k, t1);
        s += sc;
      } else {
        int k = q0 + (idx - ndel);
        s = H[(size_t)k * ld + (t1 - 1)];
        s -= dev_insertion(e, k, q1, t1 - 1, t1);
        s += sc;
      }
      s = clip0(s, local);
      if (!have || s > bs) { bs = s; bi = idx; have = true; }
    }
    // wave reduction: larger value wins, equal values -> smaller candidate index
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      float os = __shfl_xor(bs, o); int oi = __shfl_xor(bi, o); int oh = __shfl_xor((int)have, o);
      bool take = oh && (!have || os > bs || (os == bs && oi < bi));
      bs = take ? os : bs; bi = take ? oi : bi; have = have || oh;
    }
    corner = bs;
    if (bi == 0) cptr = pack_ptr(q1 - 1, t1 - 1);
    else if (bi <= ndel) cptr = pack_ptr(q1 - 1, t0 + bi);
    else cptr = pack_ptr(q0 + (bi - ndel), t1 - 1);
  }
  if (lane == 0) {
    H[(size_t)q1 * ld + t1] = corner;
    P[(size_t)q1 * ld + t1] = cptr;
    PairResult r = res[blockIdx.x];
    r.corner = corner;
    r.status = 0;
    if (local && full_build) {
      // find_max (optimal.h:108-124): seed (Q-2,T-2) keeps ties, otherwise the first strictly greater cell
      const int sq = pd.Q - 2, st = pd.T - 2;
      const float seed = H[(size_t)sq * ld + st];
      if (r.part_pos != 0xFFFFFFFFu && seed < r.part_max) {
        r.best = r.part_max; r.best_q = (int)(r.part_pos >> 16); r.best_t = (int)(r.part_pos & 0xFFFFu);
      } else {
        r.best = seed; r.best_q = sq; r.best_t = st;
      }
    } else {
      r.best = corner; r.best_q = q1; r.best_t = t1;
    }
    res[blockIdx.x] = r;
  }
}

int launch_dp_corner(aln_batch* b) {
  EvalDev proto = {};
  proto.model = b->gapdev.model;
  proto.align_type = b->gapdev.align_type;
  proto.gi = b->gapdev.gi; proto.ge = b->gapdev.ge;
  proto.sim_kind = (b->sim_kind == ALN_SIM_SUBMATRIX) ? ALN_SIM_SUBMATRIX : ALN_SIM_MATRIX;
  proto.tablef = b->d_tablef;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model == ALN_GAP_AFFINE_TPOS_MIN;
  hipLaunchKernelGGL(dp_corner_kernel, dim3(b->n_pairs), dim3(64), 0, b->ctx->stream, b->d_pairs, proto,
                     sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr,
                     tpos ? b->d_tge : nullptr, b->d_H, b->d_P, sub ? nullptr : b->d_S, b->d_res,
                     (int)b->islocal, (int)!b->have_sub);
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

}  // namespace aln
