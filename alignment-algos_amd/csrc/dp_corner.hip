// dp_corner.hip — the final cell of a build and the end of find_max (gfx950).
//
// Reference: dpmatrix.h:505-534 / :655-687 (forward: cell (q1,t1) scans the whole last interior row and
// column with the evaluator's end-gap rules), :844-874 / :995-1028 (reverse: cell (q0,t0)), the two
// degenerate shortcuts :375-390 / :558-573 / :713-728 / :899-914, and the seed rule of
// Optimal::find_max (optimal.h:108-124) / Optimal_Rev::find_max (optimal_rev.h:117-131).
// Four waves per pair (the loop is one dependent memory round trip per 256 candidates); O(Q+T) reads of the finished score plane, literal fp32 arithmetic
// (s = D; s -= g; s += S; clip), so it serves the integer fast path and the exact path alike.
// Candidate order = match, deletions k ascending (in the build frame), insertions k ascending;
// "replace on strict >" == the first candidate reaching the maximum.
#include "aln_device.h"

namespace aln {

constexpr int kCornerThreads = 256;

__global__ __launch_bounds__(kCornerThreads) void dp_corner_kernel(const PairDesc* __restrict__ pairs, EvalDev proto,
                                                       const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                       const float* __restrict__ tgi, const float* __restrict__ tge,
                                                       float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
                                                       const float* __restrict__ Sbase, PairResult* __restrict__ res,
                                                       int islocal, int full_build, int rev, int bug_b4, int ptr_mode, int h_mode,
                                                       const int* __restrict__ dp_error) {
  const PairDesc pd = pairs[blockIdx.x];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  bind_table_model(e, proto, pd);
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  auto HV = [&](int i, int j) -> float { return load_score(Hbase, pd.plane_off, pd.ld, i, j, h_mode); };
  __shared__ float red_s[kCornerThreads / 64];
  __shared__ int red_i[kCornerThreads / 64], red_h[kCornerThreads / 64];
  const int ld = pd.ld, lane = threadIdx.x;          // "lane" = thread of the pair's workgroup; thread 0 writes the results
  const Frame f = {pd.q0, pd.q1, pd.t0, pd.t1, rev};
  const int nQ = f.nQ(), nT = f.nT();
  const bool local = islocal != 0;
  if (nQ <= 0 || nT <= 0) { if (lane == 0) res[blockIdx.x].status = ALN_E_BOUNDS; return; }
  const int fq = f.rq(nQ), ft = f.rt(nT);         // the final cell, real coordinates
  float corner;
  uint32_t cptr;
  if (nQ == 1) {                      // boundary conditions force a deletion — no clip in any builder
    float s = 0.f;                    // build()/build_subdpm() zero the origin (dpmatrix.h:306-307, :333-334)
    s -= frame_del(e, f, 0, nT);
    s += dev_sim(e, fq, ft);
    corner = s; cptr = pack_ptr(f.rq(0), f.rt(0));
  } else if (nT == 1) {               // ... or an insertion
    float s = 0.f;
    s -= frame_ins(e, f, 0, nQ, 0, 1);
    s += dev_sim(e, fq, ft);
    corner = s; cptr = pack_ptr(f.rq(0), f.rt(0));
  } else {
    const float sc = dev_sim(e, fq, ft);
    const int ndel = nT - 1;          // k = 1 .. nT-1 (frame columns of row nQ-1)
    const int nins = nQ - 1;          // k = 1 .. nQ-1 (frame rows of column nT-1)
    const int ncand = 1 + ndel + nins;
    float bs = 0.f; int bi = 0x7FFFFFFF;
    bool have = false;
    for (int idx = lane; idx < ncand; idx += kCornerThreads) {
      float s;
      if (idx == 0) {
        s = HV(f.rq(nQ - 1), f.rt(nT - 1)) + sc;
      } else if (idx <= ndel) {
        s = HV(f.rq(nQ - 1), f.rt(idx));
        s -= frame_del(e, f, idx, nT);
        s += sc;
      } else {
        int k = idx - ndel;
        s = HV(f.rq(k), f.rt(nT - 1));
        s -= frame_ins(e, f, k, nQ, nT - 1, nT);
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
    // ... then across the waves, same rule
    if ((lane & 63) == 0) { red_s[lane >> 6] = bs; red_i[lane >> 6] = bi; red_h[lane >> 6] = (int)have; }
    __syncthreads();
    for (int w = 0; w < kCornerThreads / 64; ++w) {
      const float os = red_s[w]; const int oi = red_i[w]; const int oh = red_h[w];
      bool take = oh && (!have || os > bs || (os == bs && oi < bi));
      bs = take ? os : bs; bi = take ? oi : bi; have = have || oh;
    }
    corner = bs;
    if (bi == 0) cptr = pack_ptr(f.rq(nQ - 1), f.rt(nT - 1));
    else if (bi <= ndel) cptr = pack_ptr(f.rq(nQ - 1), f.rt(bi));
    else {
      // dpmatrix.h:868 stores t1_m1 (= frame column 1) instead of t0_p1 in the reverse GLOBAL builder (B4)
      int bt = (rev && !local && bug_b4) ? f.rt(1) : f.rt(nT - 1);
      cptr = pack_ptr(f.rq(bi - ndel), bt);
    }
  }
  if (lane == 0) {
    store_score(Hbase, pd.plane_off, ld, fq, ft, h_mode, corner);
    {
      int cpq, cpt;
      decode_ptr(cptr, 0, fq, ft, cpq, cpt);
      // the final cell speaks the plane's pointer dialect
      store_ptr_word(Pbase, pd.plane_off, ld, fq, ft, ptr_mode, encode_ptr(ptr_mode, fq, ft, cpq, cpt));
    }
    PairResult r = res[blockIdx.x];
    r.corner = corner;
    r.status = (dp_error && *dp_error == 1) ? ALN_E_HIP : 0;   // the DP kernel's segment queue gave up waiting (never expected)
    if (local && full_build) {
      // find_max: the seed keeps ties, otherwise the first strictly greater cell of the scan wins.
      // forward (optimal.h:111-113): seed (Q-2,T-2); reverse (optimal_rev.h:120-122): seed (0,0) = the final cell.
      const int sq = rev ? fq : pd.Q - 2, st = rev ? ft : pd.T - 2;
      const float seed = rev ? corner : HV(sq, st);
      if (r.part_pos != 0xFFFFFFFFu && seed < r.part_max) {
        r.best = r.part_max; r.best_q = (int)(r.part_pos >> 16); r.best_t = (int)(r.part_pos & 0xFFFFu);
      } else {
        r.best = seed; r.best_q = sq; r.best_t = st;
      }
    } else {
      r.best = corner; r.best_q = fq; r.best_t = ft;
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
  proto.tcn = b->d_tcn; proto.deltab = b->d_deltab; proto.deltab_off = b->d_deltab_off; proto.instab = b->d_instab;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model != ALN_GAP_AFFINE_CONST;
  hipLaunchKernelGGL(dp_corner_kernel, dim3(b->n_pairs), dim3(kCornerThreads), 0, b->ctx->stream, b->d_pairs, proto,
                     sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr,
                     tpos ? b->d_tge : nullptr, b->d_H, b->d_P, sub ? nullptr : b->d_S, b->d_res,
                     (int)b->islocal, (int)!b->have_sub, (int)(b->direction == ALN_REV), (int)b->bug_b4, (int)b->ptr_mode, (int)b->h_mode,
                     b->tag_segmented ? b->d_tagq + 2 : nullptr);
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

}  // namespace aln
