// traceback.hip — Optimal / Optimal_Rev / Optimal_Subali pointer traceback on the resident planes (gfx950).
//
// Reference: Optimal::enumerate (optimal.h:48-75), enumerate_local (:79-105), Optimal_Rev (optimal_rev.h:44-115),
// Optimal_Subali::enumerate (optimal_subali.h:60-84).  A traceback is a chain of dependent loads (~1 us each
// from HBM); almost every step of a real alignment is the diagonal neighbour, so one wave per pair loads the 64
// cells (q -/+ l, t -/+ l), l = 0..63, of the current diagonal at once, finds with one ballot how far the
// stored pointers really follow that diagonal, emits that whole run, and only then takes the (rare) gap
// jump: one memory round trip per run of matches instead of one per aligned pair.
// Output: the pairs in TRAVERSAL order (forward builds: end -> start, the host getter flips it;
// reverse builds: start -> end, which is already list order).
#include "aln_device.h"

namespace aln {

struct TbParams {
  int islocal;
  int subali;       // start at the rectangle's far corner and stop at its origin (Optimal_Subali)
  int rev;          // pointers lead towards larger indices (reverse build, Optimal_Rev)
  int stride;       // capacity in pairs of each pair's output list
  int ptr_mode;     // pointer word encoding of the P plane
  int h_mode;       // score plane element type (aln_device.h load_score)
};

__global__ __launch_bounds__(64) void traceback_kernel(const PairDesc* __restrict__ pairs, const float* __restrict__ Hbase,
                                                       const uint32_t* __restrict__ Pbase, PairResult* __restrict__ res,
                                                       int32_t* __restrict__ out, TbParams prm) {
  const PairDesc pd = pairs[blockIdx.x];
  const int ld = pd.ld, lane = threadIdx.x;
  int32_t* o = out + (size_t)blockIdx.x * prm.stride * 2;
  PairResult r = res[blockIdx.x];
  if (r.status == ALN_E_HIP) return;            // the build of this batch failed (dp_corner.hip): nothing to trace, keep the status
  const int Q = pd.Q, T = pd.T;
  const int sg = prm.rev ? 1 : -1;              // direction the stored pointers lead in
  int n = 0, status = 0;
  auto emit1 = [&](int q, int t) {
    if (n < prm.stride) { if (lane == 0) { o[2 * n] = q; o[2 * n + 1] = t; } }
    else status = ALN_E_OVERFLOW;
    ++n;
  };
  int q, t, qstop, tstop;
  if (prm.subali) { q = pd.q1; t = pd.t1; qstop = pd.q0; tstop = pd.t0; }
  else if (prm.rev) { q = 0; t = 0; qstop = Q - 1; tstop = T - 1; }
  else { q = Q - 1; t = T - 1; qstop = 0; tstop = 0; }
  emit1(q, t);                                  // as[k].append(...)  optimal.h:63,:89 / optimal_rev.h:62,:93
  if (prm.islocal && !prm.subali) { q = r.best_q; t = r.best_t; emit1(q, t); }   // find_max result
  int lq = q, lt = t;                           // (q_last,t_last) / (q_first,t_first) after the loop
  auto before_stop = [&](int x) { return prm.rev ? (x < qstop) : (x > qstop); };
  while (before_stop(q)) {
    // lane l looks at cell (q + sg l, t + sg l)
    const int cq = q + sg * lane, ct = t + sg * lane;
    const bool valid = cq >= 0 && ct >= 0 && cq < Q && ct < T;
    uint32_t p = kNullPtr; float h = 0.f;
    if (valid) { p = load_ptr_word(Pbase, pd.plane_off, ld, cq, ct, prm.ptr_mode); h = load_score(Hbase, pd.plane_off, ld, cq, ct, prm.h_mode); }
    const float hnext = __shfl_down(h, 1);      // score of the diagonal neighbour (lane+1's cell)
    const bool active = valid && before_stop(cq);   // the while loop would process this cell
    const int nq_ = cq + sg, nt_ = ct + sg;
    int dq_, dt_;
    decode_ptr(p, prm.ptr_mode, cq, ct, dq_, dt_);
    const bool diag = active && p != kNullPtr && nq_ >= 0 && nt_ >= 0 && nq_ < Q && nt_ < T && dq_ == nq_ && dt_ == nt_;
    bool go = diag && lane < 63;                // lane 63 only supplies hnext for lane 62
    if (prm.islocal) go = go && !(hnext <= 0.f);   // the local loops break BEFORE adding a cell with score <= 0
    const unsigned long long m = __ballot(go);
    const int L = (~m == 0ull) ? 64 : __builtin_ctzll(~m);   // lanes 0..L-1 follow the diagonal
    if (lane < L) {
      int k = n + lane;
      if (k < prm.stride) { o[2 * k] = nq_; o[2 * k + 1] = nt_; }
    }
    if (n + L > prm.stride) status = ALN_E_OVERFLOW;
    n += L;
    // state at lane L (the first cell that does not simply continue); L <= 63 because lane 63 never goes
    const int sq = q + sg * L, st = t + sg * L;
    const uint32_t pL = (uint32_t)__shfl((int)p, L);
    const int jq = __shfl(dq_, L), jt = __shfl(dt_, L);
    const bool diagL = __shfl((int)diag, L) != 0;
    q = sq; t = st; lq = sq; lt = st;
    if (!before_stop(sq)) break;                // natural end of the while loop
    if (diagL && L == 63) continue;             // the run filled the window: reload from (sq,st)
    if (diagL) { lq = sq + sg; lt = st + sg; break; }   // local: neighbour's score <= 0 (optimal.h:98-100)
    // a gap jump (or an untouched cell)
    const int nq = jq, nt = jt;
    if (pL == kNullPtr) { lq = -1; lt = -1; if (!prm.islocal) status = ALN_E_STARTPAIR; break; }
    if (prm.islocal) {
      const float hn = load_score(Hbase, pd.plane_off, ld, nq, nt, prm.h_mode);
      if (hn <= 0.f) { lq = nq; lt = nt; break; }
    }
    emit1(nq, nt);
    q = nq; t = nt; lq = nq; lt = nt;
  }
  if (prm.islocal && !prm.subali) {
    if (lq != qstop && lt != tstop) emit1(qstop, tstop);        // optimal.h:104 / optimal_rev.h:114
  } else {
    if (lq != qstop || lt != tstop) status = ALN_E_STARTPAIR;   // optimal.h:74 / optimal_rev.h:74 / optimal_subali.h:82
  }
  if (lane == 0) {
    r.n_path = n < prm.stride ? n : prm.stride;
    r.status = status;
    res[blockIdx.x] = r;
  }
}

int launch_traceback(aln_batch* b, bool subali) {
  TbParams prm;
  prm.islocal = (b->islocal && !subali) ? 1 : 0;
  prm.subali = subali ? 1 : 0;
  prm.rev = (b->direction == ALN_REV && !subali) ? 1 : 0;
  prm.stride = b->path_stride;
  prm.ptr_mode = b->ptr_mode;
  prm.h_mode = b->h_mode;
  hipLaunchKernelGGL(traceback_kernel, dim3(b->n_pairs), dim3(64), 0, b->ctx->stream, b->d_pairs, b->d_H, b->d_P,
                     b->d_res, b->d_path, prm);
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

}  // namespace aln
