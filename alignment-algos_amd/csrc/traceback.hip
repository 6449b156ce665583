// traceback.hip — Optimal / Optimal_Subali pointer traceback on the resident planes (gfx950).
//
// Reference: Optimal::enumerate (optimal.h:48-75), enumerate_local (:79-105), Optimal_Subali::enumerate
// (optimal_subali.h:60-84).  A traceback is a chain of dependent loads (~1 us each from HBM); almost every
// step of a real alignment is the diagonal predecessor, so one wave per pair loads the 64 cells
// (q-l, t-l), l = 0..63, of the current diagonal at once, finds with one ballot how far the stored
// pointers really follow that diagonal, emits that whole run, and only then takes the (rare) gap jump:
// one memory round trip per run of matches instead of one per aligned pair.
// Output: the list in REVERSE order (end -> start); the host getter flips it.
#include "aln_device.h"

namespace aln {

struct TbParams {
  int islocal;
  int subali;       // start at (q1,t1) of the pair's rectangle and stop at (q0,t0) (Optimal_Subali)
  int stride;       // capacity in pairs of each pair's output list
};

__global__ __launch_bounds__(64) void traceback_kernel(const PairDesc* __restrict__ pairs, const float* __restrict__ Hbase,
                                                       const uint32_t* __restrict__ Pbase, PairResult* __restrict__ res,
                                                       int32_t* __restrict__ out, TbParams prm) {
  const PairDesc pd = pairs[blockIdx.x];
  const float* H = Hbase + pd.plane_off;
  const uint32_t* P = Pbase + pd.plane_off;
  const int ld = pd.ld, lane = threadIdx.x;
  int32_t* o = out + (size_t)blockIdx.x * prm.stride * 2;
  PairResult r = res[blockIdx.x];
  const int Q = pd.Q, T = pd.T;
  int n = 0, status = 0;
  auto emit1 = [&](int q, int t) {
    if (n < prm.stride) { if (lane == 0) { o[2 * n] = q; o[2 * n + 1] = t; } }
    else status = ALN_E_OVERFLOW;
    ++n;
  };
  int q, t, qstop, tstop;
  if (prm.subali) { q = pd.q1; t = pd.t1; qstop = pd.q0; tstop = pd.t0; }
  else { q = Q - 1; t = T - 1; qstop = 0; tstop = 0; }
  emit1(q, t);                                  // as[k].append(q_last,t_last)  optimal.h:63 / :89
  if (prm.islocal && !prm.subali) { q = r.best_q; t = r.best_t; emit1(q, t); }   // find_max result, :90-93
  int lq = q, lt = t;                           // (q_last,t_last) after the loop
  bool running = true;
  while (running && q > qstop) {
    // lane l looks at cell (q-l, t-l)
    const int cq = q - lane, ct = t - lane;
    const bool valid = cq >= 0 && ct >= 0;
    uint32_t p = kNullPtr; float h = 0.f;
    if (valid) { p = P[(size_t)cq * ld + ct]; h = H[(size_t)cq * ld + ct]; }
    const float hnext = __shfl_down(h, 1);      // score of the diagonal predecessor (lane+1's cell)
    const bool active = valid && cq > qstop;    // the while loop would process this cell
    const bool diag = active && ct >= 1 && p == (((uint32_t)(cq - 1) << 16) | (uint32_t)(ct - 1));
    bool go = diag && lane < 63;                // lane 63 only supplies hnext for lane 62
    if (prm.islocal) go = go && !(hnext <= 0.f);   // enumerate_local breaks BEFORE prepending a cell with score <= 0
    const unsigned long long m = __ballot(go);
    const int L = (~m == 0ull) ? 64 : __builtin_ctzll(~m);   // lanes 0..L-1 follow the diagonal
    // emit the run: lane l < L contributes its predecessor (cq-1, ct-1)
    if (lane < L) {
      int k = n + lane;
      if (k < prm.stride) { o[2 * k] = cq - 1; o[2 * k + 1] = ct - 1; }
    }
    if (n + L > prm.stride) status = ALN_E_OVERFLOW;
    n += L;
    // state at lane L (the first cell that does not simply continue); L <= 63 because lane 63 never goes
    const int sq = q - L, st = t - L;
    const uint32_t pL = (uint32_t)__shfl((int)p, L);
    const bool activeL = sq > qstop;            // sq,st >= 0 here: every earlier step was a stored diagonal pointer
    const bool diagL = __shfl((int)diag, L) != 0;
    q = sq; t = st; lq = sq; lt = st;
    if (!activeL) break;                        // natural end of the while loop (q_last == qstop)
    if (diagL && L == 63) continue;             // run filled the window: reload from (sq,st)
    if (diagL) {
      // local: predecessor's score <= 0 -> break with (q_last,t_last) = predecessor (optimal.h:98-100)
      lq = sq - 1; lt = st - 1; running = false;
      break;
    }
    // a gap jump (or an untouched cell)
    const int nq = (int)(pL >> 16), nt = (int)(pL & 0xFFFFu);
    if (pL == kNullPtr) { lq = -1; lt = -1; running = false; if (!prm.islocal) status = ALN_E_STARTPAIR; break; }
    if (prm.islocal) {
      const float hn = H[(size_t)nq * ld + nt];
      if (hn <= 0.f) { lq = nq; lt = nt; running = false; break; }
    }
    emit1(nq, nt);
    q = nq; t = nt; lq = nq; lt = nt;
  }
  if (prm.islocal && !prm.subali) {
    if (lq != 0 && lt != 0) emit1(0, 0);        // optimal.h:104
  } else {
    if (lq != qstop || lt != tstop) status = ALN_E_STARTPAIR;   // optimal.h:74 / optimal_subali.h:82
  }
  if (lane == 0) {
    r.n_path = n < prm.stride ? n : prm.stride;
    if (status == 0 && r.status != 0) status = r.status;
    r.status = status;
    res[blockIdx.x] = r;
  }
}

int launch_traceback(aln_batch* b, bool subali) {
  TbParams prm;
  prm.islocal = b->islocal ? 1 : 0;
  prm.subali = subali ? 1 : 0;
  prm.stride = b->path_stride;
  hipLaunchKernelGGL(traceback_kernel, dim3(b->n_pairs), dim3(64), 0, b->ctx->stream, b->d_pairs, b->d_H, b->d_P,
                     b->d_res, b->d_path, prm);
  ALN_HIP_CHECK(b->ctx, hipGetLastError());
  return ALN_OK;
}

}  // namespace aln
