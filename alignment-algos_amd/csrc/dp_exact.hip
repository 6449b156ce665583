// dp_exact.hip — exact-order O(n^3) DP for arbitrary fp32 similarities and gap functions (gfx950).
//
// Replaces the four builders of the reference, dpmatrix.h:356-1030, literally: every candidate is
//   s = D[pred]; s -= gap; s += S[i][j]; s = max(0,s) (local); if (s > opt_s) take it
// in the order match, deletions (k ascending), insertions (k ascending), so that non-integer gap
// penalties (the reference's defaults are 4.73 / 0.34, alib.cpp:17-18) and profile evaluators round
// exactly like the scalar C++.  fp contraction is off for the whole library.
//
// A reverse build (dpmatrix.h:691-1030) is the same programme in the mirrored frame a = q1 - i,
// b = t1 - j: "k descending from t1-1" there is "k' ascending from 1" here, so one kernel serves both
// directions and sub-rectangles (build_subdpm, :319-353); only the gap functions see real positions.
//
// Mapping: one workgroup per pair, a thread per column (strided), row by row.  The previous row lives
// in LDS (every thread walks it at the same k: broadcast reads); the insertion scan walks column b-1 of
// the score plane in global memory, coalesced across the threads of a wave.
#include "aln_device.h"

#include <cstdlib>

namespace aln {

constexpr int kExactThreads = 256;

__global__ __launch_bounds__(kExactThreads) void dp_exact_kernel(const PairDesc* __restrict__ pairs, EvalDev proto,
                                                                 const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                                 const float* __restrict__ tgi, const float* __restrict__ tge,
                                                                 float* __restrict__ Hbase, uint32_t* __restrict__ Pbase,
                                                                 const float* __restrict__ Sbase, PairResult* __restrict__ res,
                                                                 int islocal, int rev) {
  extern __shared__ __attribute__((aligned(16))) float lds_rows[];   // two rows of (nT+1) floats
  __shared__ float red_v[kExactThreads / 64];
  __shared__ uint32_t red_p[kExactThreads / 64];
  const PairDesc pd = pairs[blockIdx.x];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  bind_table_model(e, proto, pd);
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  float* __restrict__ H = Hbase + pd.plane_off;
  uint32_t* __restrict__ P = Pbase + pd.plane_off;
  const int ld = pd.ld;
  Frame f = {pd.q0, pd.q1, pd.t0, pd.t1, rev};
  const int nQ = f.nQ(), nT = f.nT();
  const bool local = islocal != 0;
  float* prev = lds_rows;
  float* cur = lds_rows + (nT + 1);
  float lmax = 0.f; uint32_t lpos = 0xFFFFFFFFu;
  const uint32_t origin = pack_ptr(f.rq(0), f.rt(0));

  if (nQ >= 2 && nT >= 2) {
    for (int a = 1; a <= nQ - 1; ++a) {
      const int i = f.rq(a);
      for (int b = 1 + (int)threadIdx.x; b <= nT - 1; b += kExactThreads) {
        const int j = f.rt(b);
        const float sim = dev_sim(e, i, j);
        float opt_s; uint32_t opt_p;
        if (a == 1 && b == 1) {                       // dpmatrix.h:409-410
          opt_s = clip0(0.f + sim, local); opt_p = origin;
        } else if (a == 1) {                          // :413-418
          float s = 0.f;
          s -= frame_del(e, f, 0, b);
          s += sim;
          opt_s = clip0(s, local); opt_p = origin;
        } else if (b == 1) {                          // :421-426
          float s = 0.f;
          s -= frame_ins(e, f, 0, a, 0, 1);
          s += sim;
          opt_s = clip0(s, local); opt_p = origin;
        } else {                                      // :447-486
          opt_s = clip0(prev[b - 1] + sim, local);
          int oa = a - 1, ob = b - 1;
          for (int k = 1; k < b - 1; ++k) {
            float s = prev[k];
            s -= frame_del(e, f, k, b);
            s += sim;
            s = clip0(s, local);
            if (s > opt_s) { oa = a - 1; ob = k; opt_s = s; }
          }
          const size_t colb = (size_t)f.rt(b - 1);
          for (int k = 1; k < a - 1; ++k) {
            // agent-scope load: served by L2, never by a stale L1 line that was filled before its row was finished
            float s = __hip_atomic_load(&H[(size_t)f.rq(k) * ld + colb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s -= frame_ins(e, f, k, a, b - 1, b);
            s += sim;
            s = clip0(s, local);
            if (s > opt_s) { oa = k; ob = b - 1; opt_s = s; }
          }
          opt_p = pack_ptr(f.rq(oa), f.rt(ob));
        }
        H[(size_t)i * ld + j] = opt_s;
        P[(size_t)i * ld + j] = opt_p;
        cur[b] = opt_s;
        if (opt_s > lmax) { lmax = opt_s; lpos = ((uint32_t)a << 16) | (uint32_t)b; }
      }
      __threadfence_block();
      __syncthreads();          // row a complete in LDS and visible in the plane for the column walks of row a+2
      float* tmp = prev; prev = cur; cur = tmp;
    }
  }
  // find_max partial in frame coordinates (first in frame row-major order among the maxima)
  float m = lmax; uint32_t p = lpos;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    float om = __shfl_xor(m, o); uint32_t op = (uint32_t)__shfl_xor((int)p, o);
    bool take = om > m || (om == m && op < p);
    m = take ? om : m; p = take ? op : p;
  }
  if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = m; red_p[threadIdx.x >> 6] = p; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kExactThreads / 64; ++w) {
      bool take = red_v[w] > m || (red_v[w] == m && red_p[w] < p);
      if (take) { m = red_v[w]; p = red_p[w]; }
    }
    uint32_t rp = 0xFFFFFFFFu;
    if (p != 0xFFFFFFFFu && m > 0.f) rp = ((uint32_t)f.rq((int)(p >> 16)) << 16) | (uint32_t)f.rt((int)(p & 0xFFFFu));
    res[blockIdx.x].part_max = m;
    res[blockIdx.x].part_pos = rp;
  }
}

int launch_dp_exact(aln_batch* b) {
  aln_ctx* ctx = b->ctx;
  // same results, restructured scans (dp_exact_blocked.hip); this literal kernel stays for templates beyond 4096 columns
  // and for the table gap model
  if (dp_exact_blocked_legal(b) && !ctx->hints.exact_literal) return launch_dp_exact_blocked(b);
  // untouched cells read score 0 / pointer (-1,-1) (dpmatrix.cpp:17-25): the kernel only writes computed cells
  ALN_HIP_CHECK(ctx, hipMemsetAsync(b->d_H, 0, (size_t)b->plane_elems * 4, ctx->stream));
  ALN_HIP_CHECK(ctx, hipMemsetAsync(b->d_P, 0xFF, (size_t)b->plane_elems * 4, ctx->stream));
  EvalDev proto = {};
  proto.model = b->gapdev.model;
  proto.align_type = b->gapdev.align_type;
  proto.gi = b->gapdev.gi; proto.ge = b->gapdev.ge;
  proto.sim_kind = (b->sim_kind == ALN_SIM_SUBMATRIX) ? ALN_SIM_SUBMATRIX : ALN_SIM_MATRIX;
  proto.tablef = b->d_tablef;
  proto.tcn = b->d_tcn; proto.deltab = b->d_deltab; proto.deltab_off = b->d_deltab_off; proto.instab = b->d_instab;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model != ALN_GAP_AFFINE_CONST;
  const size_t lds = (size_t)2 * (b->maxT + 1) * sizeof(float);
  hipLaunchKernelGGL(dp_exact_kernel, dim3(b->n_pairs), dim3(kExactThreads), lds, ctx->stream, b->d_pairs, proto,
                     sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr,
                     tpos ? b->d_tge : nullptr, b->d_H, b->d_P, sub ? nullptr : b->d_S, b->d_res, (int)b->islocal,
                     (int)(b->direction == ALN_REV));
  ALN_HIP_CHECK(ctx, hipGetLastError());
  b->kernel_name = std::string("dp_exact_kernel<") + (b->islocal ? "local" : "global") + (b->direction == ALN_REV ? ",rev>" : ",fwd>");
  return ALN_OK;
}

}  // namespace aln
