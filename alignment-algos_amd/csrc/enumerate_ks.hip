// enumerate_ks.hip — K-sorted constrained near-optimal enumeration on the resident DP planes (gfx950).
//
// Reference: KSConstrainedNearOptimal (kscw.h:109-351).  Like ConstrainedNearOptimal (enumerate.hip) it branches where the
// template's SuboptFlags bit flips, but a branch node first collects EVERY predecessor that passes Waterman's condition
// (match, row q0-1 for i = t0-2..1, column t0-1 for j = q0-2..1), sorts the operations by f + r - g (std::sort, or
// std::partial_sort when there are more than the node's limit), keeps the limit best, gives the best one the node's own limit
// and the others half of it, and lets an operation whose limit has fallen to 1 only follow stored pointers.
//
// kscw.h cannot be compiled on this platform (min(size_t, unsigned) at :188; debug operator<< for Troll-only types), so there
// is no golden for this kernel: it is checked against the oracle's restatement of the source (parity UNPINNED).
//
// Device form: one wave per pair, the trie / (head, score) representation and the diagonal-run walk of enumerate.hip.  The
// candidate scan is 64-wide with ballot compaction into LDS arrays (sum, candidate index) in the reference's order; the sort
// is libstdc++'s own algorithm (introsort + final insertion sort / heap select + sort_heap, GCC 11 bits/stl_algo.h,
// stl_heap.h) restated for one lane on those arrays, because the order of equal scores is observable in the result.
#include "enum_common.h"
#include "enum_sort.h"

namespace aln {


__global__ __launch_bounds__(64) void enumerate_ks_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto,
                                                          const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                          const float* __restrict__ tgi, const float* __restrict__ tge,
                                                          const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                                          const float* __restrict__ Sbase, EnumArgs a) {
  extern __shared__ float ks_lds[];
  float* csc = ks_lds;                                           // candidate sums
  int* cix = reinterpret_cast<int*>(ks_lds + a.cand_cap);        // candidate indices (position in the reference's scan order)
  {                                                              // batched launch (aln_batch_enumerate_all): one workgroup per pair,
    const size_t bi = blockIdx.x;                                // each with its own slice of the pools
    pair = a.pair_list ? a.pair_list[bi] : pair + (int)bi;
    a.node_pair += bi * a.node_cap; a.node_next += bi * a.node_cap;
    a.head += bi * a.ali_cap; a.score += bi * a.ali_cap; a.uid += bi * a.ali_cap;
    a.stack += bi * (size_t)a.stack_cap * (8 + 4 * a.k_limit);
    a.flags += (size_t)(a.pair_list ? pair : (int)bi) * (size_t)a.flags_stride;
    a.out += bi * 4;
  }
  const PairDesc pd = pairs[pair];
  EvalDev e = proto;
  e.Q = pd.Q; e.T = pd.T; e.ld = pd.ld;
  e.qc = qcodes ? qcodes + pd.q_off : nullptr;
  e.tc = tcodes ? tcodes + pd.t_off : nullptr;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  bind_table_model(e, proto, pd);
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  auto HV = [&](int i, int j) -> float { return load_score(Hbase, pd.plane_off, pd.ld, i, j, a.h_mode); };
  const int ld = pd.ld, lane = threadIdx.x;
  const int Q = pd.Q, T = pd.T;
  const int FW = 8 + 4 * (int)a.k_limit;                         // frame: q0 t0 k0 nops cursor curr_head curr_score - | ops (pq, pt, new_r, limit)

  uint32_t n_as = (uint32_t)a.first_slot + 1;                    // as.push_back(SingleAlignment())  kscw.h:120
  uint32_t n_nodes = 0;
  int status = 0;
  auto sync_mem = [&]() { __builtin_amdgcn_s_waitcnt(0); };
  if (lane == 0) { st_u(&a.head[a.first_slot], kNoNode); st_f(&a.score[a.first_slot], 0.f); a.uid[a.first_slot] = 1; }   // uid = 1, :121
  sync_mem();

  const float top = HV(Q - 1, T - 1);
  float thr = (1.f - a.delta_ratio) * top;                       // kscw.h:124-126
  { float alt = top - 0.1f; thr = (alt < thr) ? alt : thr; }

  auto prepend = [&](int k, int q, int t) {
    if (n_nodes >= a.node_cap) { status = ALN_E_OVERFLOW; return; }
    if (lane == 0) {
      a.node_pair[n_nodes] = ((uint32_t)q << 16) | (uint32_t)t;
      a.node_next[n_nodes] = ld_u(&a.head[k]);
      st_u(&a.head[k], n_nodes);
    }
    ++n_nodes;
    sync_mem();
  };
  auto base_case = [&](int q0, int t0, int k0) {                 // kscw.h:147-155 / :308-316
    prepend(k0, q0, t0);
    prepend(k0, 0, 0);
    float s = ld_f(&a.score[k0]);
    s += HV(q0, t0);
    if (lane == 0) st_f(&a.score[k0], s);
    sync_mem();
  };
  // the pointer-following loop of opt_path (kscw.h:329-348), 64 cells of a diagonal at a time (see enumerate.hip)
  auto walk = [&](int& q0, int& t0, int k0, bool force) {
    const bool flag = !a.flags[t0];
    float sc = ld_f(&a.score[k0]);
    uint32_t hd = ld_u(&a.head[k0]);
    while (t0 > 1 && q0 > 1 && status == 0) {
      const int q = q0 - lane, t = t0 - lane;
      bool stop = !(q > 1 && t > 1);
      if (!stop && !force && ((a.flags[t] != 0) == flag)) stop = true;
      int pq = 0, pt = 0; float sv = 0.f, g = 0.f;
      if (!stop) {
        const uint32_t p = load_ptr_word(Pbase, pd.plane_off, ld, q, t, a.ptr_mode);
        decode_ptr(p, a.ptr_mode, q, t, pq, pt);
        sv = dev_sim(e, q, t);
      }
      const bool diag = !stop && pq == q - 1 && pt == t - 1;
      const unsigned long long m_end = __ballot(!diag);
      const int F = m_end ? __builtin_ctzll(m_end) : 64;
      const bool gap_cell = F < 64 && !(((__ballot(stop)) >> F) & 1ull);
      const int n_proc = gap_cell ? F + 1 : F;
      if (n_proc == 0) break;
      if (n_nodes + (uint32_t)n_proc > a.node_cap) { status = ALN_E_OVERFLOW; break; }
      if (lane < n_proc) {
        a.node_pair[n_nodes + lane] = ((uint32_t)q << 16) | (uint32_t)t;
        a.node_next[n_nodes + lane] = lane == 0 ? hd : n_nodes + lane - 1;
      }
      hd = n_nodes + n_proc - 1;
      n_nodes += n_proc;
      if (gap_cell && lane == F) {
        if (q - pq == 1) g = dev_deletion(e, pt, t);
        else g = dev_insertion(e, pq, q, pt, t);
      }
      sc = add_in_path_order(sc, sv, n_proc);
      if (gap_cell) {
        sc -= __shfl(g, F);
        q0 = __shfl(pq, F); t0 = __shfl(pt, F);
      } else { q0 -= n_proc; t0 -= n_proc; }
    }
    if (lane == 0) { st_u(&a.head[k0], hd); st_f(&a.score[k0], sc); }
    sync_mem();
  };

  enum { CALL_NONE = 0, CALL_BRANCH = 1, CALL_OPT = 2 };
  int call = CALL_BRANCH, cq = Q - 1, ct = T - 1, ck = a.first_slot; bool cforce = false;
  uint32_t climit = a.k_limit;
  int sp = 0;
  long guard = 0;
  while ((call != CALL_NONE || sp > 0) && status == 0) {
    if (++guard > (1L << 36)) { status = ALN_E_OVERFLOW; break; }
    if (call == CALL_OPT) {                                      // opt_path, kscw.h:291-351
      call = CALL_NONE;
      int q0 = cq, t0 = ct; const int k0 = ck; bool force = cforce;
      if (climit <= 1) force = true;
      if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); continue; }
      walk(q0, t0, k0, force);
      call = CALL_BRANCH; cq = q0; ct = t0; ck = k0;             // branch(op(k_limit, pq, pt, k0)), :344-346
      continue;
    }
    if (call == CALL_BRANCH) {                                   // branch, kscw.h:139-288
      call = CALL_NONE;
      const int q0 = cq, t0 = ct, k0 = ck;
      const uint32_t k_limit = climit;
      if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); continue; }
      if (q0 < 1 || t0 < 1) { status = ALN_E_ARG; break; }       // the reference would index row / column -1
      if (n_as > a.user_limit) { call = CALL_OPT; cforce = true; continue; }   // :168-180 (same op)
      if ((uint32_t)sp >= a.stack_cap) { status = ALN_E_OVERFLOW; break; }
      const uint32_t ch = ld_u(&a.head[k0]);
      const float cs = ld_f(&a.score[k0]);
      const float r = cs + dev_sim(e, q0, t0);
      // ---- every candidate that passes Waterman's condition, in the reference's order, into LDS -----------------------
      const int ndel = t0 - 2, nins = q0 - 2;
      const int ncand = 1 + ndel + nins;
      int n = 0;
      // one 64-candidate group: the lanes that pass append (sum, candidate index) to the LDS arrays in lane order
      auto collect = [&](bool ok, float sum, int idx) {
        const unsigned long long m = __ballot(ok);
        if (m) {
          const int pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          if (ok && pos < (int)a.cand_cap) { csc[pos] = sum; cix[pos] = idx; }
          n += __builtin_popcountll(m);
        }
      };
      if (e.model == ALN_GAP_AFFINE_CONST && a.rowmax && Q <= 4096 && T <= 4096 && e.gi >= 0.f && e.ge >= 0.f) {
        // pruned scan (see enumerate_par.hip): only the 64-cell blocks of the candidate row / column whose maximum can pass are read
        const bool fdel = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
        const bool fins = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
        const float* rmax = a.rowmax + ((size_t)(pair - a.bm_pair0) * a.bm_rows + (size_t)(q0 - 1)) * a.nbt;
        const float* cmax = a.colmax + ((size_t)(pair - a.bm_pair0) * a.bm_cols + (size_t)(t0 - 1)) * a.nbq;
        bool pass_d = false, pass_i = false;
        {
          const int lo = lane * 64 > 1 ? lane * 64 : 1;
          int hi = lane * 64 + 63; hi = hi < t0 - 2 ? hi : t0 - 2;
          if (lo <= hi) {
            const int len = t0 - hi - 1;
            const float g = (len < 1 || (fdel && t0 == T - 1)) ? 0.f : e.gi + e.ge * (float)(len - 1);
            pass_d = (rmax[lane] + r) - g > thr;
          }
          int hq = lane * 64 + 63; hq = hq < q0 - 2 ? hq : q0 - 2;
          if (lo <= hq) {
            const int len = q0 - hq - 1;
            const float g = (len < 1 || (fins && q0 == Q - 1)) ? 0.f : e.gi + e.ge * (float)(len - 1);
            pass_i = (cmax[lane] + r) - g > thr;
          }
        }
        const unsigned long long md = __ballot(pass_d), mi = __ballot(pass_i);
        collect(lane == 0 && HV(q0 - 1, t0 - 1) + r > thr, HV(q0 - 1, t0 - 1) + r, 0);                     // the match
        for (unsigned long long mm = md; mm; ) {                                                           // deletions, t0-2 downwards
          const int blk = 63 - __builtin_clzll(mm); mm &= ~(1ull << blk);
          const int pt = blk * 64 + 63 - lane;
          const bool in = pt >= 1 && pt <= t0 - 2;
          const float sum = in ? HV(q0 - 1, pt) + r - dev_deletion(e, pt, t0) : 0.f;
          collect(in && sum > thr, sum, t0 - 1 - pt);
        }
        for (unsigned long long mm = mi; mm; ) {                                                           // insertions, q0-2 downwards
          const int blk = 63 - __builtin_clzll(mm); mm &= ~(1ull << blk);
          const int pq = blk * 64 + 63 - lane;
          const bool in = pq >= 1 && pq <= q0 - 2;
          const float sum = in ? HV(pq, t0 - 1) + r - dev_insertion(e, pq, q0, t0 - 1, t0) : 0.f;
          collect(in && sum > thr, sum, ndel + 1 + (q0 - 2 - pq));
        }
      } else
      for (int base = 0; base < ncand; base += 64) {
        const int idx = base + lane;
        bool ok = false; float sum = 0.f;
        if (idx < ncand) {
          if (idx == 0) { sum = HV(q0 - 1, t0 - 1) + r; }
          else if (idx <= ndel) { const int pt = t0 - 1 - idx; sum = HV(q0 - 1, pt) + r - dev_deletion(e, pt, t0); }
          else { const int pq = q0 - 2 - (idx - ndel - 1); sum = HV(pq, t0 - 1) + r - dev_insertion(e, pq, q0, t0 - 1, t0); }
          ok = sum > thr;
        }
        collect(ok, sum, idx);
      }
      if (n > (int)a.cand_cap) { status = ALN_E_OVERFLOW; break; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (n == 0) { call = CALL_OPT; climit = 1; cforce = true; continue; }    // :222-228: op(1, q0, t0, k0), forced
      // ---- sort, keep the k_limit best (kscw.h:232-241) ------------------------------------------------------------------
      int m_keep = n;
      if (lane == 0) {
        kssort::Arr arr = {csc, cix};
        if ((uint32_t)n > k_limit) kssort::partial_sort(arr, 0, (int)k_limit, n);
        else kssort::sort(arr, 0, n);
      }
      if ((uint32_t)n > k_limit) m_keep = (int)k_limit;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (m_keep == 0) { status = ALN_E_ARG; break; }            // limit 0 with candidates: `it->limit *= 2` on an empty vector in the reference
      // ---- the frame: operations in sorted order --------------------------------------------------------------------
      uint32_t* f = a.stack + (size_t)sp * FW;
      if (lane < m_keep) {
        const int idx = cix[lane];
        int pq, pt; float nr;
        if (idx == 0) { pq = q0 - 1; pt = t0 - 1; nr = r; }
        else if (idx <= ndel) { pq = q0 - 1; pt = t0 - 1 - idx; nr = r - dev_deletion(e, pt, t0); }
        else { pq = q0 - 2 - (idx - ndel - 1); pt = t0 - 1; nr = r - dev_insertion(e, pq, q0, pt, t0); }
        uint32_t lim = k_limit / 2;
        if (lane == 0) lim *= 2;                                 // only the best operation keeps (about) the node's limit, :246-247
        uint32_t* op = f + 8 + 4 * lane;
        st_u(op + 0, (uint32_t)pq); st_u(op + 1, (uint32_t)pt); st_u(op + 2, __float_as_uint(nr)); st_u(op + 3, lim);
      }
      if (lane == 0) {
        st_u(f + 0, (uint32_t)q0); st_u(f + 1, (uint32_t)t0); st_u(f + 2, (uint32_t)k0); st_u(f + 3, (uint32_t)m_keep);
        st_u(f + 4, 0u); st_u(f + 5, ch); st_u(f + 6, __float_as_uint(cs));
      }
      sync_mem();
      ++sp;
      continue;
    }
    // ---- resume the frame on top of the stack: the next sorted operation (kscw.h:250-262) ------------------------------
    uint32_t* f = a.stack + (size_t)(sp - 1) * FW;
    const int q0 = (int)ld_u(f + 0), t0 = (int)ld_u(f + 1), k0 = (int)ld_u(f + 2), nops = (int)ld_u(f + 3), cursor = (int)ld_u(f + 4);
    if (cursor >= nops) { --sp; continue; }
    const uint32_t curr_head = ld_u(f + 5);
    const float curr_score = __uint_as_float(ld_u(f + 6));
    const uint32_t* op = f + 8 + 4 * cursor;
    const int pq = (int)ld_u(op + 0), pt = (int)ld_u(op + 1);
    const float nr = __uint_as_float(ld_u(op + 2));
    const uint32_t lim = ld_u(op + 3);
    const int k = (cursor == 0) ? k0 : (int)n_as;
    if ((uint32_t)k == n_as) {                                   // as.push_back(curr); as[k].uid = k
      if (n_as >= a.ali_cap) { status = ALN_E_OVERFLOW; break; }
      if (lane == 0) { st_u(&a.head[n_as], curr_head); st_f(&a.score[n_as], curr_score); a.uid[n_as] = (int32_t)n_as; }
      sync_mem();
      ++n_as;
    }
    prepend(k, q0, t0);
    if (lane == 0) { st_f(&a.score[k], nr); st_u(f + 4, (uint32_t)(cursor + 1)); }
    sync_mem();
    call = CALL_OPT; cq = pq; ct = pt; ck = k; climit = lim; cforce = false;
  }
  if (lane == 0) { a.out[0] = (int32_t)n_as; a.out[1] = (int32_t)n_nodes; a.out[2] = status; }
}

}  // namespace aln
