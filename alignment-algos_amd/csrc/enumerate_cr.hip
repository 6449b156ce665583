// enumerate_cr.hip — "controlled redundancy" constrained near-optimal enumeration on the resident DP planes (gfx950).
//
// Reference: CRConstrainedNearOptimal (crcw.h:134-594), the enumerator the profile drivers default to (nalign2.cpp:114-130,
// gn2.cpp:139-185).  A branch node collects every predecessor that passes Waterman's condition (like kscw.h), sorts the
// operations by f + r - g and keeps the sort_limit best (std::sort / std::partial_sort, crcw.h:314-320), then
// filter_and_extend (crcw.h:345-550):
//   * every operation is followed along the stored pointers to the end of the template's current flag region; the visited
//     cells (template position -> query position), the sub-path's length, end cell, end region and running score are kept;
//   * in sorted order an operation is accepted unless an accepted, better one that ends in the same region shares more than
//     max_overlap * (its own length) cells with it (+1 if both end in the same cell); at most `limit` are accepted;
//   * each accepted operation becomes an alignment (the first continues the node's alignment, the others copy its state before
//     the node), extended by its whole sub-path; the best keeps the node's limit, the others get max(2, limit/2);
//   * sub-paths that end within two cells of the origin are finished along the stored pointers at once; the others recurse.
// An operation whose limit is below 2, and every operation once the set has user_limit alignments, only follows pointers.
//
// crcw.h cannot be compiled on this platform (min(size_t, unsigned) at :242; debug operator<< for Troll-only types), so there
// is no golden for this kernel: it is checked against the oracle's restatement of the source (parity UNPINNED).  One read of
// the source is out of bounds — `regions[t-1]` with t == 0 when a sub-path reaches the origin (crcw.h:387) — and yields, with
// glibc's allocator, a number no real region has; here (and in the oracle) such sub-paths end in a region of their own.
//
// Device form: one wave per pair; trie / (head, score) alignments and the explicit stack of enumerate_ks.hip.  Candidates go
// to LDS in the reference's order, lane 0 runs libstdc++'s sort on them (enum_sort.h).  Sub-paths are walked one per lane (64
// at a time) into a [sort_limit][T] table of 16-bit query positions in global scratch; the filter compares operation i with all
// earlier operations in parallel (lane j: overlap with operation j); extending an alignment by a sub-path writes its trie
// nodes 64 template positions at a time.
#include "enum_common.h"
#include "enum_sort.h"

namespace aln {

namespace {
constexpr int kOriginRegion = -1;       // the region of a sub-path that ended at template position 0 (crcw.h:387 reads regions[-1])
constexpr uint16_t kNoQ = 0xFFFFu;      // alignments[i][k] == -1
}  // namespace

__global__ __launch_bounds__(64) void enumerate_cr_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto,
                                                          const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                          const float* __restrict__ tgi, const float* __restrict__ tge,
                                                          const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                                          const float* __restrict__ Sbase, EnumArgs a) {
  extern __shared__ float cr_lds[];
  const int SL = (int)a.sort_limit;
  float* csc = cr_lds;                                           // candidate sums
  int* cix = reinterpret_cast<int*>(cr_lds + a.cand_cap);        // candidate indices (position in the reference's scan order)
  int* o_q = cix + a.cand_cap;                                   // per sorted operation: first cell of its sub-path ...
  int* o_t = o_q + SL;
  int* o_rq = o_t + SL;                                          // ... the cell it stopped at (p_rq, p_rt) ...
  int* o_rt = o_rq + SL;
  int* o_len = o_rt + SL;                                        // ... l_sp, the end region, the accumulated reverse score
  int* o_state = o_len + SL;
  float* o_rs = reinterpret_cast<float*>(o_state + SL);
  int* o_keep = reinterpret_cast<int*>(o_rs + SL);               // filter[]
  {
    const size_t bi = blockIdx.x;
    pair = a.pair_list ? a.pair_list[bi] : pair + (int)bi;
    a.node_pair += bi * a.node_cap; a.node_next += bi * a.node_cap;
    a.head += bi * a.ali_cap; a.score += bi * a.ali_cap; a.uid += bi * a.ali_cap;
    a.stack += bi * (size_t)a.stack_cap * (8 + 4 * a.k_limit);
    a.flags += (size_t)(a.pair_list ? pair : (int)bi) * (size_t)a.flags_stride;
    a.cr_ali += bi * (size_t)SL * a.cr_tpad;
    a.cr_reg += bi * (size_t)a.cr_tpad;
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
  const int FW = 8 + 4 * (int)a.k_limit;                         // frame: - - - nops cursor - - - | ops (q0, t0, k0 or -1, limit)
  const int tpad = a.cr_tpad;

  uint32_t n_as = (uint32_t)a.first_slot + 1;                    // as.push_back(SingleAlignment())  crcw.h:146
  uint32_t n_nodes = 0;
  int status = 0;
  auto sync_mem = [&]() { __builtin_amdgcn_s_waitcnt(0); };
  auto lds_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
  if (lane == 0) { st_u(&a.head[a.first_slot], kNoNode); st_f(&a.score[a.first_slot], 0.f); a.uid[a.first_slot] = 1; }   // uid = 1, :147

  // init_mem (crcw.h:174-179): regions[i] = number of positions m <= i with flags[m+1] != flags[m]; reg[t] = regions[t-1], t >= 1
  {
    int carry = 0;
    for (int base = 0; base < T - 1; base += 64) {
      const int m = base + lane;
      const bool flip = m < T - 1 && ((a.flags[m + 1] != 0) != (a.flags[m] != 0));
      const unsigned long long bm = __ballot(flip);
      const int incl = carry + __builtin_popcountll(bm & ((2ull << lane) - 1ull));
      if (m < T - 1) a.cr_reg[m + 1] = incl;
      carry += __builtin_popcountll(bm);
    }
    if (lane == 0) a.cr_reg[0] = kOriginRegion;
  }
  sync_mem();
  auto ld16 = [&](const uint16_t* p) -> uint16_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto region_at = [&](int t) -> int { return ld_u(reinterpret_cast<const uint32_t*>(&a.cr_reg[t])); };   // t >= 0; t == 0 -> kOriginRegion

  const float top = HV(Q - 1, T - 1);
  float thr = (1.f - a.delta_ratio) * top;                       // crcw.h:151-152
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
  // force_opt_path (crcw.h:552-592): follow the stored pointers while q0 > 0 and t0 > 0 — 64 cells of a diagonal at a time, a
  // run of match pointers in one step (see enumerate.hip) — then prepend (0,0)
  auto force_opt_path = [&](int q0, int t0, int k0) {
    float sc = ld_f(&a.score[k0]);
    uint32_t hd = ld_u(&a.head[k0]);
    while (t0 > 0 && q0 > 0 && status == 0) {
      const int q = q0 - lane, t = t0 - lane;
      const bool stop = !(q > 0 && t > 0);
      int pq = 0, pt = 0; float sv = 0.f, g = 0.f;
      if (!stop) {
        const uint32_t p = load_ptr_word(Pbase, pd.plane_off, ld, q, t, a.ptr_mode);
        decode_ptr(p, a.ptr_mode, q, t, pq, pt);
        sv = dev_sim(e, q, t);
      }
      const bool diag = !stop && pq == q - 1 && pt == t - 1 && q > 1 && t > 1;   // (1,1) -> (0,0) is handled as a jump: it ends the loop
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
      sc = add_in_path_order(sc, sv, n_proc);      // score += sim, then -= 0 for a match step, in path order
      if (gap_cell) {
        sc -= __shfl(g, F);
        q0 = __shfl(pq, F); t0 = __shfl(pt, F);
      } else { q0 -= n_proc; t0 -= n_proc; }
    }
    if (lane == 0) { st_u(&a.head[k0], hd); st_f(&a.score[k0], sc); }
    sync_mem();
    prepend(k0, 0, 0);
  };

  enum { CALL_NONE = 0, CALL_BRANCH = 1 };
  int call = CALL_BRANCH, cq = Q - 1, ct = T - 1, ck = a.first_slot;
  uint32_t climit = a.k_limit;
  int sp = 0;
  long guard = 0;
  sync_mem();
  while ((call != CALL_NONE || sp > 0) && status == 0) {
    if (++guard > (1L << 36)) { status = ALN_E_OVERFLOW; break; }
    if (call == CALL_BRANCH) {                                   // branch, crcw.h:205-338
      call = CALL_NONE;
      const int q0 = cq, t0 = ct, k0 = ck;
      const uint32_t k_limit = climit;
      if (k_limit < 2) { force_opt_path(q0, t0, k0); continue; }                    // :219
      if (n_as > a.user_limit) { force_opt_path(q0, t0, k0); continue; }            // :224-236
      if (q0 < 1 || t0 < 1) { status = ALN_E_ARG; break; }       // the reference would index row / column -1
      if ((uint32_t)sp >= a.stack_cap) { status = ALN_E_OVERFLOW; break; }
      const uint32_t curr_head = ld_u(&a.head[k0]);
      const float curr_score = ld_f(&a.score[k0]);
      const float r = curr_score + dev_sim(e, q0, t0);
      // ---- every candidate that passes Waterman's condition, in the reference's order, into LDS (crcw.h:268-299) --------
      const int ndel = t0 - 2, nins = q0 - 2;
      const int ncand = 1 + ndel + nins;
      int n = 0;
      for (int base = 0; base < ncand; base += 64) {
        const int idx = base + lane;
        bool ok = false; float sum = 0.f;
        if (idx < ncand) {
          if (idx == 0) { sum = HV(q0 - 1, t0 - 1) + r; }
          else if (idx <= ndel) { const int pt = t0 - 1 - idx; sum = HV(q0 - 1, pt) + r - dev_deletion(e, pt, t0); }
          else { const int pq = q0 - 2 - (idx - ndel - 1); sum = HV(pq, t0 - 1) + r - dev_insertion(e, pq, q0, t0 - 1, t0); }
          ok = sum > thr;
        }
        const unsigned long long m = __ballot(ok);
        if (m) {
          const int pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          if (ok && pos < (int)a.cand_cap) { csc[pos] = sum; cix[pos] = idx; }
          n += __builtin_popcountll(m);
        }
      }
      if (n > (int)a.cand_cap) { status = ALN_E_OVERFLOW; break; }
      lds_sync();
      if (n == 0) { force_opt_path(q0, t0, k0); continue; }      // :303-310
      if (lane == 0) {                                           // :314-320
        kssort::Arr arr = {csc, cix};
        if (n > SL) kssort::partial_sort(arr, 0, SL, n);
        else kssort::sort(arr, 0, n);
      }
      if (n > SL) n = SL;
      lds_sync();
      // ---- filter_and_extend (crcw.h:345-550).  (1) follow every operation to the end of its flag region: one lane each ----
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) {
          const int idx = cix[i];
          int q, t; float rs;
          if (idx == 0) { q = q0 - 1; t = t0 - 1; rs = r; }
          else if (idx <= ndel) { q = q0 - 1; t = t0 - 1 - idx; rs = r - dev_deletion(e, t, t0); }
          else { q = q0 - 2 - (idx - ndel - 1); t = t0 - 1; rs = r - dev_insertion(e, q, q0, t, t0); }
          o_q[i] = q; o_t[i] = t;
          uint16_t* row = a.cr_ali + (size_t)i * tpad;
          const int st = region_at(t);                           // t >= 1
          int len = 1;
          while (q > 0 && t > 0 && region_at(t) == st) {
            row[t - 1] = (uint16_t)q;
            ++len;
            int pq, pt;
            decode_ptr(load_ptr_word(Pbase, pd.plane_off, ld, q, t, a.ptr_mode), a.ptr_mode, q, t, pq, pt);
            float g;
            if (q - pq == 1) g = dev_deletion(e, pt, t);
            else g = dev_insertion(e, pq, q, pt, t);
            rs += dev_sim(e, q, t);
            rs -= g;
            for (int u = pt + 1; u < t; ++u) row[u - 1] = kNoQ;  // template positions a deletion jumps over hold -1 (reinit_mem)
            q = pq; t = pt;
          }
          o_rq[i] = q; o_rt[i] = t; o_len[i] = len; o_rs[i] = rs;
          o_state[i] = region_at(t < 0 ? 0 : t);                 // crcw.h:387; t == 0 -> the origin's own region
          o_keep[i] = 0;
        }
      }
      sync_mem();
      lds_sync();
      // (2) the redundancy filter (crcw.h:400-441): operation i against every accepted earlier one, lane j <-> operation j
      const uint32_t lim = k_limit;                              // v_op.back().limit: every operation was created with the node's limit
      if (lane == 0) o_keep[0] = 1;
      lds_sync();
      uint32_t accepted = 1;
      for (int i = 1; i < n && accepted < lim; ++i) {
        bool rejected = false;
        const int ti = o_t[i], rti = o_rt[i], rqi = o_rq[i], sti = o_state[i];
        const uint16_t* ri = a.cr_ali + (size_t)i * tpad;
        for (int base = 0; base < i; base += 64) {
          const int j = base + lane;
          bool rej = false;
          if (j < i && o_keep[j] && o_state[j] == sti) {
            float overlap = 0.f;
            const float overlap_max = a.max_overlap * (float)o_len[j];
            if (rqi == o_rq[j] && rti == o_rt[j]) ++overlap;
            // cells of both sub-paths exist only on template positions (p_rt, t_start]; everything else reads -1
            const int hi = (ti < o_t[j] ? ti : o_t[j]), lo = (rti > o_rt[j] ? rti : o_rt[j]);
            const uint16_t* rj = a.cr_ali + (size_t)j * tpad;
            for (int k = hi - 1; k >= lo; --k) {
              const uint16_t x = ld16(ri + k);
              if (x != kNoQ && x == ld16(rj + k)) ++overlap;
            }
            rej = overlap > overlap_max;
          }
          if (__ballot(rej)) rejected = true;
        }
        if (!rejected) { if (lane == 0) o_keep[i] = 1; ++accepted; }
        lds_sync();
      }
      // (3) the accepted operations in order (at most lim) -> the frame; (4) their alignments (crcw.h:470-531)
      uint32_t* f = a.stack + (size_t)sp * FW;
      int nops = 0;
      int k = k0;
      for (int i = 0; i < n && (uint32_t)nops < lim; ++i) {
        if (!o_keep[i]) continue;
        if ((uint32_t)k == n_as) {                               // as.push_back(curr); as[k].uid = k
          if (n_as >= a.ali_cap) { status = ALN_E_OVERFLOW; break; }
          if (lane == 0) { st_u(&a.head[n_as], curr_head); st_f(&a.score[n_as], curr_score); a.uid[n_as] = (int32_t)n_as; }
          sync_mem();
          ++n_as;
        }
        prepend(k, q0, t0);
        // as[k].prepend(alignments[i][j-1], j) for j = t0-1 .. p_rt+1 where set: 64 template positions per step
        {
          const uint16_t* row = a.cr_ali + (size_t)i * tpad;
          uint32_t hd = ld_u(&a.head[k]);
          for (int jb = o_t[i]; jb > o_rt[i] && status == 0; jb -= 64) {
            const int j = jb - lane;
            const uint16_t aq = j > o_rt[i] ? ld16(row + j - 1) : kNoQ;
            const bool have = aq != kNoQ;
            const unsigned long long m = __ballot(have);
            const int cnt = __builtin_popcountll(m);
            if (cnt == 0) continue;
            if (n_nodes + (uint32_t)cnt > a.node_cap) { status = ALN_E_OVERFLOW; break; }
            const int pos = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (have) {
              a.node_pair[n_nodes + pos] = ((uint32_t)aq << 16) | (uint32_t)j;
              a.node_next[n_nodes + pos] = pos == 0 ? hd : n_nodes + pos - 1;
            }
            hd = n_nodes + cnt - 1;
            n_nodes += cnt;
          }
          if (lane == 0) { st_u(&a.head[k], hd); st_f(&a.score[k], o_rs[i]); }
          sync_mem();
        }
        if (status) break;
        const int nq = o_rq[i], nt = o_rt[i];
        int kk = k;
        if (nq <= 2 || nt <= 2) { force_opt_path(nq, nt, k); kk = -1; }             // end_alignment = 2, :514-517
        if (lane == 0) {
          uint32_t* op = f + 8 + 4 * nops;
          st_u(op + 0, (uint32_t)nq); st_u(op + 1, (uint32_t)nt); st_u(op + 2, (uint32_t)kk);
          st_u(op + 3, nops == 0 ? lim : (lim / 2 > 2u ? lim / 2 : 2u));            // :466-468
        }
        ++nops;
        k = (int)n_as;
      }
      if (status) break;
      if (lane == 0) { st_u(f + 3, (uint32_t)nops); st_u(f + 4, 0u); }
      sync_mem();
      ++sp;
      continue;
    }
    // ---- resume the frame on top of the stack: branch() on the next operation that is still open (crcw.h:330-336) ----------
    uint32_t* f = a.stack + (size_t)(sp - 1) * FW;
    const int nops = (int)ld_u(f + 3), cursor = (int)ld_u(f + 4);
    if (cursor >= nops) { --sp; continue; }
    const uint32_t* op = f + 8 + 4 * cursor;
    const int nq = (int)ld_u(op + 0), nt = (int)ld_u(op + 1), kk = (int)ld_u(op + 2);
    const uint32_t lim = ld_u(op + 3);
    if (lane == 0) st_u(f + 4, (uint32_t)(cursor + 1));
    sync_mem();
    if (kk < 0) continue;
    call = CALL_BRANCH; cq = nq; ct = nt; ck = kk; climit = lim;
  }
  if (lane == 0) { a.out[0] = (int32_t)n_as; a.out[1] = (int32_t)n_nodes; a.out[2] = status; }
}

}  // namespace aln
