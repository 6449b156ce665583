// enumerate_par.hip — ConstrainedNearOptimal / UnconstrainedNearOptimal (cw.h:68-284, ucw.h:64-236) searched by SEVERAL waves
// per pair (gfx950).
//
// enumerate.hip walks a pair's depth-first search with one wave: every step of it is a chain of dependent memory round trips
// (frame -> candidate scores -> pool words -> pointer word -> ...), so 1024 pairs keep 1024 SIMDs busy ~1 % of the time.  The
// search tree itself is wide: every candidate a branch node accepts starts an independent sub-search (its own alignment, its
// own score; the only shared state of the reference's recursion is the ORDER in which alignments are appended to the set).
// This kernel therefore
//   * turns every sub-search into a task (cell, alignment slot, trie head, score so far) on a per-pair stack in HBM,
//   * lets the W waves of the pair's workgroup pop W tasks per round (three barriers a round),
//   * scans a branch node ONCE, taking all accepted candidates of a 64-candidate group in parallel (the serial kernel resumes
//     the scan after each accepted candidate's subtree), and
//   * records for every new slot where the reference would have created it: (slot the branch node belongs to, template
//     index t0 of the node, candidate index).  Along one alignment t0 strictly decreases from node to node, the recursion
//     returns from the deepest node first, and a slot's whole subtree is appended before its next sibling — so the reference's
//     set order is the pre-order of the slot tree with siblings sorted by (t0 ascending, candidate index ascending).  The host
//     renumbers the slots that way (enumerate.hip: slot_order) before the reference's own sort calls see them.
// The one thing that cannot be decided locally is "as.size() > user_limit" (cw.h:127 / ucw.h:110), which depends on how many
// alignments exist at that moment of the serial order: if a pair's set outgrows user_limit, the kernel reports kParSerial and
// the host repeats that pair with the one-wave kernel.
// Sequence codes, the substitution table and the SuboptFlags row live in LDS; trie nodes, slots and tasks are handed out with
// LDS atomics.  Results are bit-identical to enumerate_kernel's (tests run both against the reference's sets).
#include "enum_common.h"

namespace aln {

__global__ __launch_bounds__(1024) void enumerate_par_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto,
                                                            const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                            const float* __restrict__ tgi, const float* __restrict__ tge,
                                                            const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                                            const float* __restrict__ Sbase, EnumArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_top, s_status;
  __shared__ unsigned s_nodes, s_slots;
  {
    const size_t bi = blockIdx.x;
    pair = a.pair_list ? a.pair_list[bi] : pair + (int)bi;
    a.node_pair += bi * a.node_cap; a.node_next += bi * a.node_cap;
    a.head += bi * a.ali_cap; a.score += bi * a.ali_cap;
    a.task += bi * (size_t)a.ali_cap * kTaskWords;
    a.slot_info += bi * (size_t)a.ali_cap * 3;
    a.flags += (size_t)(a.pair_list ? pair : (int)bi) * (size_t)a.flags_stride;
    a.out += bi * 4;
  }
  const PairDesc pd = pairs[pair];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const bool cw = a.kind == ALN_ENUM_CW;

  // ---- per-pair constants into LDS: flags | template codes | query codes | 32 x 32 table -----------------------------------
  const int Tp = (T + 15) & ~15, Qp = (Q + 15) & ~15;
  uint8_t* l_fl = lds_raw;
  uint8_t* l_tc = l_fl + Tp;
  uint8_t* l_qc = l_tc + Tp;
  float* l_tab = reinterpret_cast<float*>(l_qc + Qp);
  for (int i = threadIdx.x; i < T; i += blockDim.x) l_fl[i] = a.flags[i];
  const bool sub = proto.sim_kind == ALN_SIM_SUBMATRIX;
  if (sub) {
    for (int i = threadIdx.x; i < T; i += blockDim.x) l_tc[i] = tcodes[pd.t_off + i];
    for (int i = threadIdx.x; i < Q; i += blockDim.x) l_qc[i] = qcodes[pd.q_off + i];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) l_tab[i] = proto.tablef[i];
  }
  EvalDev e = proto;
  e.Q = Q; e.T = T; e.ld = ld;
  e.qc = sub ? l_qc : nullptr;
  e.tc = sub ? l_tc : nullptr;
  e.tablef = sub ? l_tab : proto.tablef;
  e.tgi = tgi ? tgi + pd.t_off : nullptr;
  e.tge = tge ? tge + pd.t_off : nullptr;
  bind_table_model(e, proto, pd);
  e.S = Sbase ? Sbase + pd.plane_off : nullptr;
  auto HV = [&](int i, int j) -> float { return load_score(Hbase, pd.plane_off, ld, i, j, a.h_mode); };

  const float top = HV(Q - 1, T - 1);
  float thr = (1.f - a.delta_ratio) * top;       // cw.h:86-88
  { float alt = top - 0.1f; thr = (alt < thr) ? alt : thr; }

  if (threadIdx.x == 0) {
    s_top = 1; s_status = 0; s_nodes = 0u; s_slots = (unsigned)a.first_slot + 1u;     // as.push_back(SingleAlignment())  cw.h:82 / ucw.h:78
    uint32_t* tk = a.task;                                                            // branch(final cell, seed slot)  cw.h:92 / ucw.h:86
    st_u(tk + 0, ((uint32_t)(Q - 1) << 16) | (uint32_t)(T - 1)); st_u(tk + 1, (uint32_t)a.first_slot); st_u(tk + 2, kNoNode);
    st_u(tk + 3, __float_as_uint(0.f)); st_u(tk + 4, 1u);
    st_u(&a.head[a.first_slot], kNoNode); st_f(&a.score[a.first_slot], 0.f);
  }

  auto fail = [&](int code) { if (lane == 0) atomicCAS(&s_status, 0, code); };
  // n consecutive trie nodes for this wave; kNoNode: the pool is exhausted (status set)
  auto alloc_nodes = [&](int n) -> uint32_t {
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(&s_nodes, (unsigned)n);
    b = (uint32_t)__shfl((int)b, 0);
    if (b + (uint32_t)n > a.node_cap) { fail(ALN_E_OVERFLOW); return kNoNode; }
    return b;
  };

  for (;;) {
    __threadfence();
    __syncthreads();                                   // (A) the previous round's tasks, counters and status are complete
    const int tp = s_top, st = s_status;
    const int ntake = tp < W ? tp : W;
    uint32_t tw0 = 0, tw1 = 0, tw2 = 0, tw3 = 0, tw4 = 0;
    if (st == 0 && w < ntake) {                        // read the task before anybody may push over it
      const uint32_t* tk = a.task + (size_t)(tp - 1 - w) * kTaskWords;
      const uint32_t v = lane < 5 ? ld_u(tk + lane) : 0u;
      tw0 = (uint32_t)__shfl((int)v, 0); tw1 = (uint32_t)__shfl((int)v, 1); tw2 = (uint32_t)__shfl((int)v, 2);
      tw3 = (uint32_t)__shfl((int)v, 3); tw4 = (uint32_t)__shfl((int)v, 4);
    }
    __syncthreads();                                   // (B) every wave has read tp, st and its task
    if (st != 0 || tp == 0) break;
    if (threadIdx.x == 0) s_top = tp - ntake;
    __syncthreads();                                   // (C) pushes start above the remaining tasks
    if (w >= ntake) continue;

    // ---- one task: opt_path / branch of ONE alignment slot until its branch node has spawned its children ---------------
    int q0 = (int)(tw0 >> 16), t0 = (int)(tw0 & 0xFFFFu);
    const uint32_t slot = tw1;
    uint32_t hd = tw2;
    float sc = __uint_as_float(tw3);
    bool is_branch = (tw4 & 1u) != 0, force = (tw4 & 2u) != 0;
    bool dead = false;

    // as[slot].prepend(q0,t0); as[slot].prepend(0,0); score += H(q0,t0)   (cw.h:100-108 / ucw.h:93-101), then publish the slot
    auto base_case = [&]() {
      const uint32_t b = alloc_nodes(2);
      if (b == kNoNode) return;
      if (lane == 0) {
        a.node_pair[b] = ((uint32_t)q0 << 16) | (uint32_t)t0; a.node_next[b] = hd;
        a.node_pair[b + 1] = 0u; a.node_next[b + 1] = b;
        a.head[slot] = b + 1;
        a.score[slot] = sc + HV(q0, t0);
      }
    };

    int guard = 0;
    while (!dead) {
      if (++guard > 2 * (Q + T) + 64) { fail(ALN_E_OVERFLOW); break; }       // (a path that never reaches row / column 1: not a DP matrix)
      if (q0 == 1 || t0 == 1) { base_case(); break; }                      // cw.h:220-228, :113-121
      if (!is_branch) {
        // opt_path: follow stored pointers, 64 diagonal cells at a time (see enumerate.hip walk())
        const bool flag = cw ? !l_fl[t0] : false;
        while (t0 > 1 && q0 > 1) {
          const int q = q0 - lane, t = t0 - lane;
          bool stop = !(q > 1 && t > 1);
          if (!stop && cw && !force && ((l_fl[t] != 0) == flag)) stop = true;
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
          const uint32_t b = alloc_nodes(n_proc);
          if (b == kNoNode) { dead = true; break; }
          if (lane < n_proc) {
            a.node_pair[b + lane] = ((uint32_t)q << 16) | (uint32_t)t;
            a.node_next[b + lane] = lane == 0 ? hd : b + lane - 1;
          }
          hd = b + n_proc - 1;
          if (gap_cell && lane == F) {
            if (q - pq == 1) g = dev_deletion(e, pt, t);
            else g = dev_insertion(e, pq, q, pt, t);
          }
          for (int l = 0; l < n_proc; ++l) sc += __shfl(sv, l);            // path order: fp32 is not associative
          if (gap_cell) {
            sc -= __shfl(g, F);
            q0 = __shfl(pq, F); t0 = __shfl(pt, F);
          } else { q0 -= n_proc; t0 -= n_proc; }
        }
        if (dead) break;
        if (!cw) { base_case(); break; }                                    // ucw.h:232-234
        is_branch = true;                                                   // branch(pq,pt,k0,force)  cw.h:276
        continue;
      }
      // ---- branch(q0,t0,slot) ----
      if (cw && force) { is_branch = false; continue; }                     // cw.h:205-209: opt_path(..., true)
      const float r = sc + dev_sim(e, q0, t0);
      const int ndel = t0 - 2, nins = q0 - 2;
      const int ncand = 1 + ndel + nins;
      bool first = true;                                                    // the first accepted candidate continues in `slot`
      // every accepted candidate of one 64-candidate group: a trie node (q0,t0) in front of hd, a slot (new ones are recorded
      // with their place in the reference's order) and a task
      auto spawn = [&](unsigned long long m, float g, int cq, int ct, int idx) {
        const int n = __popcll(m);
        const int nnew = n - (first ? 1 : 0);
        uint32_t bt = 0, bn = 0, bs = 0;
        if (lane == 0) {
          bt = (uint32_t)atomicAdd(&s_top, n);
          bn = atomicAdd(&s_nodes, (unsigned)n);
          if (nnew) bs = atomicAdd(&s_slots, (unsigned)nnew);
        }
        bt = (uint32_t)__shfl((int)bt, 0); bn = (uint32_t)__shfl((int)bn, 0); bs = (uint32_t)__shfl((int)bs, 0);
        if (nnew && bs + (uint32_t)nnew > a.user_limit) { fail(kParSerial); dead = true; return; }   // the serial order decides what user_limit cuts
        if (bt + (uint32_t)n > a.ali_cap || bn + (uint32_t)n > a.node_cap || bs + (uint32_t)nnew > a.ali_cap) {
          fail(ALN_E_OVERFLOW); dead = true; return;
        }
        if ((m >> lane) & 1ull) {
          const int rnk = __popcll(m & ((1ull << lane) - 1ull));
          const uint32_t nd = bn + (uint32_t)rnk;
          a.node_pair[nd] = ((uint32_t)q0 << 16) | (uint32_t)t0;
          a.node_next[nd] = hd;
          const int srank = rnk - (first ? 1 : 0);
          uint32_t sl = slot;
          if (srank >= 0) {
            sl = bs + (uint32_t)srank;
            uint32_t* si = a.slot_info + (size_t)sl * 3;
            si[0] = slot; si[1] = (uint32_t)t0; si[2] = (uint32_t)idx;
          }
          uint32_t* tk = a.task + (size_t)(bt + (uint32_t)rnk) * kTaskWords;
          st_u(tk + 0, ((uint32_t)cq << 16) | (uint32_t)ct); st_u(tk + 1, sl); st_u(tk + 2, nd);
          st_u(tk + 3, __float_as_uint(r - g)); st_u(tk + 4, cw ? 0u : 1u);   // cw: opt_path(cand, k, false); ucw: branch(cand, k)
        }
        first = false;
      };
      if (e.model == ALN_GAP_AFFINE_CONST) {
        // as in enumerate.hip: 16 x 64 score loads in flight per trip; the accept masks of a trip are parked one per lane
        constexpr int kTrip = 16;
        const bool fdel = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
        const bool fins = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
        const uint16_t* H16p = reinterpret_cast<const uint16_t*>(Hbase) + pd.plane_off;
        const float* H32p = Hbase + pd.plane_off;
        auto cand = [&](int idx, int& q, int& t, float& g) {
          const bool isdel = idx <= ndel;                       // idx 0 (match) has the same row
          q = isdel ? q0 - 1 : q0 - 2 - (idx - ndel - 1);
          t = idx == 0 ? t0 - 1 : isdel ? t0 - 1 - idx : t0 - 1;
          const bool in = idx < ncand;
          q = in ? q : q0 - 1; t = in ? t : t0 - 1;            // lanes past the end read a harmless cell
          g = 0.f;
          if (idx != 0) {
            if (isdel) {                                        // aasubalib.h:27-51
              const int len = t0 - t - 1;
              g = (len < 1 || (fdel && (t == 0 || t0 == T - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
            } else {                                            // aasubalib.h:53-77
              const int len = q0 - q - 1;
              g = (len < 1 || (fins && (q == 0 || q0 == Q - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
            }
          }
        };
        for (int base = 0; base < ncand && !dead; base += 64 * kTrip) {
          float fsc[kTrip];
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            int q, t; float g;
            cand(base + 64 * u + lane, q, t, g);
            fsc[u] = a.h_mode == 0 ? H32p[(size_t)q * ld + t] : (float)H16p[(size_t)q * ld + t];
          }
          unsigned long long mine = 0ull; bool any = false;
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            const int idx = base + 64 * u + lane;
            int q, t; float g;
            cand(idx, q, t, g);
            const bool ok = idx < ncand && (idx == 0 ? fsc[u] + r > thr : fsc[u] + r - g > thr);
            const unsigned long long m = __ballot(ok);
            if (lane == u) mine = m;
            any = any || m != 0ull;
          }
          if (!any) continue;
          for (int u = 0; u < kTrip && !dead; ++u) {
            const unsigned long long m = ((unsigned long long)(uint32_t)__shfl((int)(mine >> 32), u) << 32) | (uint32_t)__shfl((int)(mine & 0xFFFFFFFFull), u);
            if (!m) continue;
            const int idx = base + 64 * u + lane;
            int q, t; float g;
            cand(idx, q, t, g);
            spawn(m, g, q, t, idx);
          }
        }
      } else {
        for (int base = 0; base < ncand && !dead; base += 64) {
          const int idx = base + lane;
          bool ok = false; float g = 0.f; int pq = 0, pt = 0;
          if (idx < ncand) {
            if (idx == 0) {                                     // match, cw.h:151-162
              pq = q0 - 1; pt = t0 - 1;
              ok = HV(pq, pt) + r > thr;
            } else if (idx <= ndel) {                           // deletions i = t0-2 .. 1, cw.h:166-178
              pq = q0 - 1; pt = t0 - 1 - idx;
              g = dev_deletion(e, pt, t0);
              ok = HV(pq, pt) + r - g > thr;
            } else {                                            // insertions j = q0-2 .. 1, cw.h:182-194
              pq = q0 - 2 - (idx - ndel - 1); pt = t0 - 1;
              g = dev_insertion(e, pq, q0, pt, t0);
              ok = HV(pq, pt) + r - g > thr;
            }
          }
          const unsigned long long m = __ballot(ok);
          if (m) spawn(m, g, pq, pt, idx);
        }
      }
      if (dead || !first) break;                                            // children carry on (or the pools are exhausted)
      is_branch = false; force = true;                                      // nothing passed: finish along stored pointers, cw.h:196-203 / ucw.h:186-191
    }
  }
  if (threadIdx.x == 0) {
    const int status = s_status;
    a.out[0] = (int32_t)s_slots; a.out[1] = (int32_t)s_nodes; a.out[2] = status;
  }
}

}  // namespace aln
