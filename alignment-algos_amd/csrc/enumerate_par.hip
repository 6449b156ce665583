// enumerate_par.hip — ConstrainedNearOptimal / UnconstrainedNearOptimal (cw.h:68-284, ucw.h:64-236) searched by SEVERAL waves
// per pair (gfx950).
//
// enumerate.hip walks a pair's depth-first search with one wave: every step of it is a chain of dependent memory round trips
// (frame -> candidate scores -> pool words -> pointer word -> ...), so 1024 pairs keep 1024 SIMDs busy ~1 % of the time.  The
// search tree itself is wide: every candidate a branch node accepts starts an independent sub-search (its own alignment, its
// own score; the only shared state of the reference's recursion is the ORDER in which alignments are appended to the set).
// This kernel therefore
//   * turns every sub-search into a task (cell, alignment slot, trie head, score so far) in a per-pair ring in HBM,
//   * lets the W waves of the pair's workgroup take tasks on their own (tickets handed out with LDS atomics, a ready word per
//     record; no barrier inside the search),
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
// Termination: s_tail counts tickets handed out, s_done tasks finished (a task's pushes come before its s_done increment); a wave
// that finds no ticket reads s_done, then s_tail: equal means that at the moment s_done was read nothing was pending or running.
// Everything the waves exchange is workgroup scope (one CU): an agent-scope fence per step made this kernel 3 x slower (it writes
// the XCD's L2 back).  Sequence codes, the substitution table and the SuboptFlags row live in LDS; trie nodes, slots and tickets
// are handed out with LDS atomics.  Results are bit-identical to enumerate_kernel's (tests run both against the reference's sets).
#include "enum_common.h"
#include "enum_sort.h"

namespace aln {

constexpr int kNotReading = 0x7FFFFFFF;   // s_reading: this wave holds no unread ticket
constexpr int kKsCap = 256;      // KSCW in this kernel: candidates of one branch node that may pass the threshold (more: the one-wave kernel)

__global__ __launch_bounds__(1024) void enumerate_par_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto,
                                                            const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                            const float* __restrict__ tgi, const float* __restrict__ tge,
                                                            const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                                            const float* __restrict__ Sbase, EnumArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int s_head, s_tail, s_done, s_status;
  __shared__ unsigned s_nodes, s_slots;
  __shared__ unsigned long long s_alloc;  // node allocator of this pair: current chunk << 32 | nodes used in it
  __shared__ int s_lock;
  __shared__ int s_reading[16];           // per wave: the ticket it is taking / has taken and not yet read (kNotReading otherwise)
  __shared__ uint16_t s_chunk[16][136];   // per wave: the blocks of a branch node's candidate row / column that can hold a passing candidate
  {
    const size_t bi = blockIdx.x;
    pair = a.pair_list ? a.pair_list[bi] : pair + (int)bi;
    a.head += bi * a.ali_cap; a.score += bi * a.ali_cap;
    {                                                                 // node pool of this workgroup (enum_pool_of scatters the pairs over the pools)
      const size_t pool = enum_pool_of((uint32_t)bi, a.n_pools);
      a.node_pair += pool * (size_t)a.n_chunks * kChunkNodes; a.node_next += pool * (size_t)a.n_chunks * kChunkNodes;
      a.node_len += pool * (size_t)a.n_chunks * kChunkNodes;
      a.chunk_next += pool;
    }
    a.task += bi * (size_t)a.ali_cap * kTaskWords;
    a.slot_info += bi * (size_t)a.ali_cap * 3;
    a.flags += (size_t)(a.pair_list ? pair : (int)bi) * (size_t)a.flags_stride;
    a.out += bi * 4;
  }
  const PairDesc pd = pairs[pair];
  const int Q = pd.Q, T = pd.T, ld = pd.ld;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool cw = a.kind == ALN_ENUM_CW, ks = a.kind == ALN_ENUM_KSCW;
  const bool constrained = cw || ks;                  // opt_path stops where the template's SuboptFlags bit flips

  // ---- per-pair constants into LDS: flags | template codes | query codes | 32 x 32 table -----------------------------------
  const int Tp = (T + 15) & ~15, Qp = (Q + 15) & ~15;
  uint8_t* l_fl = lds_raw;
  uint8_t* l_tc = l_fl + Tp;
  uint8_t* l_qc = l_tc + Tp;
  float* l_tab = reinterpret_cast<float*>(l_qc + Qp);
  float* ks_sc = l_tab + 1024 + (size_t)w * kKsCap;                        // KSCW: this wave's candidate sums ...
  int* ks_ix = reinterpret_cast<int*>(l_tab + 1024 + (size_t)(blockDim.x >> 6) * kKsCap) + (size_t)w * kKsCap;   // ... and indices
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
  // DPMatrix::getSim: from the LDS copies (ds_read, not the flat loads a generic EvalDev pointer costs) for a substitution table
  auto SIM = [&](int i, int j) -> float {
    if (!sub) return dev_sim(e, i, j);
    if (i <= 0 || j <= 0 || i >= Q - 1 || j >= T - 1) return 0.f;
    return l_tab[(int)l_qc[i] * 32 + (int)l_tc[j]];
  };

  const float top = HV(Q - 1, T - 1);
  float thr = (1.f - a.delta_ratio) * top;       // cw.h:86-88
  { float alt = top - 0.1f; thr = (alt < thr) ? alt : thr; }

  const uint32_t qcap = a.ali_cap;                    // ring of task records (every pending task owns a distinct slot, so <= ali_cap pend)
  const uint32_t qmargin = qcap / 4u < 2048u ? qcap / 4u : 2048u;
  if (threadIdx.x < 16) s_reading[threadIdx.x] = kNotReading;
  if (threadIdx.x == 0) {
    s_head = 0; s_tail = 1; s_done = 0; s_status = 0; s_nodes = 0u; s_lock = 0;
    s_slots = (unsigned)a.first_slot + 1u;             // as.push_back(SingleAlignment())  cw.h:82 / ucw.h:78
    s_alloc = (unsigned long long)kChunkNodes;        // "current chunk is full": the first allocation fetches one
    uint32_t* tk = a.task;                                                            // branch(final cell, seed slot)  cw.h:92 / ucw.h:86
    st_w(tk + 0, ((uint32_t)(Q - 1) << 16) | (uint32_t)(T - 1)); st_w(tk + 1, (uint32_t)a.first_slot); st_w(tk + 2, kNoNode);
    st_w(tk + 3, __float_as_uint(0.f)); st_w(tk + 4, 1u); st_w(tk + 6, a.k_limit);
    __hip_atomic_store(tk + 5, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // ticket 0 is ready
    st_u(&a.head[a.first_slot], kNoNode); st_f(&a.score[a.first_slot], 0.f);
  }
  __syncthreads();

  auto fail = [&](int code) { if (lane == 0) atomicCAS(&s_status, 0, code); };
  // n <= 512 consecutive trie nodes for this wave, out of the pair's current chunk of the launch-wide pool; kNoNode: the pool is
  // exhausted (status set).  Chunk and offset come from ONE 64-bit LDS atomic, so a wave can never pair an old offset with a new
  // chunk; the wave that finds the chunk full fetches the next one under a lock (one device-scope atomic per 65536 nodes).
  auto alloc_nodes = [&](int n) -> uint32_t {
    uint32_t b = kNoNode;
    if (lane == 0) {
      for (int spin = 0; spin < (1 << 22); ++spin) {
        const unsigned long long v = atomicAdd(&s_alloc, (unsigned long long)n);
        const uint32_t chunk = (uint32_t)(v >> 32), off = (uint32_t)v;
        if (off + (uint32_t)n <= kChunkNodes) { b = chunk * kChunkNodes + off; atomicAdd(&s_nodes, (unsigned)n); break; }
        if (__hip_atomic_load(&s_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
        if (atomicCAS(&s_lock, 0, 1) == 0) {
          const unsigned long long cur = __hip_atomic_load(&s_alloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((uint32_t)(cur >> 32) == chunk && (uint32_t)cur + 512u > kChunkNodes) {        // still the full chunk: replace it
            const uint32_t c = __hip_atomic_fetch_add(a.chunk_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= a.n_chunks) atomicCAS(&s_status, 0, ALN_E_OVERFLOW);
            else __hip_atomic_store(&s_alloc, (unsigned long long)c << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          __hip_atomic_store(&s_lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (b == kNoNode) atomicCAS(&s_status, 0, ALN_E_OVERFLOW);
    }
    return (uint32_t)__shfl((int)b, 0);
  };

  // Each wave keeps a reserve of nodes it took from the pair's chunk (512 at a time), so that a step of a walk costs no LDS atomic.
  uint32_t w_base = 0, w_left = 0;
  auto take_nodes = [&](int n) -> uint32_t {
    if (w_left < (uint32_t)n) {                          // (what is left of the old reserve stays unused: < 64 nodes per 512)
      const uint32_t b = alloc_nodes(512);
      if (b == kNoNode) return kNoNode;
      w_base = b; w_left = 512;
    }
    const uint32_t b = w_base;
    w_base += (uint32_t)n; w_left -= (uint32_t)n;
    return b;
  };

  // Every wave takes tickets on its own: s_head = next ticket to take, s_tail = next ticket to hand out (a task's children are
  // pushed by the wave that ran the task), s_done = tasks finished.  The search is over when every ticket handed out is done.
  for (;;) {
    int tkt = -1;
    if (lane == 0) {
      for (long spin = 0;; ++spin) {
        if (__hip_atomic_load(&s_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
        const int d = __hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (read before s_tail: see header)
        const int h = __hip_atomic_load(&s_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int t = __hip_atomic_load(&s_tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (h < t) {
          // announce the ticket BEFORE taking it (LDS operations of one wave are served in order): a pusher that sees s_head
          // beyond h also sees the announcement, so min(s_head, announcements) never passes an unread record
          __hip_atomic_store(&s_reading[w], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          int expect = h;
          if (__hip_atomic_compare_exchange_strong(&s_head, &expect, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { tkt = h; break; }
          __hip_atomic_store(&s_reading[w], kNotReading, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // lost the race for h
          continue;
        }
        if (d == t) break;                               // nothing pending, nothing running
        if (spin > (1L << 22)) { atomicCAS(&s_status, 0, ALN_E_OVERFLOW); break; }   // (a wait of seconds: something is broken, end the search)
        __builtin_amdgcn_s_sleep(1);
      }
    }
    tkt = __shfl(tkt, 0);
    if (tkt < 0) break;
    uint32_t tw0, tw1, tw2, tw3, tw4, tw6;
    {
      const uint32_t* tk = a.task + (size_t)((uint32_t)tkt % qcap) * kTaskWords;
      bool ready = false;
      for (long spin = 0; spin < (1L << 22); ++spin) {    // the pusher writes the record after it took the ticket range
        if (__hip_atomic_load(tk + 5, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)tkt + 1u) { ready = true; break; }
        if (__hip_atomic_load(&s_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!__shfl((int)ready, 0)) {                      // the search has failed elsewhere (or a record never came: broken): leave
        if (lane == 0) atomicCAS(&s_status, 0, ALN_E_OVERFLOW);
        break;
      }
      const uint32_t v = lane < 7 ? ld_w(tk + lane) : 0u;
      tw0 = (uint32_t)__shfl((int)v, 0); tw1 = (uint32_t)__shfl((int)v, 1); tw2 = (uint32_t)__shfl((int)v, 2);
      tw3 = (uint32_t)__shfl((int)v, 3); tw4 = (uint32_t)__shfl((int)v, 4); tw6 = (uint32_t)__shfl((int)v, 6);
      if (lane == 0) __hip_atomic_store(&s_reading[w], kNotReading, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the record is in registers
    }

    // ---- one task: opt_path / branch of ONE alignment slot until its branch node has spawned its children ---------------
    int q0 = (int)(tw0 >> 16), t0 = (int)(tw0 & 0xFFFFu);
    const uint32_t slot = tw1;
    uint32_t hd = tw2;
    float sc = __uint_as_float(tw3);
    bool is_branch = (tw4 & 1u) != 0, force = (tw4 & 2u) != 0;
    uint32_t klimit = tw6;                                // KSCW: operations this alignment's next branch node may keep (kscw.h:246-247)
    bool dead = false;
    if (q0 < 1 || t0 < 1 || q0 >= Q || t0 >= T || slot >= a.ali_cap) { fail(ALN_E_OVERFLOW); break; }   // not a record of this search

    // as[slot].prepend(q0,t0); as[slot].prepend(0,0); score += H(q0,t0)   (cw.h:100-108 / ucw.h:93-101), then publish the slot
    auto base_case = [&]() {
      const uint32_t b = take_nodes(2);
      if (b == kNoNode) return;
      if (lane == 0) {
        a.node_pair[b] = ((uint32_t)q0 << 16) | (uint32_t)t0; a.node_next[b] = hd; a.node_len[b] = 1;
        a.node_pair[b + 1] = 0u; a.node_next[b + 1] = b; a.node_len[b + 1] = 1;
        a.head[slot] = b + 1;
        a.score[slot] = sc + HV(q0, t0);
      }
    };

    int guard = 0;
    while (!dead) {
      if (++guard > 2 * (Q + T) + 64) { fail(ALN_E_OVERFLOW); break; }       // (a path that never reaches row / column 1: not a DP matrix)
      if (q0 == 1 || t0 == 1) { base_case(); break; }                      // cw.h:220-228, :113-121
      if (!is_branch) {
        if (ks && klimit <= 1) force = true;                              // kscw.h:297-299
        // opt_path: follow stored pointers, 64 diagonal cells at a time (see enumerate.hip walk())
        const bool flag = constrained ? !l_fl[t0] : false;
        while (t0 > 1 && q0 > 1) {
          const int q = q0 - lane, t = t0 - lane;
          bool stop = !(q > 1 && t > 1);
          if (!stop && constrained && !force && ((l_fl[t] != 0) == flag)) stop = true;
          int pq = 0, pt = 0; float sv = 0.f, g = 0.f;
          if (!stop) {
            const uint32_t p = load_ptr_word(Pbase, pd.plane_off, ld, q, t, a.ptr_mode);
            decode_ptr(p, a.ptr_mode, q, t, pq, pt);
            sv = SIM(q, t);
          }
          const bool diag = !stop && pq == q - 1 && pt == t - 1;
          const unsigned long long m_end = __ballot(!diag);
          const int F = m_end ? __builtin_ctzll(m_end) : 64;
          const bool gap_cell = F < 64 && !(((__ballot(stop)) >> F) & 1ull);
          const int n_proc = gap_cell ? F + 1 : F;
          if (n_proc == 0) break;
          const uint32_t b = take_nodes(1);              // ONE trie node for the run of n_proc diagonal cells from (q0,t0) down
          if (b == kNoNode) { dead = true; break; }
          if (lane == 0) { a.node_pair[b] = ((uint32_t)q << 16) | (uint32_t)t; a.node_next[b] = hd; a.node_len[b] = (uint8_t)n_proc; }
          hd = b;
          if (gap_cell && lane == F) {
            if (q - pq == 1) g = dev_deletion(e, pt, t);
            else g = dev_insertion(e, pq, q, pt, t);
          }
          if (a.int_sums) {                                   // integer-valued scores below 2^24: any order gives the reference's sum
            sc += (float)wave_sum_i32(lane < n_proc ? (int)sv : 0);
          } else {
            sc = add_in_path_order(sc, sv, n_proc);            // path order: fp32 is not associative
          }
          if (gap_cell) {
            sc -= __shfl(g, F);
            q0 = __shfl(pq, F); t0 = __shfl(pt, F);
          } else { q0 -= n_proc; t0 -= n_proc; }
        }
        if (dead) break;
        if (!constrained) { base_case(); break; }                           // ucw.h:232-234
        is_branch = true;                                                   // branch(pq,pt,k0,force)  cw.h:276
        continue;
      }
      // ---- branch(q0,t0,slot) ----
      if (cw && force) { is_branch = false; continue; }                     // cw.h:205-209: opt_path(..., true)
      const float r = sc + SIM(q0, t0);
      const int ndel = t0 - 2, nins = q0 - 2;
      const int ncand = 1 + ndel + nins;
      bool first = true;                                                    // the first accepted candidate continues in `slot`
      // every accepted candidate of one 64-candidate group: a trie node (q0,t0) in front of hd, a slot (new ones are recorded
      // with their place in the reference's order) and a task
      auto spawn = [&](unsigned long long m, float g, int cq, int ct, int idx, uint32_t lim = 0u) {
        const int n = __popcll(m);
        const int nnew = n - (first ? 1 : 0);
        uint32_t bt = 0, bn = 0, bs = 0;
        bn = take_nodes(n);
        if (bn == kNoNode) { dead = true; return; }
        if (lane == 0) {
          bt = (uint32_t)atomicAdd(&s_tail, n);
          if (nnew) bs = atomicAdd(&s_slots, (unsigned)nnew);
        }
        bt = (uint32_t)__shfl((int)bt, 0); bs = (uint32_t)__shfl((int)bs, 0);
        if (nnew && bs + (uint32_t)nnew > a.user_limit) { fail(kParSerial); dead = true; return; }   // the serial order decides what user_limit cuts
        // floor = the oldest ticket whose record may still be unread: s_head (read first), lowered by every wave's announcement
        int floor_t = __hip_atomic_load(&s_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
          int mine = lane < 16 ? __hip_atomic_load(&s_reading[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : kNotReading;
          for (int o = 8; o; o >>= 1) { const int other = __shfl_xor(mine, o); mine = other < mine ? other : mine; }
          mine = __shfl(mine, 0);
          floor_t = mine < floor_t ? mine : floor_t;
        }
        const uint32_t pending = bt + (uint32_t)n - (uint32_t)floor_t;
        // (every wave of the workgroup may be pushing 64 records at this moment: keep that much of the ring free as well)
        if (pending + qmargin > qcap || bs + (uint32_t)nnew > a.ali_cap) {
          fail(ALN_E_OVERFLOW); dead = true; return;
        }
        if ((m >> lane) & 1ull) {
          const int rnk = __popcll(m & ((1ull << lane) - 1ull));
          const uint32_t nd = bn + (uint32_t)rnk;
          a.node_pair[nd] = ((uint32_t)q0 << 16) | (uint32_t)t0;
          a.node_next[nd] = hd;
          a.node_len[nd] = 1;
          const int srank = rnk - (first ? 1 : 0);
          uint32_t sl = slot;
          if (srank >= 0) {
            sl = bs + (uint32_t)srank;
            uint32_t* si = a.slot_info + (size_t)sl * 3;
            si[0] = slot; si[1] = (uint32_t)t0; si[2] = (uint32_t)idx;
          }
          const uint32_t ticket = bt + (uint32_t)rnk;
          uint32_t* tk = a.task + (size_t)(ticket % qcap) * kTaskWords;
          st_w(tk + 0, ((uint32_t)cq << 16) | (uint32_t)ct); st_w(tk + 1, sl); st_w(tk + 2, nd);
          st_w(tk + 3, __float_as_uint(r - g)); st_w(tk + 4, constrained ? 0u : 1u);   // cw, kscw: opt_path(cand, k, false); ucw: branch(cand, k)
          st_w(tk + 6, lim);
          __hip_atomic_store(tk + 5, ticket + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        first = false;
      };
      if (ks) {
        // KSConstrainedNearOptimal's branch node (kscw.h:139-288): EVERY passing predecessor in the reference's scan order into this
        // wave's LDS arrays, libstdc++'s sort / partial_sort on them (enum_sort.h, one lane: the order of equal scores is observable),
        // the k_limit best become the node's operations — the best keeps (about) the node's limit, the others get half of it.
        int n = 0;
        auto collect = [&](bool ok, float sum, int idx) {                   // one 64-candidate group, appended in lane order
          const unsigned long long m = __ballot(ok);
          if (m) {
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && pos < kKsCap) { ks_sc[pos] = sum; ks_ix[pos] = idx; }
            n += __popcll(m);
          }
        };
        if (e.model == ALN_GAP_AFFINE_CONST && a.rowmax && Q <= 4096 && T <= 4096 && e.gi >= 0.f && e.ge >= 0.f) {
          // only the 64-cell blocks whose maximum can pass (see the pruned scan below), in the reference's candidate order
          const bool kfdel = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
          const bool kfins = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
          const float* rmax = a.rowmax + ((size_t)(pair - a.bm_pair0) * a.bm_rows + (size_t)(q0 - 1)) * a.nbt;
          const float* cmax = a.colmax + ((size_t)(pair - a.bm_pair0) * a.bm_cols + (size_t)(t0 - 1)) * a.nbq;
          bool pass_d = false, pass_i = false;
          {
            const int lo = lane * 64 > 1 ? lane * 64 : 1;
            int hi = lane * 64 + 63; hi = hi < t0 - 2 ? hi : t0 - 2;
            if (lo <= hi) {
              const int len = t0 - hi - 1;
              const float g = (len < 1 || (kfdel && t0 == T - 1)) ? 0.f : e.gi + e.ge * (float)(len - 1);
              pass_d = (rmax[lane] + r) - g > thr;
            }
            int hq = lane * 64 + 63; hq = hq < q0 - 2 ? hq : q0 - 2;
            if (lo <= hq) {
              const int len = q0 - hq - 1;
              const float g = (len < 1 || (kfins && q0 == Q - 1)) ? 0.f : e.gi + e.ge * (float)(len - 1);
              pass_i = (cmax[lane] + r) - g > thr;
            }
          }
          const unsigned long long md = __ballot(pass_d), mi = __ballot(pass_i);
          { const float sum = HV(q0 - 1, t0 - 1) + r; collect(lane == 0 && sum > thr, sum, 0); }
          for (unsigned long long mm = md; mm; ) {
            const int blk = 63 - __builtin_clzll(mm); mm &= ~(1ull << blk);
            const int pt = blk * 64 + 63 - lane;
            const bool in = pt >= 1 && pt <= t0 - 2;
            const float sum = in ? HV(q0 - 1, pt) + r - dev_deletion(e, pt, t0) : 0.f;
            collect(in && sum > thr, sum, t0 - 1 - pt);
          }
          for (unsigned long long mm = mi; mm; ) {
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
            if (idx == 0) sum = HV(q0 - 1, t0 - 1) + r;
            else if (idx <= ndel) { const int pt = t0 - 1 - idx; sum = HV(q0 - 1, pt) + r - dev_deletion(e, pt, t0); }
            else { const int pq = q0 - 2 - (idx - ndel - 1); sum = HV(pq, t0 - 1) + r - dev_insertion(e, pq, q0, t0 - 1, t0); }
            ok = sum > thr;
          }
          collect(ok, sum, idx);
        }
        if (n > kKsCap) { fail(kParSerial); break; }                        // more candidates than this kernel keeps per wave
        if (n == 0) { is_branch = false; force = true; klimit = 1; continue; }   // kscw.h:222-228: op(1, q0, t0, k0), forced
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int m_keep = n;
        if (lane == 0) {
          kssort::Arr arr = {ks_sc, ks_ix};
          if ((uint32_t)n > klimit) kssort::partial_sort(arr, 0, (int)klimit, n);
          else kssort::sort(arr, 0, n);
        }
        if ((uint32_t)n > klimit) m_keep = (int)klimit;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (m_keep == 0) { fail(ALN_E_ARG); break; }                        // limit 0 with candidates (`it->limit *= 2` on an empty vector in the reference)
        int pq = q0 - 1, pt = t0 - 1; float g = 0.f;
        if (lane < m_keep) {
          const int idx = ks_ix[lane];
          if (idx == 0) { }
          else if (idx <= ndel) { pt = t0 - 1 - idx; g = dev_deletion(e, pt, t0); }
          else { pq = q0 - 2 - (idx - ndel - 1); pt = t0 - 1; g = dev_insertion(e, pq, q0, pt, t0); }
        }
        uint32_t lim = klimit / 2;
        if (lane == 0) lim *= 2;                                            // only the best operation keeps (about) the node's limit
        spawn(m_keep >= 64 ? ~0ull : ((1ull << m_keep) - 1ull), g, pq, pt, lane, lim);   // ordinal = place in the sorted list
        __builtin_amdgcn_wave_barrier();
        break;                                                              // the operations carry on
      }
      const bool fdel = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
      const bool fins = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
      const uint16_t* H16p = reinterpret_cast<const uint16_t*>(Hbase) + pd.plane_off;
      const float* H32p = Hbase + pd.plane_off;
      if (e.model == ALN_GAP_AFFINE_CONST && a.rowmax && Q <= 4096 && T <= 4096 && e.gi >= 0.f && e.ge >= 0.f) {
        // Pruned scan.  A candidate passes when fl(fl(H + r) - g) > thr; both roundings are monotone and, with gi, ge >= 0, g does
        // not decrease with the gap's length.  So a 64-cell block of the candidate row / column can hold a passing candidate only
        // if fl(fl(max H of the block + r) - g of its cell nearest to the node) > thr: one lane tests one block against the
        // block maxima enum_blockmax_kernel prepared, and only the surviving blocks are read (for a near-optimal search: the one
        // or two next to the node instead of all 31 + 31, and the column candidates are one 64-byte sector each).
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
        const int nd = __popcll(md), total = 1 + nd + __popcll(mi);
        // chunk list in candidate order: the match, deletion blocks from t0-2 down, insertion blocks from q0-2 down
        uint16_t* chunks = s_chunk[w];
        if (lane == 0) chunks[0] = 0;
        if (pass_d) chunks[1 + __popcll(lane < 63 ? md >> (lane + 1) : 0ull)] = (uint16_t)(0x100 | lane);
        if (pass_i) chunks[1 + nd + __popcll(lane < 63 ? mi >> (lane + 1) : 0ull)] = (uint16_t)(0x200 | lane);
        __builtin_amdgcn_wave_barrier();
        // one candidate per lane and chunk; inside a block the lanes run against the index so that lane order = candidate order
        auto cand = [&](int code, int& q, int& t, float& g, int& idx) -> bool {
          const int kind = code >> 8, blk = code & 0xFF;
          q = q0 - 1; t = t0 - 1; g = 0.f; idx = 0;
          if (kind == 0) return lane == 0;
          const int pos = blk * 64 + 63 - lane;
          if (kind == 1) {                                      // aasubalib.h:27-51
            const bool in = pos >= 1 && pos <= t0 - 2;
            if (in) t = pos;
            idx = t0 - 1 - t;
            const int len = t0 - t - 1;
            g = (len < 1 || (fdel && (t == 0 || t0 == T - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
            return in;
          }
          const bool in = pos >= 1 && pos <= q0 - 2;           // aasubalib.h:53-77
          if (in) q = pos;
          idx = ndel + 1 + (q0 - 2 - q);
          const int len = q0 - q - 1;
          g = (len < 1 || (fins && (q == 0 || q0 == Q - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
          return in;
        };
        constexpr int kTrip = 4;                              // (a near-optimal search rarely keeps more than the match and 2-3 blocks)
        for (int c0 = 0; c0 < total && !dead; c0 += kTrip) {
          float fsc[kTrip];
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            const int code = c0 + u < total ? (int)chunks[c0 + u] : 0;
            int q, t, idx; float g;
            cand(code, q, t, g, idx);
            fsc[u] = a.h_mode == 0 ? H32p[(size_t)q * ld + t] : (float)H16p[(size_t)q * ld + t];
          }
          unsigned long long mine = 0ull; bool any = false;
#pragma unroll
          for (int u = 0; u < kTrip; ++u) {
            const int code = c0 + u < total ? (int)chunks[c0 + u] : 0;
            int q, t, idx; float g;
            const bool in = cand(code, q, t, g, idx) && c0 + u < total;
            const bool ok = in && (idx == 0 ? fsc[u] + r > thr : fsc[u] + r - g > thr);
            const unsigned long long m = __ballot(ok);
            if (lane == u) mine = m;
            any = any || m != 0ull;
          }
          if (!any) continue;
          for (int u = 0; u < kTrip && !dead; ++u) {
            const unsigned long long m = ((unsigned long long)(uint32_t)__shfl((int)(mine >> 32), u) << 32) | (uint32_t)__shfl((int)(mine & 0xFFFFFFFFull), u);
            if (!m) continue;
            int q, t, idx; float g;
            cand((int)chunks[c0 + u], q, t, g, idx);
            spawn(m, g, q, t, idx);
          }
        }
        __builtin_amdgcn_wave_barrier();
      } else if (e.model == ALN_GAP_AFFINE_CONST) {
        // as in enumerate.hip, 8 x 64 score loads in flight per trip; the accept masks of a trip are parked one per lane
        constexpr int kTrip = 8;
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
    if (lane == 0) __hip_atomic_fetch_add(&s_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // after this task's pushes
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int status = s_status;
    a.out[0] = (int32_t)s_slots; a.out[1] = (int32_t)s_nodes; a.out[2] = status;
  }
}


// Block maxima of the score plane for the pruned scan above: colmax[t][bq] = max of H over rows 64 bq .. 64 bq + 63 of column t,
// rowmax[q][bt] = max over columns 64 bt .. 64 bt + 63 of row q.  grid (row tiles, pairs), 256 threads; every cell is read once.
__global__ __launch_bounds__(256) void enum_blockmax_kernel(const PairDesc* __restrict__ pairs, const int32_t* __restrict__ pair_list, int pair0,
                                                           const float* __restrict__ Hbase, int h_mode, float* __restrict__ rowmax,
                                                           float* __restrict__ colmax, int bm_rows, int bm_cols, int nbt, int nbq, int bm_pair0) {
  const int pair = pair_list ? pair_list[blockIdx.y] : pair0 + (int)blockIdx.y;
  const PairDesc pd = pairs[pair];
  const int Q = pd.Q, T = pd.T, ld = pd.ld, bq = blockIdx.x;
  const int qa = bq * 64, qb = qa + 64 < Q ? qa + 64 : Q;
  if (qa >= Q) return;
  const float NEG = -3.0e38f;
  float* cm = colmax + (size_t)(pair - bm_pair0) * bm_cols * nbq;
  float* rm = rowmax + (size_t)(pair - bm_pair0) * bm_rows * nbt;
  // one pass: thread = one column of a 256-column strip; the column maximum of the 64 rows stays in a register, the row
  // maxima of the strip's four 64-column blocks come from a DPP max per row and wave
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool in = t < T;
    float cmaxv = NEG;
    for (int q = qa; q < qb; ++q) {
      const float v = in ? load_score(Hbase, pd.plane_off, ld, q, t, h_mode) : NEG;
      cmaxv = fmaxf(cmaxv, v);
      const float rv = wave_max_f32(v);
      const int bt = (t0 >> 6) + w;
      if (lane == 0 && bt * 64 < T) rm[(size_t)q * nbt + bt] = rv;
    }
    if (in) cm[(size_t)t * nbq + bq] = cmaxv;
  }
}

}  // namespace aln
