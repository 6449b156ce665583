// enumerate.hip — Waterman-style near-optimal alignment enumeration on the resident DP planes (gfx950).
//
// Reference: ConstrainedNearOptimal (cw.h:68-284) and UnconstrainedNearOptimal (ucw.h:64-236): a depth-first
// branching traceback from the final cell; at a branch node every predecessor (match, then row q0-1 for
// i = t0-2..1 descending, then column t0-1 for j = q0-2..1 descending) whose forward score plus the reverse
// score so far minus the gap exceeds the threshold starts a new alignment (the first one continues in the
// current slot, later ones copy the state before the node); opt_path follows stored pointers until the
// template's SuboptFlags bit flips.  Discovery order decides the position of every alignment in the set.
//
// Device representation: alignments only ever grow by prepend(), so they form a trie — a node is one aligned
// pair plus a link to the pair after it; an alignment is (head node, score); "copy the alignment" is copying
// two words.  One wave walks one pair's DFS with an explicit stack of branch frames; the candidate scan of a
// branch node is done 64 predecessors at a time (ballot keeps the reference's order); fp32 sums are formed in
// the written order (f + r > thr, (f + r) - g > thr, score = r - g).  The host then sorts (score, index)
// with the same std::sort / std::partial_sort call the reference uses (alignment.h:922-932) and asks the
// device to unroll only the surviving alignments.
#include "enum_common.h"

namespace aln {

__global__ __launch_bounds__(64) void enumerate_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto,
                                                       const uint8_t* __restrict__ qcodes, const uint8_t* __restrict__ tcodes,
                                                       const float* __restrict__ tgi, const float* __restrict__ tge,
                                                       const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                                       const float* __restrict__ Sbase, EnumArgs a) {
  {
    const size_t bi = blockIdx.x;                 // position inside the group: every pool is sliced by it
    pair = a.pair_list ? a.pair_list[bi] : pair + (int)bi;
    a.node_pair += bi * a.node_cap; a.node_next += bi * a.node_cap;
    a.head += bi * a.ali_cap; a.score += bi * a.ali_cap;
    a.stack += bi * (size_t)a.stack_cap * kFrameWords;
    a.flags += (size_t)(a.pair_list ? pair : (int)bi) * (size_t)a.flags_stride;   // flag rows are indexed by the pair's place in the batch
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
  const bool cw = a.kind == ALN_ENUM_CW;

  uint32_t n_as = (uint32_t)a.first_slot + 1;   // as.push_back(SingleAlignment())  cw.h:82 / ucw.h:78
  uint32_t n_nodes = 0;
  int status = 0;
  // Lane 0 writes the pools, every lane reads them back through L2.  A store is only waited for (one memory round trip) right
  // before the next cross-lane read: `dirty` marks stores in flight, flush() is the s_waitcnt.  (Lane 0 re-reading a word it
  // wrote itself needs no wait: same lane, same address, program order.)
  bool dirty = false;
  auto flush = [&]() { if (dirty) { __builtin_amdgcn_s_waitcnt(0); dirty = false; } };
  if (lane == 0) { st_u(&a.head[a.first_slot], kNoNode); st_f(&a.score[a.first_slot], 0.f); }
  dirty = true;
  flush();

  const float top = HV(Q - 1, T - 1);
  float thr = (1.f - a.delta_ratio) * top;       // cw.h:86-88
  { float alt = top - 0.1f; thr = (alt < thr) ? alt : thr; }

  auto prepend = [&](int k, int q, int t) {     // as[k].prepend(q,t): a new trie node in front of the list
    if (n_nodes >= a.node_cap) { status = ALN_E_OVERFLOW; return; }
    if (lane == 0) {
      a.node_pair[n_nodes] = ((uint32_t)q << 16) | (uint32_t)t;
      a.node_next[n_nodes] = ld_u(&a.head[k]);
      st_u(&a.head[k], n_nodes);
    }
    ++n_nodes;
    dirty = true;
  };
  auto set_score = [&](int k, float s) { if (lane == 0) st_f(&a.score[k], s); dirty = true; };
  auto base_case = [&](int q0, int t0, int k0) {   // cw.h:100-108 / ucw.h:93-101
    prepend(k0, q0, t0);
    prepend(k0, 0, 0);
    flush();
    float s = ld_f(&a.score[k0]);
    s += HV(q0, t0);
    set_score(k0, s);
  };
  // the pointer-following loop of opt_path (cw.h:242-272 / ucw.h:207-228); returns the cell it stopped at.
  // 64 cells of the current diagonal are examined at once (lane l: cell (q0-l, t0-l)): the leading run of cells whose
  // stored pointer is the match pointer is taken in one step — their trie nodes are written in parallel, their
  // similarities are summed in path order (fp32 is not associative) — and a gap cell ends the run with one jump.
  auto walk = [&](int& q0, int& t0, int k0, bool force) {
    const bool flag = cw ? !a.flags[t0] : false;
    float sc = ld_f(&a.score[k0]);
    uint32_t hd = ld_u(&a.head[k0]);
    while (t0 > 1 && q0 > 1 && status == 0) {
      const int q = q0 - lane, t = t0 - lane;
      bool stop = !(q > 1 && t > 1);
      if (!stop && cw && !force && ((a.flags[t] != 0) == flag)) stop = true;   // the template's SuboptFlags bit flipped: branch point
      int pq = 0, pt = 0; float sv = 0.f, g = 0.f;
      if (!stop) {
        const uint32_t p = load_ptr_word(Pbase, pd.plane_off, ld, q, t, a.ptr_mode);
        decode_ptr(p, a.ptr_mode, q, t, pq, pt);
        sv = dev_sim(e, q, t);
      }
      const bool diag = !stop && pq == q - 1 && pt == t - 1;
      const unsigned long long m_end = __ballot(!diag);
      const int F = m_end ? __builtin_ctzll(m_end) : 64;          // first cell that is not a plain match step
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
      sc = add_in_path_order(sc, sv, n_proc);       // sc += sim (then sc -= 0 for a match step), in path order
      if (gap_cell) {
        sc -= __shfl(g, F);
        q0 = __shfl(pq, F); t0 = __shfl(pt, F);
      } else { q0 -= n_proc; t0 -= n_proc; }
    }
    if (lane == 0) { st_u(&a.head[k0], hd); st_f(&a.score[k0], sc); }
    dirty = true;
  };

  // Depth-first search with an explicit stack of branch frames.  A pending call is either
  //   CALL_BRANCH: branch(q,t,k,force)      or      CALL_OPT: opt_path(q,t,k,force)
  enum { CALL_NONE = 0, CALL_BRANCH = 1, CALL_OPT = 2 };
  int call = CALL_BRANCH, cq = Q - 1, ct = T - 1, ck = a.first_slot; bool cforce = false;
  int sp = 0;
  long guard = 0;
  while ((call != CALL_NONE || sp > 0) && status == 0) {
    if (++guard > (1L << 40)) { status = ALN_E_OVERFLOW; break; }
    flush();
    if (call == CALL_OPT) {
      call = CALL_NONE;
      int q0 = cq, t0 = ct; const int k0 = ck; const bool force = cforce;
      if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); continue; }           // cw.h:220-228
      walk(q0, t0, k0, force);
      if (cw) { call = CALL_BRANCH; cq = q0; ct = t0; ck = k0; cforce = force; }   // branch(pq,pt,k0,force)  cw.h:276
      else base_case(q0, t0, k0);                                                    // ucw.h:232-234
      continue;
    }
    if (call == CALL_BRANCH) {
      call = CALL_NONE;
      const int q0 = cq, t0 = ct, k0 = ck;
      if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); continue; }
      if (cw && cforce) { call = CALL_OPT; cforce = true; continue; }                          // cw.h:205-209
      if (n_as > a.user_limit) { call = CALL_OPT; cforce = true; continue; }                   // cw.h:127-140 / ucw.h:110-121
      if ((uint32_t)sp >= a.stack_cap) { status = ALN_E_OVERFLOW; break; }
      const uint32_t ch = ld_u(&a.head[k0]);
      const float cs = ld_f(&a.score[k0]);
      const float r = cs + dev_sim(e, q0, t0);
      if (lane == 0) {
        uint32_t* f = a.stack + (size_t)sp * kFrameWords;
        st_u(f + 0, (uint32_t)q0); st_u(f + 1, (uint32_t)t0); st_u(f + 2, (uint32_t)k0); st_u(f + 3, 0u);
        st_u(f + 4, ch); st_u(f + 5, __float_as_uint(cs)); st_u(f + 6, __float_as_uint(r));
      }
      dirty = true;
      ++sp;
      continue;
    }
    // ---- resume the branch frame on top of the stack: scan for the next accepted candidate ----
    uint32_t* f = a.stack + (size_t)(sp - 1) * kFrameWords;
    const int q0 = (int)ld_u(f + 0), t0 = (int)ld_u(f + 1), k0 = (int)ld_u(f + 2), cursor = (int)ld_u(f + 3);
    const uint32_t curr_head = ld_u(f + 4);
    const float curr_score = __uint_as_float(ld_u(f + 5)), r = __uint_as_float(ld_u(f + 6));
    const int k = (cursor == 0) ? k0 : (int)n_as;         // k = as.size() after every accepted candidate's subtree
    const int ndel = t0 - 2, nins = q0 - 2;
    const int ncand = 1 + ndel + nins;
    int found = -1; float fg = 0.f; int fq = 0, ft = 0;
    if (e.model == ALN_GAP_AFFINE_CONST) {
      // Constant affine gaps: the candidate's cell and its gap cost are pure arithmetic, so the scan is written without a single
      // branch around a load — 16 x 64 score loads are in flight per trip and one HBM round trip (~2 us) covers 1024 candidates.
      // (With loads inside the if/else ladder below the compiler has to wait for each one before the next group.)
      constexpr int kTrip = 16;
      const bool fdel = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_LOCAL_GLOBAL);
      const bool fins = (e.align_type == ALN_LOCAL || e.align_type == ALN_SEMI_LOCAL || e.align_type == ALN_GLOBAL_LOCAL);
      const uint16_t* H16p = reinterpret_cast<const uint16_t*>(Hbase) + pd.plane_off;
      const float* H32p = Hbase + pd.plane_off;
      for (int base = cursor; base < ncand && found < 0; base += 64 * kTrip) {
        int cq[kTrip], ct[kTrip]; float fsc[kTrip];
#pragma unroll
        for (int u = 0; u < kTrip; ++u) {
          const int idx = base + 64 * u + lane;
          const bool isdel = idx <= ndel;                       // idx 0 (match) has the same row
          int q = isdel ? q0 - 1 : q0 - 2 - (idx - ndel - 1);
          int t = idx == 0 ? t0 - 1 : isdel ? t0 - 1 - idx : t0 - 1;
          const bool in = idx < ncand;
          q = in ? q : q0 - 1; t = in ? t : t0 - 1;            // lanes past the end read a harmless cell
          cq[u] = q; ct[u] = t;
        }
        if (a.h_mode == 0) {
#pragma unroll
          for (int u = 0; u < kTrip; ++u) fsc[u] = H32p[(size_t)cq[u] * ld + ct[u]];
        } else {
#pragma unroll
          for (int u = 0; u < kTrip; ++u) fsc[u] = (float)H16p[(size_t)cq[u] * ld + ct[u]];
        }
#pragma unroll
        for (int u = 0; u < kTrip; ++u) {
          if (found >= 0 || base + 64 * u >= ncand) break;
          const int idx = base + 64 * u + lane;
          const int q = cq[u], t = ct[u];
          float g = 0.f;
          if (idx != 0) {
            if (idx <= ndel) {                                  // aasubalib.h:27-51
              const int len = t0 - t - 1;
              g = (len < 1 || (fdel && (t == 0 || t0 == T - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
            } else {                                            // aasubalib.h:53-77
              const int len = q0 - q - 1;
              g = (len < 1 || (fins && (q == 0 || q0 == Q - 1))) ? 0.f : e.gi + e.ge * (float)(len - 1);
            }
          }
          const bool ok = idx < ncand && (idx == 0 ? fsc[u] + r > thr : fsc[u] + r - g > thr);
          const unsigned long long m = __ballot(ok);
          if (m) {
            const int l = __builtin_ctzll(m);
            found = base + 64 * u + l;
            fg = __shfl(g, l); fq = __shfl(q, l); ft = __shfl(t, l);
          }
        }
      }
    } else {
    // 4 x 64 candidates per trip: the score loads of all four groups are in flight together, the ballots keep the order
      for (int base = cursor; base < ncand && found < 0; base += 256) {
        bool ok[4]; float g[4]; int pq[4], pt[4];
  #pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + 64 * u + lane;
          ok[u] = false; g[u] = 0.f; pq[u] = 0; pt[u] = 0;
          if (idx < ncand) {
            if (idx == 0) {                                     // match, cw.h:151-162
              pq[u] = q0 - 1; pt[u] = t0 - 1;
              const float fsc = HV(pq[u], pt[u]);
              ok[u] = fsc + r > thr;
            } else if (idx <= ndel) {                           // deletions i = t0-2 .. 1, cw.h:166-178
              pq[u] = q0 - 1; pt[u] = t0 - 1 - idx;
              const float fsc = HV(pq[u], pt[u]);
              g[u] = dev_deletion(e, pt[u], t0);
              ok[u] = fsc + r - g[u] > thr;
            } else {                                            // insertions j = q0-2 .. 1, cw.h:182-194
              pq[u] = q0 - 2 - (idx - ndel - 1); pt[u] = t0 - 1;
              const float fsc = HV(pq[u], pt[u]);
              g[u] = dev_insertion(e, pq[u], q0, pt[u], t0);
              ok[u] = fsc + r - g[u] > thr;
            }
          }
        }
  #pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long m = __ballot(ok[u]);
          if (m && found < 0) {
            const int l = __builtin_ctzll(m);
            found = base + 64 * u + l;
            fg = __shfl(g[u], l); fq = __shfl(pq[u], l); ft = __shfl(pt[u], l);
          }
        }
      }
  }
    if (found < 0) {
      --sp;                                                 // branch() returns
      if (cursor == 0) {                                    // k == k0: nothing passed, finish along stored pointers
        call = CALL_OPT; cq = q0; ct = t0; ck = k0; cforce = true;   // cw.h:196-203 / ucw.h:186-191
      }
      continue;
    }
    if ((uint32_t)k == n_as) {                              // as.push_back(curr)
      if (n_as >= a.ali_cap) { status = ALN_E_OVERFLOW; break; }
      if (lane == 0) { st_u(&a.head[n_as], curr_head); st_f(&a.score[n_as], curr_score); }
      dirty = true;
      ++n_as;
    }
    prepend(k, q0, t0);
    set_score(k, r - fg);
    if (lane == 0) st_u(f + 3, (uint32_t)(found + 1));
    dirty = true;
    if (cw) { call = CALL_OPT; cq = fq; ct = ft; ck = k; cforce = false; }      // opt_path(cand,k,false)
    else { call = CALL_BRANCH; cq = fq; ct = ft; ck = k; cforce = false; }      // ucw: branch(cand,k)
  }
  if (lane == 0) { a.out[0] = (int32_t)n_as; a.out[1] = (int32_t)n_nodes; a.out[2] = status; }
}

// unroll selected alignments: one wave per alignment walks the trie (list order = head -> end)
__global__ __launch_bounds__(64) void enum_unroll_kernel(const uint32_t* __restrict__ node_pair, const uint32_t* __restrict__ node_next,
                                                         const uint8_t* __restrict__ node_len,
                                                         const uint32_t* __restrict__ head, const int32_t* __restrict__ sel, int nsel,
                                                         int32_t* __restrict__ out_pairs, int32_t* __restrict__ out_n, int stride) {
  const int s = blockIdx.x;
  if (s >= nsel) return;
  const int n = unroll_alignment(node_pair, node_next, node_len, head[sel[s]], out_pairs + (size_t)s * stride * 2, stride);
  if (threadIdx.x == 0) out_n[s] = n;
}

}  // namespace aln

namespace aln {
__global__ void enumerate_ks_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto, const uint8_t* __restrict__ qcodes,
                                    const uint8_t* __restrict__ tcodes, const float* __restrict__ tgi, const float* __restrict__ tge,
                                    const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                    const float* __restrict__ Sbase, EnumArgs a);
__global__ void enumerate_par_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto, const uint8_t* __restrict__ qcodes,
                                     const uint8_t* __restrict__ tcodes, const float* __restrict__ tgi, const float* __restrict__ tge,
                                     const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                     const float* __restrict__ Sbase, EnumArgs a);
__global__ void enum_blockmax_kernel(const PairDesc* __restrict__ pairs, const int32_t* __restrict__ pair_list, int pair0,
                                     const float* __restrict__ Hbase, int h_mode, float* __restrict__ rowmax, float* __restrict__ colmax,
                                     int bm_rows, int bm_cols, int nbt, int nbq, int bm_pair0);
__global__ void enumerate_cr_kernel(const PairDesc* __restrict__ pairs, int pair, EvalDev proto, const uint8_t* __restrict__ qcodes,
                                    const uint8_t* __restrict__ tcodes, const float* __restrict__ tgi, const float* __restrict__ tge,
                                    const float* __restrict__ Hbase, const uint32_t* __restrict__ Pbase,
                                    const float* __restrict__ Sbase, EnumArgs a);
}

// ---------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace aln;

namespace {
struct SortKey {
  float score; int32_t idx;
  bool operator<(const SortKey& o) const { return score > o.score; }   // alignment.h:104-105 "higher score first"
};
// NOaliParams::user_limit when the caller leaves it 0: the enumerators' own hard-coded / default limits — cw.h:76 (1000000),
// ucw.h:72 (100000), kscw.h:172 via NOaliParams::default_user_limit (noalib.cpp:19-20: 100000).  One rule for both entry points.
uint32_t default_user_limit(const aln_noa* noa) {
  if (noa->user_limit) return noa->user_limit;
  return noa->kind == ALN_ENUM_CW ? 1000000u : 100000u;
}
bool pruned_kind(int kind) { return kind == ALN_ENUM_KSCW || kind == ALN_ENUM_CRCW; }
// CRCW's own parameters (NOaliParams::sort_limit 100, max_overlap 0.30: noalib.cpp:18,21); false: not usable
bool set_cr_params(EnumArgs& a, const aln_noa* noa, int maxT) {
  a.sort_limit = noa->sort_limit ? noa->sort_limit : 100u;
  a.max_overlap = noa->max_overlap;
  a.cr_tpad = (maxT + 7) & ~7;
  return a.sort_limit >= 1 && a.sort_limit <= 512 && a.k_limit >= 1;
}
size_t cr_lds_bytes(const EnumArgs& a) { return (size_t)a.cand_cap * 8 + (size_t)a.sort_limit * 9 * 4; }

// ---- the several-waves-per-pair search (enumerate_par.hip) -------------------------------------------------------------
// LDS of one workgroup: flags + template codes + query codes (16-byte padded) + the 32 x 32 table
size_t par_lds_bytes(int maxQ, int maxT, int waves = 0) {           // + KSCW's candidate arrays: 256 x (sum, index) per wave
  return (size_t)((maxT + 15) & ~15) * 2 + (size_t)((maxQ + 15) & ~15) + 4096 + (size_t)waves * 256 * 8;
}
// waves per pair: context hint "enum_waves" (1 = the one-wave kernel, 2..16), else 16; 0 = not usable
int par_waves(const aln_batch* b, int kind, int n_pairs) {
  if (kind != ALN_ENUM_CW && kind != ALN_ENUM_UCW && kind != ALN_ENUM_KSCW) return 0;
  const int h = b->ctx->hints.enum_waves;
  if (h == 1 || par_lds_bytes(b->maxQ, b->maxT, 16) > 58000 || b->maxQ > 65535 || b->maxT > 65535) return 0;
  if (h >= 2) return std::min(h, 16);
  (void)n_pairs;
  return 16;                      // measured on 1024 config-4 pairs: 16 waves 0.33 s, 8 waves 0.48 s, 4 waves 0.75 s, one wave 4.6 s
}
// Block maxima of pairs [pair0, pair0 + np) for the pruned candidate scan (constant affine gaps, sequences up to 4096): fills
// a.rowmax / a.colmax (caller frees) or leaves them null when the model does not qualify or the arrays would not pay.
int make_blockmax(aln_batch* b, EnumArgs& a, int pair0, int np, float** d_row, float** d_col) {
  *d_row = *d_col = nullptr;
  if (b->gapdev.model != ALN_GAP_AFFINE_CONST || b->maxQ > 4096 || b->maxT > 4096 || !(b->gapdev.gi >= 0.f) || !(b->gapdev.ge >= 0.f)) return ALN_OK;
  a.bm_rows = b->maxQ; a.bm_cols = b->maxT; a.nbt = (b->maxT + 63) / 64; a.nbq = (b->maxQ + 63) / 64; a.bm_pair0 = pair0;
  const size_t nr = (size_t)np * a.bm_rows * a.nbt, nc = (size_t)np * a.bm_cols * a.nbq;
  if ((nr + nc) * 4 > ((size_t)8 << 30)) return ALN_OK;
  aln_ctx* ctx = b->ctx;
  if (hipMalloc((void**)d_row, nr * 4) != hipSuccess || hipMalloc((void**)d_col, nc * 4) != hipSuccess) {
    hipFree(*d_row); hipFree(*d_col); *d_row = *d_col = nullptr;
    (void)hipGetLastError();
    return ALN_OK;                                           // no memory for the shortcut: scan everything
  }
  hipLaunchKernelGGL(enum_blockmax_kernel, dim3(a.nbq, np), dim3(256), 0, ctx->stream, b->d_pairs, (const int32_t*)nullptr, pair0, b->d_H,
                     b->h_mode, *d_row, *d_col, a.bm_rows, a.bm_cols, a.nbt, a.nbq, pair0);
  ALN_HIP_CHECK(ctx, hipGetLastError());
  a.rowmax = *d_row; a.colmax = *d_col;
  return ALN_OK;
}
// The reference's set order from the slot tree enumerate_par_kernel recorded (see its header): pre-order, siblings by
// (t0 of the branch node ascending, candidate index ascending).  info = 3 words per slot (parent, t0, candidate), valid for
// slots > first; slots <= first keep their places.  old_of_new[k] = slot that holds the set's k-th alignment.
void slot_order(int n_as, int first, const uint32_t* info, std::vector<int32_t>& old_of_new) {
  old_of_new.resize(n_as);
  for (int k = 0; k <= first && k < n_as; ++k) old_of_new[k] = k;
  if (n_as <= first + 1) return;
  const int m = n_as - first - 1;
  std::vector<std::pair<uint64_t, int32_t>> ch(m);
  for (int k = 0; k < m; ++k) {
    const uint32_t* si = info + (size_t)(first + 1 + k) * 3;
    ch[k].first = ((uint64_t)si[0] << 40) | ((uint64_t)(si[1] & 0xFFFFFu) << 20) | (uint64_t)(si[2] & 0xFFFFFu);
    ch[k].second = first + 1 + k;
  }
  std::sort(ch.begin(), ch.end());
  std::vector<int32_t> beg(n_as + 1, 0);                      // children of slot p: ch[beg[p] .. beg[p+1])
  for (int k = 0; k < m; ++k) ++beg[(ch[k].first >> 40) + 1];
  for (int p = 0; p < n_as; ++p) beg[p + 1] += beg[p];
  std::vector<std::pair<int32_t, int32_t>> st;               // (slot, next child)
  int counter = first;
  old_of_new[counter++] = first;
  st.emplace_back(first, beg[first]);
  while (!st.empty()) {
    auto& top = st.back();
    if (top.second == beg[top.first + 1]) { st.pop_back(); continue; }
    const int32_t c = ch[top.second++].second;
    old_of_new[counter++] = c;
    st.emplace_back(c, beg[c]);
  }
}
}  // namespace

extern "C" int aln_batch_enumerate(aln_batch* b, int32_t pair, const aln_noa* noa, const uint8_t* flags, aln_alignment* out,
                                   int32_t max_alignments, int32_t* pairs, int64_t pairs_capacity, int32_t* n_out) {
  if (!b || !noa || !out || !pairs || !n_out || pair < 0 || pair >= b->n_pairs) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub || b->direction != ALN_FWD) return ALN_E_STATE;
  const bool ks = pruned_kind(noa->kind);      // the pruned enumerators share the frame layout and the uid array
  const bool cr = noa->kind == ALN_ENUM_CRCW;
  if ((noa->kind == ALN_ENUM_CW || ks) && !flags) return ALN_E_ARG;
  if (noa->kind != ALN_ENUM_CW && noa->kind != ALN_ENUM_UCW && !ks) return ALN_E_ARG;
  aln_ctx* ctx = b->ctx;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const PairDesc& d = b->h_pairs[pair];
  // n_existing < 0: the set starts with the pair's Optimal alignment, like the drivers (aa_ali.cpp:83);
  // otherwise the caller's set already holds n_existing alignments whose scores take part in sortSet.
  const bool seed_opt = noa->n_existing < 0;
  const int n_ex = seed_opt ? 1 : noa->n_existing;
  if (!seed_opt && n_ex > 0 && !noa->existing_scores) return ALN_E_ARG;
  std::vector<PairResult> res(b->n_pairs);
  std::vector<int32_t> optpath((size_t)b->path_stride * 2);
  if (seed_opt) {
    int rc = launch_traceback(b, false);
    if (rc) return rc;
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(res.data(), b->d_res, sizeof(PairResult) * b->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(optpath.data(), b->d_path + (size_t)pair * b->path_stride * 2, optpath.size() * 4,
                                      hipMemcpyDeviceToHost, ctx->stream));
    ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (res[pair].status != 0) return res[pair].status;
  }

  const uint32_t user_limit = default_user_limit(noa);
  EnumArgs a = {};
  a.kind = noa->kind;
  a.user_limit = user_limit;
  a.delta_ratio = noa->delta_ratio;
  a.first_slot = n_ex;
  a.k_limit = ks ? (noa->k_limit ? noa->k_limit : 16u) : 0u;
  if (a.k_limit > 64u) return ALN_E_ARG;
  a.cand_cap = (uint32_t)(d.Q + d.T);
  a.ali_cap = user_limit + 65536u + (uint32_t)n_ex;
  a.node_cap = 48u << 20;
  if (ctx->hints.enum_node_cap > 0) a.node_cap = (uint32_t)ctx->hints.enum_node_cap;
  const int pw = par_waves(b, noa->kind, 1);         // cw / ucw: several waves search the pair (enumerate_par.hip)
  if (pw) a.node_cap = std::max(a.node_cap, kChunkNodes) / kChunkNodes * kChunkNodes;   // its node pool comes in chunks
  a.stack_cap = (uint32_t)(d.Q + d.T + 8);
  if (cr && !set_cr_params(a, noa, d.T)) return ALN_E_ARG;
  uint8_t* d_flags = nullptr; int32_t* d_out = nullptr;
  auto cleanup = [&]() {
    hipFree(a.node_pair); hipFree(a.node_next); hipFree(a.head); hipFree(a.score); hipFree(a.stack); hipFree(a.uid);
    hipFree(a.cr_ali); hipFree(a.cr_reg); hipFree(a.task); hipFree(a.slot_info); hipFree(a.chunk_next); hipFree(a.node_len);
    hipFree((void*)a.rowmax); hipFree((void*)a.colmax);
    hipFree(d_flags); hipFree(d_out);
  };
#define ETRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return ALN_E_HIP; } } while (0)
  ETRY(hipMalloc((void**)&a.node_pair, (size_t)a.node_cap * 4));
  ETRY(hipMalloc((void**)&a.node_next, (size_t)a.node_cap * 4));
  ETRY(hipMalloc((void**)&a.head, (size_t)a.ali_cap * 4));
  ETRY(hipMalloc((void**)&a.score, (size_t)a.ali_cap * 4));
  ETRY(hipMalloc((void**)&a.stack, (size_t)a.stack_cap * (ks ? 8 + 4 * a.k_limit : kFrameWords) * 4));
  if (ks) ETRY(hipMalloc((void**)&a.uid, (size_t)a.ali_cap * 4));
  if (pw) {
    ETRY(hipMalloc((void**)&a.task, (size_t)a.ali_cap * kTaskWords * 4));
    ETRY(hipMemsetAsync(a.task, 0, (size_t)a.ali_cap * kTaskWords * 4, ctx->stream));      // ready words: ticket + 1, never 0
    ETRY(hipMalloc((void**)&a.slot_info, (size_t)a.ali_cap * 12));
    ETRY(hipMalloc((void**)&a.node_len, (size_t)a.node_cap));
    a.n_chunks = a.node_cap / kChunkNodes;              // (node_cap was rounded to chunks above)
    a.n_pools = 1;
    ETRY(hipMalloc((void**)&a.chunk_next, 4));
    ETRY(hipMemsetAsync(a.chunk_next, 0, 4, ctx->stream));
  }
  if (cr) {
    ETRY(hipMalloc((void**)&a.cr_ali, (size_t)a.sort_limit * a.cr_tpad * 2));
    ETRY(hipMalloc((void**)&a.cr_reg, (size_t)a.cr_tpad * 4));
  }
  ETRY(hipMalloc((void**)&d_flags, (size_t)d.T));
  ETRY(hipMalloc((void**)&d_out, 16));
  if (flags) ETRY(hipMemcpyAsync(d_flags, flags, (size_t)d.T, hipMemcpyHostToDevice, ctx->stream));
  else ETRY(hipMemsetAsync(d_flags, 1, (size_t)d.T, ctx->stream));
  a.flags = d_flags;
  a.ptr_mode = b->ptr_mode;
  a.h_mode = b->h_mode;
  a.int_sums = b->ptr_mode != 0;
  a.out = d_out;
  float *d_bmr = nullptr, *d_bmc = nullptr;
  if (pw || noa->kind == ALN_ENUM_KSCW) { int rcb = make_blockmax(b, a, pair, 1, &d_bmr, &d_bmc); if (rcb) { cleanup(); return rcb; } }

  EvalDev proto = {};
  proto.model = b->gapdev.model; proto.align_type = b->gapdev.align_type;
  proto.gi = b->gapdev.gi; proto.ge = b->gapdev.ge;
  proto.sim_kind = (b->sim_kind == ALN_SIM_SUBMATRIX) ? ALN_SIM_SUBMATRIX : ALN_SIM_MATRIX;
  proto.tablef = b->d_tablef;
  proto.tcn = b->d_tcn; proto.deltab = b->d_deltab; proto.deltab_off = b->d_deltab_off; proto.instab = b->d_instab;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model != ALN_GAP_AFFINE_CONST;
  auto launch_one_wave = [&]() {
    if (cr)
      hipLaunchKernelGGL(enumerate_cr_kernel, dim3(1), dim3(64), cr_lds_bytes(a), ctx->stream, b->d_pairs, pair, proto,
                         sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                         b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
    else if (ks)
      hipLaunchKernelGGL(enumerate_ks_kernel, dim3(1), dim3(64), (size_t)a.cand_cap * 8, ctx->stream, b->d_pairs, pair, proto,
                         sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                         b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
    else
      hipLaunchKernelGGL(enumerate_kernel, dim3(1), dim3(64), 0, ctx->stream, b->d_pairs, pair, proto, sub ? b->d_qcodes : nullptr,
                         sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr, b->d_H, b->d_P,
                         sub ? nullptr : b->d_S, a);
  };
  if (pw)
    hipLaunchKernelGGL(enumerate_par_kernel, dim3(1), dim3(64 * pw), par_lds_bytes(b->maxQ, b->maxT, pw), ctx->stream, b->d_pairs, pair, proto,
                       sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                       b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
  else
    launch_one_wave();
  ETRY(hipGetLastError());
  int32_t hout[4] = {0, 0, 0, 0};
  ETRY(hipMemcpyAsync(hout, d_out, 12, hipMemcpyDeviceToHost, ctx->stream));
  ETRY(hipStreamSynchronize(ctx->stream));
  bool used_par = pw != 0;
  if (used_par && hout[2] == kParSerial) {            // the set outgrows user_limit: the serial order decides what is cut
    used_par = false;
    launch_one_wave();
    ETRY(hipGetLastError());
    ETRY(hipMemcpyAsync(hout, d_out, 12, hipMemcpyDeviceToHost, ctx->stream));
    ETRY(hipStreamSynchronize(ctx->stream));
  }
  if (hout[2] != 0) { cleanup(); return hout[2]; }
  const int n_as = hout[0];
  std::vector<float> scores(n_as);
  std::vector<int32_t> old_of_new;                    // set position -> slot (identity after the one-wave kernel)
  if (used_par) {
    std::vector<uint32_t> info((size_t)n_as * 3);
    std::vector<float> raw(n_as);
    ETRY(hipMemcpyAsync(info.data(), a.slot_info, info.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    ETRY(hipMemcpyAsync(raw.data() + n_ex, a.score + n_ex, (size_t)(n_as - n_ex) * 4, hipMemcpyDeviceToHost, ctx->stream));
    ETRY(hipStreamSynchronize(ctx->stream));
    slot_order(n_as, n_ex, info.data(), old_of_new);
    for (int k = n_ex; k < n_as; ++k) scores[k] = raw[old_of_new[k]];
  } else {
    ETRY(hipMemcpyAsync(scores.data() + n_ex, a.score + n_ex, (size_t)(n_as - n_ex) * 4, hipMemcpyDeviceToHost, ctx->stream));
    ETRY(hipStreamSynchronize(ctx->stream));
  }
  if (seed_opt) scores[0] = b->islocal ? res[pair].best : res[pair].corner;
  else for (int k = 0; k < n_ex; ++k) scores[k] = noa->existing_scores[k];
  std::vector<int32_t> uids;
  if (ks && used_par) {                               // uid = the alignment's place at creation (kscw.h:262), the seed's is 1 (:121)
    uids.assign(n_as, -1);
    for (int k = n_ex; k < n_as; ++k) uids[k] = k == n_ex ? 1 : k;
  } else if (ks) {
    uids.assign(n_as, -1);
    ETRY(hipMemcpy(uids.data() + n_ex, a.uid + n_ex, (size_t)(n_as - n_ex) * 4, hipMemcpyDeviceToHost));
  }

  // AlignmentSet::sortSet(number_suboptimal) — alignment.h:922-932, same libstdc++ calls on the same order of keys
  std::vector<SortKey> keys(n_as);
  for (int k = 0; k < n_as; ++k) { keys[k].score = scores[k]; keys[k].idx = k; }
  const int mx = noa->number_suboptimal;
  if (mx >= n_as) std::sort(keys.begin(), keys.end());
  else if (mx > 0) { std::partial_sort(keys.begin(), keys.begin() + mx, keys.end()); keys.erase(keys.begin() + mx, keys.end()); }
  const int n_keep = (int)keys.size();
  *n_out = n_keep;
  if (n_keep > max_alignments) { cleanup(); return ALN_E_OVERFLOW; }

  // unroll the survivors on the device
  std::vector<int32_t> sel;
  for (int k = 0; k < n_keep; ++k) if (keys[k].idx >= n_ex) sel.push_back(used_par ? old_of_new[keys[k].idx] : keys[k].idx);
  const int stride = b->path_stride;
  std::vector<int32_t> lists((size_t)std::max<size_t>(sel.size(), 1) * stride * 2), lens(std::max<size_t>(sel.size(), 1));
  if (!sel.empty()) {
    int32_t *d_sel = nullptr, *d_lists = nullptr, *d_lens = nullptr;
    hipError_t e1 = hipMalloc((void**)&d_sel, sel.size() * 4);
    hipError_t e2 = hipMalloc((void**)&d_lists, lists.size() * 4);
    hipError_t e3 = hipMalloc((void**)&d_lens, sel.size() * 4);
    bool okk = e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess;
    if (okk) okk = hipMemcpyAsync(d_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    if (okk) {
      hipLaunchKernelGGL(enum_unroll_kernel, dim3((unsigned)sel.size()), dim3(64), 0, ctx->stream, a.node_pair, a.node_next,
                         used_par ? a.node_len : (const uint8_t*)nullptr, a.head,
                         d_sel, (int)sel.size(), d_lists, d_lens, stride);
      okk = hipGetLastError() == hipSuccess;
    }
    if (okk) okk = hipMemcpyAsync(lists.data(), d_lists, lists.size() * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    if (okk) okk = hipMemcpyAsync(lens.data(), d_lens, sel.size() * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    if (okk) okk = hipStreamSynchronize(ctx->stream) == hipSuccess;
    hipFree(d_sel); hipFree(d_lists); hipFree(d_lens);
    if (!okk) { ctx->last_error = "enumeration unroll failed"; cleanup(); return ALN_E_HIP; }
  }
  cleanup();
#undef ETRY
  // assemble outputs in set order
  const std::string qs(b->q_res.data() + d.q_off, d.Q), ts(b->t_res.data() + d.t_off, d.T);
  int64_t off = 0;
  size_t si = 0;
  for (int k = 0; k < n_keep; ++k) {
    const int idx = keys[k].idx;
    const int32_t* src; int len;
    std::vector<int32_t> tmp;
    if (idx < n_ex && !seed_opt) {        // one of the caller's own alignments: only its new position is reported
      out[k].score = keys[k].score; out[k].identity = 0.f; out[k].uid = -1; out[k].n_pairs = -1; out[k].pair_off = idx;
      continue;
    }
    if (idx < n_ex) {
      len = res[pair].n_path;
      tmp.resize((size_t)len * 2);
      for (int i = 0; i < len; ++i) { tmp[2 * i] = optpath[2 * (len - 1 - i)]; tmp[2 * i + 1] = optpath[2 * (len - 1 - i) + 1]; }
      src = tmp.data();
    } else {
      len = lens[si];
      src = lists.data() + si * (size_t)stride * 2;
      ++si;
      if (len < 0) return ALN_E_OVERFLOW;
    }
    if (off + len > pairs_capacity) return ALN_E_OVERFLOW;
    memcpy(pairs + 2 * off, src, (size_t)len * 8);
    out[k].score = keys[k].score;
    out[k].uid = (idx < n_ex) ? -1 : ks ? uids[idx] : (noa->kind == ALN_ENUM_CW ? 0 : -1);   // cw.h:83 sets uid 0 on its seed; copies inherit it
    out[k].n_pairs = len;
    out[k].pair_off = off;
    out[k].identity = aln_identity(qs.c_str(), d.Q, ts.c_str(), d.T, pairs + 2 * off, len);
    off += len;
  }
  return ALN_OK;
}


// ---- batched form (BASELINE config 4): every pair of the batch in one launch ----------------------------------
namespace aln {
// one wave per (pair, slot): slot's alignment index comes from sel[]; -1 = empty, 0 = the pair's Optimal alignment
// (taken from the traceback list, which is stored end -> start), otherwise a trie walk.
__global__ __launch_bounds__(64) void enum_unroll_all_kernel(const uint32_t* __restrict__ node_pair, const uint32_t* __restrict__ node_next,
                                                             const uint8_t* __restrict__ node_len,
                                                             const uint32_t* __restrict__ head, size_t node_stride, int n_pools, uint32_t ali_cap,
                                                             const int32_t* __restrict__ sel, int K, const int32_t* __restrict__ pair_list,
                                                             const int32_t* __restrict__ path,
                                                             int path_stride, const PairResult* __restrict__ res,
                                                             int32_t* __restrict__ out_pairs, int32_t* __restrict__ out_n, int stride,
                                                             int out_by_pair) {
  // blockIdx.y = position of the pair inside its group (pools and sel are group-local); gp = its index in the batch.  The outputs
  // are group-local too, unless the group is the whole batch (out_by_pair): then they are laid out in batch order, whatever
  // order the pairs were launched in, and the host takes them with one copy.
  const int p = blockIdx.y, k = blockIdx.x;
  const int gp = pair_list[p];
  const int idx = sel[(size_t)p * K + k];
  const size_t op = out_by_pair ? (size_t)gp : (size_t)p;
  int32_t* o = out_pairs ? out_pairs + (op * K + k) * stride * 2 : nullptr;
  if (idx < 0) { if (threadIdx.x == 0) out_n[op * K + k] = 0; return; }
  if (idx == 0) {
    const int n = res[gp].n_path;
    const int32_t* src = path + (size_t)gp * path_stride * 2;
    if (o) for (int i = threadIdx.x; i < n && i < stride; i += 64) { o[2 * i] = src[2 * (n - 1 - i)]; o[2 * i + 1] = src[2 * (n - 1 - i) + 1]; }
    if (threadIdx.x == 0) out_n[op * K + k] = n;
    return;
  }
  // a slice per pair (one-wave kernels: n_pools = 0) or the pool workgroup p of enumerate_par.hip used (enum_pool_of)
  const size_t nbase = n_pools ? (size_t)enum_pool_of((uint32_t)p, (uint32_t)n_pools) * node_stride : (size_t)p * node_stride;
  const int n = unroll_alignment(node_pair + nbase, node_next + nbase, node_len ? node_len + nbase : nullptr,
                                 head[(size_t)p * ali_cap + idx], o, stride);
  if (threadIdx.x == 0) out_n[op * K + k] = n;
}
}  // namespace aln

namespace aln {
// {score, parent slot, t0, candidate} of every slot of every finished set, packed in set order of the launch: one D2H copy
__global__ __launch_bounds__(256) void enum_pack_sets_kernel(const float* __restrict__ score, const uint32_t* __restrict__ slot_info,
                                                            uint32_t ali_cap, const int64_t* __restrict__ off, uint32_t* __restrict__ out) {
  const int g = blockIdx.x;
  const int64_t o = off[g];
  const int n = (int)(off[g + 1] - o);
  for (int k = threadIdx.x; k < n; k += 256) {
    uint32_t* r = out + (size_t)(o + k) * 4;
    r[0] = __float_as_uint(score[(size_t)g * ali_cap + k]);
    if (slot_info) { const uint32_t* si = slot_info + ((size_t)g * ali_cap + k) * 3; r[1] = si[0]; r[2] = si[1]; r[3] = si[2]; }
    else { r[1] = 0u; r[2] = 0u; r[3] = 0u; }
  }
}
}  // namespace aln

extern "C" int aln_batch_enumerate_all(aln_batch* b, const aln_noa* noa, const uint8_t* flags, int32_t flags_stride,
                                       uint32_t node_cap_per_pair, uint32_t ali_cap_per_pair, int32_t K, int32_t* n_out, float* scores,
                                       int32_t* lengths, int32_t* pairs, int32_t pair_stride, int32_t* status) {
  if (!b || !noa || !n_out || !scores || !lengths || !status || K <= 0) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub || b->direction != ALN_FWD) return ALN_E_STATE;
  const bool ks = pruned_kind(noa->kind);
  const bool cr = noa->kind == ALN_ENUM_CRCW;
  if ((noa->kind == ALN_ENUM_CW || ks) && !flags) return ALN_E_ARG;
  if (noa->kind != ALN_ENUM_CW && noa->kind != ALN_ENUM_UCW && !ks) return ALN_E_ARG;
  if (pairs && pair_stride < b->path_stride) return ALN_E_ARG;
  aln_ctx* ctx = b->ctx;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int n = b->n_pairs;
  if (n == 0) return ALN_OK;
  // every set starts with the pair's Optimal alignment (aa_ali.cpp:83)
  int rc = launch_traceback(b, false);
  if (rc) return rc;
  EnumArgs a0 = {};
  a0.kind = noa->kind; a0.user_limit = default_user_limit(noa); a0.delta_ratio = noa->delta_ratio; a0.first_slot = 1;
  a0.k_limit = ks ? (noa->k_limit ? noa->k_limit : 16u) : 0u;
  if (a0.k_limit > 64u) return ALN_E_ARG;
  a0.cand_cap = (uint32_t)(b->maxQ + b->maxT);
  a0.stack_cap = (uint32_t)(b->maxQ + b->maxT + 8);
  a0.ptr_mode = b->ptr_mode; a0.h_mode = b->h_mode;
  a0.int_sums = b->ptr_mode != 0;                      // tagged planes exist only for integer tables and gaps with bounded scores
  a0.flags_stride = flags ? flags_stride : 0;
  if (cr && !set_cr_params(a0, noa, b->maxT)) return ALN_E_ARG;
  const size_t frame_words = ks ? 8 + 4 * a0.k_limit : kFrameWords;

  uint8_t* d_flags = nullptr;
  float *d_bmr = nullptr, *d_bmc = nullptr;
  hipEvent_t evs[4] = {nullptr, nullptr, nullptr, nullptr};
  // buffers of one group of pairs (see below)
  EnumArgs a = a0;
  int32_t *d_out = nullptr, *d_sel = nullptr, *d_lists = nullptr, *d_lens = nullptr, *d_list = nullptr;
  auto free_group = [&]() {
    // (node_pair, node_next, head, score, task, slot_info, d_lists are the batch's: b->enum_scratch)
    hipFree(a.stack); hipFree(a.uid);
    hipFree(a.cr_ali); hipFree(a.cr_reg); hipFree(a.chunk_next);
    hipFree(d_out); hipFree(d_sel); hipFree(d_lens); hipFree(d_list);
    a.node_pair = a.node_next = a.head = nullptr; a.score = nullptr; a.stack = nullptr; a.uid = nullptr; a.cr_ali = nullptr; a.cr_reg = nullptr;
    a.task = a.slot_info = a.chunk_next = nullptr; a.node_len = nullptr;
    d_out = d_sel = d_lists = d_lens = d_list = nullptr;
  };
  auto cleanup = [&]() {
    free_group();
    hipFree(d_flags); hipFree(d_bmr); hipFree(d_bmc);
    if (!ctx->hints.enum_keep_pools) for (auto& sc : b->enum_scratch) { hipFree(sc.p); sc.p = nullptr; sc.bytes = 0; }
    for (auto ev : evs) if (ev) hipEventDestroy(ev);
  };
#define BTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return ALN_E_HIP; } } while (0)
  const size_t fl_bytes = flags ? (flags_stride ? (size_t)n * flags_stride : (size_t)b->maxT) : (size_t)b->maxT;
  BTRY(hipMalloc((void**)&d_flags, fl_bytes));
  if (flags) BTRY(hipMemcpyAsync(d_flags, flags, fl_bytes, hipMemcpyHostToDevice, ctx->stream));
  else BTRY(hipMemsetAsync(d_flags, 1, fl_bytes, ctx->stream));
  EvalDev proto = {};
  proto.model = b->gapdev.model; proto.align_type = b->gapdev.align_type;
  proto.gi = b->gapdev.gi; proto.ge = b->gapdev.ge;
  proto.sim_kind = (b->sim_kind == ALN_SIM_SUBMATRIX) ? ALN_SIM_SUBMATRIX : ALN_SIM_MATRIX;
  proto.tablef = b->d_tablef;
  proto.tcn = b->d_tcn; proto.deltab = b->d_deltab; proto.deltab_off = b->d_deltab_off; proto.instab = b->d_instab;
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  const bool tpos = b->gapdev.model != ALN_GAP_AFFINE_CONST;
  for (auto& ev : evs) BTRY(hipEventCreate(&ev));
  std::vector<PairResult> res(n);
  BTRY(hipMemcpyAsync(res.data(), b->d_res, sizeof(PairResult) * n, hipMemcpyDeviceToHost, ctx->stream));
  BTRY(hipStreamSynchronize(ctx->stream));
  // what the previous search of this batch used, if there was one: the pairs are launched heaviest first (below)
  std::vector<int32_t> prior_nodes;
  if ((int)b->enum_usage.size() == 4 * n && ctx->hints.enum_heavy_first) {
    prior_nodes.resize(n);
    for (int p = 0; p < n; ++p) prior_nodes[p] = b->enum_usage[(size_t)p * 4 + 1];
  }
  b->enum_usage.assign((size_t)n * 4, 0);
  b->enum_search_ms = b->enum_unroll_ms = 0.f;
  if (par_waves(b, noa->kind, n) || noa->kind == ALN_ENUM_KSCW) {   // block maxima of every pair's score plane, once (the pruned scans of enumerate_par.hip / enumerate_ks.hip)
    int rcb = make_blockmax(b, a0, 0, n, &d_bmr, &d_bmc);
    if (rcb) { cleanup(); return rcb; }
  }
  for (int p = 0; p < n; ++p) { status[p] = 0; n_out[p] = 0; }

  // The search runs for GROUPS of pairs, one workgroup per pair, each with its own slice of the pools.  Round 0 is every pair
  // with the caller's capacities.  A near-optimal search can need anything between nothing (an unrelated pair) and tens of
  // millions of trie nodes (a 2000-residue homolog at DELTA_RATIO 0.01: 2-21 M), so pairs whose node or alignment pool
  // overflowed are searched again, in groups sized to a device-memory budget, with four times the capacity — up to
  // enum_pool_retries times (context hint, default 2: 16 x the caller's capacities).
  const size_t kPoolBudget = (size_t)48 << 30;                       // bytes of pools of one group
  std::vector<int32_t> todo((size_t)n);
  for (int p = 0; p < n; ++p) todo[p] = p;
  // Workgroup g of a launch searches pair todo[g], and workgroups start in index order as residency frees up.  A search's length is
  // unknown beforehand (nothing, or 27 M trie nodes), and the longest pair bounds a launch — unless it starts first.  When this
  // batch has been searched before (refinement rounds, a sweep over DELTA_RATIO, bench.py's second call), the pairs go in the
  // order of what they used then, heaviest first; results are stored per pair and do not depend on the order.
  if (!prior_nodes.empty())
    std::stable_sort(todo.begin(), todo.end(), [&](int32_t x, int32_t y) { return prior_nodes[x] > prior_nodes[y]; });
  else if (ctx->hints.enum_heavy_first) {
    // no history yet: the alignment's score is the cheapest hint there is — the threshold window (1 - DELTA_RATIO) * score widens
    // with it, and related sequences (long, high-scoring paths) are the ones that branch
    auto key = [&](int32_t p) { return b->islocal ? res[p].best : res[p].corner; };
    std::stable_sort(todo.begin(), todo.end(), [&](int32_t x, int32_t y) { return key(x) > key(y); });
  }
  uint32_t node_cap = node_cap_per_pair ? node_cap_per_pair : (1u << 20);
  uint32_t ali_cap = ali_cap_per_pair ? ali_cap_per_pair : 65536u;
  std::vector<int32_t> hout, sel, hlens, hlists, old_of_new;
  std::vector<uint32_t> info;
  std::vector<float> sc, raw;
  const int max_round = std::max(0, std::min(ctx->hints.enum_pool_retries, 3));
  // The alignment pool never has to be larger than what the reference's own brake allows: once the set is larger than user_limit every
  // branch is forced down the optimal path (cw.h:127-140), and only the candidates of the frames still on the stack are added —
  // fewer than (Q + T) per frame.  A pair that overflowed its alignment slots is therefore retried until its pool has that size
  // (beyond the enum_pool_retries rounds if need be), and then returns the user_limit-truncated set instead of ALN_E_OVERFLOW.
  const uint64_t limit_cap64 = (uint64_t)a0.user_limit + 65536ull + (uint64_t)(b->maxQ + b->maxT) * 16ull;
  const uint32_t limit_cap = (uint32_t)std::min<uint64_t>(limit_cap64, 0x7FFFFFFFull);
  std::vector<int32_t> again, serial_todo;
  // one group of pairs: search, sortSet, unroll.  use_par: the several-waves-per-pair kernel (cw / ucw); a pair whose set outgrows
  // user_limit comes back as kParSerial and is searched again, with the same capacities, by the one-wave kernel.
  // the large pools live with the batch (slot k of b->enum_scratch grows to the largest request)
  auto scratch = [&](int k, size_t bytes, void** out) -> hipError_t {
    aln_batch::Scratch& sc = b->enum_scratch[k];
    if (sc.bytes < bytes) {
      hipFree(sc.p); sc.p = nullptr; sc.bytes = 0;
      hipError_t e = hipMalloc(&sc.p, bytes);
      if (e != hipSuccess) return e;
      sc.bytes = bytes;
    }
    *out = sc.p;
    return hipSuccess;
  };
  auto run_group = [&](const int32_t* ids, int gn, bool last_round, int pw) -> int {
    double tm[6] = {0, 0, 0, 0, 0, 0};                       // enum_debug: host seconds of the group's phases
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    tm[0] = now();
    a = a0;
    a.node_cap = node_cap; a.ali_cap = ali_cap;
    // node pool: a slice per pair for the one-wave kernels; ONE pool handed out in chunks for the several-wave kernel (a pair
    // that needs ten times the average takes it from the pairs that need a tenth)
    size_t pool_nodes = (size_t)gn * a.node_cap;
    if (pw) {
      a.n_pools = (uint32_t)((pool_nodes + 0xFFFF0000u - 1) / 0xFFFF0000u);          // 32-bit node indices: < 2^32 nodes per pool
      a.n_chunks = (uint32_t)std::max<size_t>(1, pool_nodes / a.n_pools / kChunkNodes);
      pool_nodes = (size_t)a.n_pools * a.n_chunks * kChunkNodes;
      BTRY(hipMalloc((void**)&a.chunk_next, 4 * (size_t)a.n_pools));
      BTRY(hipMemsetAsync(a.chunk_next, 0, 4 * (size_t)a.n_pools, ctx->stream));
    }
    BTRY(scratch(0, pool_nodes * 4, (void**)&a.node_pair));
    BTRY(scratch(1, pool_nodes * 4, (void**)&a.node_next));
    if (pw) BTRY(scratch(8, pool_nodes, (void**)&a.node_len));
    BTRY(scratch(2, (size_t)gn * a.ali_cap * 4, (void**)&a.head));
    BTRY(scratch(3, (size_t)gn * a.ali_cap * 4, (void**)&a.score));
    if (pw) {
      BTRY(scratch(4, (size_t)gn * a.ali_cap * kTaskWords * 4, (void**)&a.task));
      BTRY(hipMemsetAsync(a.task, 0, (size_t)gn * a.ali_cap * kTaskWords * 4, ctx->stream));   // ready words: ticket + 1, never 0
      BTRY(scratch(5, (size_t)gn * a.ali_cap * 12, (void**)&a.slot_info));
    } else {
      BTRY(hipMalloc((void**)&a.stack, (size_t)gn * a.stack_cap * frame_words * 4));
    }
    if (ks) BTRY(hipMalloc((void**)&a.uid, (size_t)gn * a.ali_cap * 4));
    if (cr) {
      BTRY(hipMalloc((void**)&a.cr_ali, (size_t)gn * a.sort_limit * a.cr_tpad * 2));
      BTRY(hipMalloc((void**)&a.cr_reg, (size_t)gn * a.cr_tpad * 4));
    }
    BTRY(hipMalloc((void**)&d_out, (size_t)gn * 16));
    BTRY(hipMalloc((void**)&d_list, (size_t)gn * 4));
    BTRY(hipMemcpyAsync(d_list, ids, (size_t)gn * 4, hipMemcpyHostToDevice, ctx->stream));
    a.flags = d_flags; a.out = d_out; a.pair_list = d_list;
    tm[1] = now();
    BTRY(hipEventRecord(evs[0], ctx->stream));
    if (cr)
      hipLaunchKernelGGL(enumerate_cr_kernel, dim3(gn), dim3(64), cr_lds_bytes(a), ctx->stream, b->d_pairs, 0, proto,
                         sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                         b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
    else if (pw)
      hipLaunchKernelGGL(enumerate_par_kernel, dim3(gn), dim3(64 * pw), par_lds_bytes(b->maxQ, b->maxT, pw), ctx->stream, b->d_pairs, 0, proto,
                         sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                         b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
    else if (ks)
      hipLaunchKernelGGL(enumerate_ks_kernel, dim3(gn), dim3(64), (size_t)a.cand_cap * 8, ctx->stream, b->d_pairs, 0, proto,
                         sub ? b->d_qcodes : nullptr, sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr,
                         b->d_H, b->d_P, sub ? nullptr : b->d_S, a);
    else
      hipLaunchKernelGGL(enumerate_kernel, dim3(gn), dim3(64), 0, ctx->stream, b->d_pairs, 0, proto, sub ? b->d_qcodes : nullptr,
                         sub ? b->d_tcodes : nullptr, tpos ? b->d_tgi : nullptr, tpos ? b->d_tge : nullptr, b->d_H, b->d_P,
                         sub ? nullptr : b->d_S, a);
    BTRY(hipGetLastError());
    BTRY(hipEventRecord(evs[1], ctx->stream));
    hout.resize((size_t)gn * 4);
    BTRY(hipMemcpyAsync(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    BTRY(hipStreamSynchronize(ctx->stream));
    // per pair: sortSet on (score, index) keys, pick the survivors
    tm[2] = now();
    sel.assign((size_t)gn * K, -1);
    std::vector<char> deferred(gn, 0);
    std::vector<int64_t> off(gn + 1, 0);                    // the sets' slots, packed: pair g's slot k is record off[g] + k
    for (int g = 0; g < gn; ++g) {
      const int p = ids[g];
      for (int w = 0; w < 4; ++w) b->enum_usage[(size_t)p * 4 + w] = hout[4 * g + w];
      status[p] = hout[4 * g + 2] ? hout[4 * g + 2] : res[p].status;
      n_out[p] = 0;
      off[g + 1] = off[g];
      if (status[p] == kParSerial) { status[p] = 0; serial_todo.push_back(p); deferred[g] = 1; continue; }
      if (status[p] == ALN_E_OVERFLOW && !last_round &&
          (pw || (uint32_t)hout[4 * g + 1] >= a.node_cap - 64u || (uint32_t)hout[4 * g] >= a.ali_cap)) { again.push_back(p); deferred[g] = 1; continue; }
      if (status[p] != 0) { deferred[g] = 2; continue; }   // (2: nothing to order, lengths still reported)
      off[g + 1] = off[g] + hout[4 * g];
    }
    // one gather + one copy instead of two small copies per pair: {score, parent, t0, candidate} per slot
    const size_t total_slots = (size_t)off[gn];
    std::vector<uint32_t> packed(total_slots * 4);
    if (total_slots) {
      uint32_t* d_packed = nullptr; int64_t* d_off = nullptr;
      BTRY(scratch(7, total_slots * 16, (void**)&d_packed));
      BTRY(hipMalloc((void**)&d_off, (size_t)(gn + 1) * 8));
      hipError_t e1 = hipMemcpyAsync(d_off, off.data(), (size_t)(gn + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
      if (e1 == hipSuccess) {
        hipLaunchKernelGGL(enum_pack_sets_kernel, dim3(gn), dim3(256), 0, ctx->stream, a.score, pw ? a.slot_info : (const uint32_t*)nullptr, a.ali_cap, d_off, d_packed);
        e1 = hipGetLastError();
      }
      if (e1 == hipSuccess) e1 = hipMemcpyAsync(packed.data(), d_packed, total_slots * 16, hipMemcpyDeviceToHost, ctx->stream);
      if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
      hipFree(d_off);
      BTRY(e1);
    }
    // per pair, on host threads: the reference's set order (slot tree), then AlignmentSet::sortSet on (score, index) keys
    {
      const int mx = noa->number_suboptimal;
      auto work = [&](int tid, int nthreads) {
        std::vector<int32_t> order; std::vector<float> scv; std::vector<SortKey> keys; std::vector<uint32_t> inf;
        for (int g = tid; g < gn; g += nthreads) {
          if (deferred[g]) continue;
          const int p = ids[g];
          const int n_as = hout[4 * g];
          const uint32_t* rec = packed.data() + (size_t)off[g] * 4;
          scv.resize(n_as);
          if (pw) {                                           // set order from the slot tree (enumerate_par.hip)
            inf.resize((size_t)n_as * 3);
            for (int k = 0; k < n_as; ++k) { inf[3 * k] = rec[4 * k + 1]; inf[3 * k + 1] = rec[4 * k + 2]; inf[3 * k + 2] = rec[4 * k + 3]; }
            slot_order(n_as, 1, inf.data(), order);
            for (int k = 1; k < n_as; ++k) { uint32_t u = rec[4 * (size_t)order[k]]; memcpy(&scv[k], &u, 4); }
          } else {
            for (int k = 1; k < n_as; ++k) { uint32_t u = rec[4 * (size_t)k]; memcpy(&scv[k], &u, 4); }
          }
          scv[0] = b->islocal ? res[p].best : res[p].corner;
          keys.resize(n_as);
          for (int k = 0; k < n_as; ++k) { keys[k].score = scv[k]; keys[k].idx = k; }
          if (mx >= n_as) std::sort(keys.begin(), keys.end());
          else if (mx > 0) { std::partial_sort(keys.begin(), keys.begin() + mx, keys.end()); keys.erase(keys.begin() + mx, keys.end()); }
          int keep = (int)keys.size();
          if (keep > K) { status[p] = ALN_E_OVERFLOW; keep = K; }     // the caller's K slots are too few for this set
          n_out[p] = keep;
          for (int k = 0; k < keep; ++k) {
            sel[(size_t)g * K + k] = pw ? order[keys[k].idx] : keys[k].idx;
            scores[(size_t)p * K + k] = keys[k].score;
          }
        }
      };
      const int nthreads = (int)std::max(1u, std::min({std::thread::hardware_concurrency(), 16u, (unsigned)((total_slots >> 14) + 1)}));
      if (nthreads == 1) work(0, 1);
      else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t, nthreads);
        for (auto& t : th) t.join();
      }
    }
    // unroll every survivor on the device
    tm[3] = now();
    BTRY(hipMalloc((void**)&d_sel, sel.size() * 4));
    BTRY(hipMalloc((void**)&d_lens, sel.size() * 4));
    if (pairs) BTRY(scratch(6, sel.size() * (size_t)pair_stride * 8, (void**)&d_lists));
    BTRY(hipMemcpyAsync(d_sel, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    BTRY(hipEventRecord(evs[2], ctx->stream));
    hipLaunchKernelGGL(enum_unroll_all_kernel, dim3(K, gn), dim3(64), 0, ctx->stream, a.node_pair, a.node_next,
                       pw ? a.node_len : (const uint8_t*)nullptr, a.head,
                       pw ? (size_t)a.n_chunks * kChunkNodes : (size_t)a.node_cap, pw ? (int)a.n_pools : 0, a.ali_cap,
                       d_sel, K, d_list, b->d_path, b->path_stride, b->d_res, d_lists, d_lens, pairs ? pair_stride : (1 << 30),
                       gn == n ? 1 : 0);
    BTRY(hipGetLastError());
    BTRY(hipEventRecord(evs[3], ctx->stream));
    hlens.resize(sel.size());
    BTRY(hipMemcpyAsync(hlens.data(), d_lens, sel.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    const bool in_place = pairs && gn == n;             // every pair of the batch: laid out in batch order by the kernel, one copy
    if (in_place) BTRY(hipMemcpyAsync(pairs, d_lists, sel.size() * (size_t)pair_stride * 8, hipMemcpyDeviceToHost, ctx->stream));
    BTRY(hipStreamSynchronize(ctx->stream));
    for (int g = 0; g < gn; ++g) {
      const int p = ids[g];
      if (deferred[g] == 1) continue;
      for (int k = 0; k < K; ++k) lengths[(size_t)p * K + k] = hlens[(size_t)(gn == n ? p : g) * K + k];
      if (pairs && !in_place)
        BTRY(hipMemcpy(pairs + (size_t)p * K * pair_stride * 2, d_lists + (size_t)g * K * pair_stride * 2, (size_t)K * pair_stride * 8,
                       hipMemcpyDeviceToHost));
    }
    float ms0 = 0.f, ms1 = 0.f;
    BTRY(hipEventElapsedTime(&ms0, evs[0], evs[1]));
    BTRY(hipEventElapsedTime(&ms1, evs[2], evs[3]));
    b->enum_search_ms += ms0; b->enum_unroll_ms += ms1;
    tm[4] = now();
    if (ctx->hints.enum_debug) {
      uint32_t used = 0;
      if (pw) hipMemcpy(&used, a.chunk_next, 4, hipMemcpyDeviceToHost);      // (pool 0)
      long long nodes = 0, slots = 0; int novf = 0;
      for (int g = 0; g < gn; ++g) { nodes += (uint32_t)hout[4 * g + 1]; slots += hout[4 * g]; novf += hout[4 * g + 2] != 0; }
      fprintf(stderr, "[enumerate_all] group of %d pairs, %d waves/pair, node_cap %u ali_cap %u: search %.2f ms, unroll %.2f ms, again %zu, serial %zu; chunks %u of %u, nodes %lld, slots %lld, failed %d\n",
              gn, pw, a.node_cap, a.ali_cap, ms0, ms1, again.size(), serial_todo.size(), used, a.n_chunks, nodes, slots, novf);
      fprintf(stderr, "[enumerate_all]   host seconds: alloc %.3f, search + sync %.3f, sets %.3f, unroll + copy %.3f\n", tm[1] - tm[0], tm[2] - tm[1],
              tm[3] - tm[2], tm[4] - tm[3]);
    }
    free_group();
    return ALN_OK;
  };
  for (int round = 0; !todo.empty(); ++round) {
    if (ali_cap > limit_cap) ali_cap = limit_cap;
    // the last round: the retries are used up and the alignment pool has reached the size user_limit bounds
    const bool final_round = round >= max_round && (ali_cap >= limit_cap || round >= max_round + 12);
    again.clear(); serial_todo.clear();
    for (int pass = 0; pass < 2; ++pass) {                // pass 1: the pairs pass 0's several-wave search handed back
      const std::vector<int32_t>& list = pass == 0 ? todo : serial_todo;
      if (list.empty()) continue;
      const std::vector<int32_t> ids_all(list);          // (run_group appends to serial_todo)
      const int pw0 = pass == 0 ? par_waves(b, noa->kind, (int)ids_all.size()) : 0;
      const size_t per_pair = (size_t)node_cap * 8 + (size_t)ali_cap * (ks ? 12 : 8) +
                              (pw0 ? (size_t)ali_cap * (kTaskWords * 4 + 12) : (size_t)a0.stack_cap * frame_words * 4) +
                              (pairs ? (size_t)K * pair_stride * 8 : 0) + (cr ? (size_t)a0.sort_limit * a0.cr_tpad * 2 + (size_t)a0.cr_tpad * 4 : 0);
      size_t gmax = std::max<size_t>(1, kPoolBudget / per_pair);
      if (round == 0 && pass == 0) gmax = ids_all.size();               // the caller sized round 0
      gmax = (ids_all.size() + (ids_all.size() + gmax - 1) / gmax - 1) / ((ids_all.size() + gmax - 1) / gmax);   // groups of equal size
      for (size_t g0 = 0; g0 < ids_all.size(); g0 += gmax) {
        const int gn = (int)std::min(gmax, ids_all.size() - g0);
        const int rcg = run_group(ids_all.data() + g0, gn, final_round, pw0 ? par_waves(b, noa->kind, gn) : 0);
        if (rcg != ALN_OK) return rcg;
      }
    }
    todo.swap(again);
    if (final_round) break;
    if (node_cap <= (1u << 29)) node_cap *= 4;
    if (ali_cap <= (1u << 22)) ali_cap *= 4;
  }
#undef BTRY
  cleanup();
  int worst = ALN_OK;
  for (int p = 0; p < n; ++p) if (status[p] != 0 && worst == ALN_OK) worst = status[p];
  for (int p = 0; p < n; ++p) if (status[p] == ALN_E_OVERFLOW) worst = ALN_E_OVERFLOW;
  return worst;
}

extern "C" int aln_batch_last_enum_usage(aln_batch* b, int32_t* alignments, int32_t* nodes) {
  if (!b) return ALN_E_ARG;
  if ((int)b->enum_usage.size() != 4 * b->n_pairs) return ALN_E_STATE;
  for (int p = 0; p < b->n_pairs; ++p) {
    if (alignments) alignments[p] = b->enum_usage[4 * p];
    if (nodes) nodes[p] = b->enum_usage[4 * p + 1];
  }
  return ALN_OK;
}

extern "C" int aln_batch_last_enum_ms(aln_batch* b, float* search_ms, float* unroll_ms) {
  if (!b) return ALN_E_ARG;
  if (search_ms) *search_ms = b->enum_search_ms;
  if (unroll_ms) *unroll_ms = b->enum_unroll_ms;
  return ALN_OK;
}
