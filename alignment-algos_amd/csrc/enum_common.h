// enum_common.h — pools and helpers shared by the near-optimal enumeration kernels (enumerate.hip, enumerate_ks.hip).
#pragma once
#include "aln_device.h"

namespace aln {

constexpr uint32_t kChunkNodes = 1u << 16;   // enumerate_par.hip: trie nodes a workgroup takes from the launch's pool at a time
// which of a launch's node pools workgroup i uses: scattered, so that a pattern in the batch (bench.py alternates unrelated
// and homologous pairs) does not put all heavy searches into one pool
__host__ __device__ __forceinline__ uint32_t enum_pool_of(uint32_t i, uint32_t n_pools) { return ((i * 0x9E3779B1u) >> 12) % n_pools; }
constexpr int kTaskWords = 8;    // enumerate_par.hip: cell, slot, trie head, score, kind/force, (pad)
constexpr int kParSerial = -100; // enumerate_par.hip -> host: this pair's set outgrows user_limit, search it with the one-wave kernel
constexpr int kFrameWords = 8;   // q0, t0, k0, cursor, curr_head, curr_score, r, (pad) — one active branch() invocation

struct EnumArgs {
  int kind;             // ALN_ENUM_CW / ALN_ENUM_UCW
  uint32_t user_limit;
  float delta_ratio;
  int first_slot;       // index of the enumerator's seed alignment inside the set (what is already there stays)
  // pools (for this pair)
  uint32_t* node_pair; uint32_t* node_next; uint32_t node_cap;
  uint8_t* node_len;    // enumerate_par.hip: a node is a RUN of node_len cells down the diagonal from node_pair's cell (list order: the
                        // run's far end first); nullptr: every node is one cell (the one-wave kernels)
  uint32_t* head; float* score; uint32_t ali_cap;
  uint32_t* stack; uint32_t stack_cap;   // frames of kFrameWords words
  const uint8_t* flags; // T bytes
  int ptr_mode;         // pointer word encoding of the P plane
  int h_mode;           // score plane element type
  int32_t* out;         // [0] = set size, [1] = nodes used, [2] = status
  // batched launches (one block per pair): block b works on pair pair0 + b with the b-th slice of every pool
  int flags_stride;     // bytes between two pairs' flag rows (0: every pair shares one row)
  const int32_t* pair_list;   // block b works on pair pair_list[b] (nullptr: pair0 + b)
  // enumerate_par.hip only
  uint32_t* task;       // [ali_cap][kTaskWords]: pending sub-searches (every pending task owns a distinct slot)
  uint32_t* slot_info;  // [ali_cap][3]: for slots the search created: parent slot, t0 of the branch node, candidate index
  // node pools shared by the pairs of a launch (32-bit node indices: < 2^32 nodes each; workgroup i uses pool i % n_pools),
  // handed out in chunks of kChunkNodes
  uint32_t* chunk_next; // [n_pools] next free chunk (device counters)
  uint32_t n_chunks;    // chunks per pool
  uint32_t n_pools;
  int int_sums;         // 1: similarities, gaps and scores are integers below 2^24 (the tagged DP kernel's own precondition): sums in any order
  const float* rowmax;  // [pairs of the batch][bm_rows][nbt] block maxima of the score plane (enum_blockmax_kernel), or nullptr
  const float* colmax;  // [pairs of the batch][bm_cols][nbq]
  int bm_rows, bm_cols, nbt, nbq, bm_pair0;   // (arrays start at pair bm_pair0 of the batch)
  // KSConstrainedNearOptimal only
  uint32_t k_limit;     // NOaliParams::k_limit: operations a branch node may keep
  int32_t* uid;         // uid of every alignment (kscw.h:121,262)
  uint32_t cand_cap;    // capacity of the LDS candidate arrays
  // CRConstrainedNearOptimal only
  uint32_t sort_limit;  // NOaliParams::sort_limit: operations a branch node sorts and follows (<= 512)
  float max_overlap;    // NOaliParams::max_overlap
  uint16_t* cr_ali;     // [sort_limit][cr_tpad]: query position per template position of every operation's sub-path (0xFFFF = none)
  int32_t* cr_reg;      // [cr_tpad]: reg[t] = region of template position t (t >= 1), reg[0] = the origin's own region
  int cr_tpad;
};

constexpr uint32_t kNoNode = 0xFFFFFFFFu;

// One alignment out of the trie, in list order (head -> end), by one wave: a node is one cell, or (node_len != nullptr) a run of
// node_len[node] diagonal cells ending at node_pair[node]'s cell, whose cells the lanes write together.  -> list length, or -1
// when it exceeds `stride`.  o may be nullptr (count only).
__device__ __forceinline__ int unroll_alignment(const uint32_t* __restrict__ node_pair, const uint32_t* __restrict__ node_next,
                                                const uint8_t* __restrict__ node_len, uint32_t node, int32_t* __restrict__ o, int stride) {
  const int lane = threadIdx.x & 63;
  int n = 0;
  while (node != kNoNode) {
    const uint32_t w = node_pair[node];
    const int len = node_len ? (int)node_len[node] : 1;
    if (n + len > stride) return -1;
    if (o && lane < len) {
      const int back = len - 1 - lane;                       // ascending cells: the run's far end (smallest indices) first
      o[2 * (n + lane)] = (int32_t)(w >> 16) - back;
      o[2 * (n + lane) + 1] = (int32_t)(w & 0xFFFFu) - back;
    }
    n += len;
    node = node_next[node];
  }
  return n;
}

// All mutable pool words are accessed with agent-scope (L2-served) loads/stores: lane 0 writes, every lane reads.
__device__ __forceinline__ uint32_t ld_u(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Words that only the waves of ONE workgroup exchange (enumerate_par.hip's task stack): workgroup scope — on gfx950 an agent-scope
// release (__threadfence) writes the XCD's L2 back, which a search that has megabytes of fresh trie nodes in it pays for every round.
__device__ __forceinline__ uint32_t ld_w(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_w(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// sc + sv[lane 0] + sv[lane 1] + ... + sv[lane n-1], added in that order (fp32 is not associative; the reference adds one similarity
// per path step).  n is wave-uniform: v_readlane_b32 with a scalar lane index, no LDS round trip per step.
__device__ __forceinline__ float add_in_path_order(float sc, float sv, int n) {
  for (int l = 0; l < n; ++l) sc += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sv), l));
  return sc;
}

// Sum of an int over the wave, in every lane: six DPP adds (quad swaps, half-row and row mirrors, row broadcasts) and a readlane.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);    // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);    // row_mirror: every lane holds its row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);    // row_bcast15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);    // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_amdgcn_readlane(v, 63);
}

// Maximum of a float over the wave, in every lane (the DPP ladder of wave_sum_i32 with v_max_f32).
__device__ __forceinline__ float wave_max_f32(float v) {
  const float NEG = -3.0e38f;
  auto dpp = [&](float x, int ctrl_sel) -> float {
    const int xi = __float_as_int(x), ni = __float_as_int(NEG);
    int r;
    switch (ctrl_sel) {
      case 0: r = __builtin_amdgcn_update_dpp(ni, xi, 0xB1, 0xF, 0xF, false); break;
      case 1: r = __builtin_amdgcn_update_dpp(ni, xi, 0x4E, 0xF, 0xF, false); break;
      case 2: r = __builtin_amdgcn_update_dpp(ni, xi, 0x141, 0xF, 0xF, false); break;
      case 3: r = __builtin_amdgcn_update_dpp(ni, xi, 0x140, 0xF, 0xF, false); break;
      case 4: r = __builtin_amdgcn_update_dpp(ni, xi, 0x142, 0xA, 0xF, false); break;
      default: r = __builtin_amdgcn_update_dpp(ni, xi, 0x143, 0xC, 0xF, false); break;
    }
    return __int_as_float(r);
  };
  v = fmaxf(v, dpp(v, 0)); v = fmaxf(v, dpp(v, 1)); v = fmaxf(v, dpp(v, 2)); v = fmaxf(v, dpp(v, 3));
  v = fmaxf(v, dpp(v, 4)); v = fmaxf(v, dpp(v, 5));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace aln
