// Internal declarations shared by the HIP translation units of libalnhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <atomic>
#include <string>
#include <vector>

#include "aln_hip.h"

namespace aln {

constexpr uint32_t kNullPtr = 0xFFFFFFFFu;   // packed (prev_q<<16 | prev_t) of an untouched cell = (-1,-1)
constexpr int kMaxLen = 65534;               // 16-bit packed indices, 0xFFFF reserved for DPCell::null
constexpr int kCodeHead = 30, kCodeTail = 31;   // residue codes of '^' and '$'; alphabet codes are 0..n-1 (n <= 30)
constexpr int kNeg = -(1 << 28);             // "-infinity" of the integer kernels (leaves headroom for subtractions)
// Row stride of a pair's planes in cells: T rounded up to `align` cells (a power of two >= 8; context hint "plane_row_align").
// 8: fp32 rows are 32-byte and uint16 rows 16-byte aligned and a lane's 8-cell store never straddles the end of a row.
// 64: uint16 rows start on 128-byte lines, so every wave-wide store covers whole cache lines.
inline int row_stride(int T, int align) { return (T + align - 1) & ~(align - 1); }

// One DPMatrix of the batch, as the kernels see it.
struct PairDesc {
  int32_t Q, T;          // sizes incl. sentinels
  int32_t ld;            // row pitch of the planes in elements (T rounded up to 4 -> 16-byte aligned rows)
  int32_t q_seq, t_seq;  // sequence indices (for per-position side arrays)
  int64_t q_off, t_off;  // first residue of the query / template in their code pools
  int64_t plane_off;     // first element of this pair's Q x ld planes
  // sub-rectangle of build_subdpm (full build: 0, Q-1, 0, T-1)
  int32_t q0, q1, t0, t1;
  int64_t ins_off;       // ALN_GAP_TABLES: first element of this pair's three T x Q insertion planes
};

// Per-pair results kept on the device.
struct PairResult {
  float corner;          // score of the final cell ((q1,t1) forward, (q0,t0) reverse)
  float best;            // find_max value (local) or corner (otherwise)
  int32_t best_q, best_t;
  int32_t n_path;        // traceback length (pairs), filled by the traceback kernel
  int32_t status;        // 0 or ALN_E_STARTPAIR
  float part_max;        // scratch: running maximum over interior cells (find_max partial)
  uint32_t part_pos;     // scratch: packed (row<<16|col) of its first row-major occurrence
};

struct GapDev {
  int32_t model, align_type;
  float gi, ge;
  const float* tgi;      // device, template pool positions
  const float* tge;
  int32_t free_del, free_ins;   // free end gaps for deletions / insertions (aasubalib.h:34-49,60-75)
};

}  // namespace aln

// Tuning / kernel-selection hints of a context (aln_ctx_set_hint).  Defaults come from the environment variables named in
// aln_hints.hip's table, read ONCE when the context is created — nothing reads the environment at launch time.
struct aln_hints {
  int tag_kernel = 1;        // 0: never use the tagged-key kernel (dp_affine_int instead)
  int h16 = 1;               // 0: keep fp32 score planes in local tagged builds
  int key16 = 1;             // 0: never use the 16-bit key layout
  int tag_alt_prio = 1;      // tagged kernel: alternate s_setprio per row by hardware-slot parity (pays on lone launches; a caller that
                             // overlaps launches of several contexts sets 0)
  int tag_lag = 0;           // tagged kernel: rows wave w runs behind wave w-1 (skewed exchange, one barrier every tag_lag rows); 0 = the
                             // synchronous per-row exchange.  Measured on MI355X (config 2, lone launches): lag 0 3.11 ms, 1: 3.20, 2: 3.10,
                             // 4: 3.28; four overlapping streams 2.82 vs 3.38 — the waves of a pair drifting apart costs more in HBM
                             // row locality (the two halves of a plane row are written rows apart) than the barrier chain it removes
  int tag_occupancy = 0;     // tagged kernel, 16-cell lanes: waves per SIMD it is compiled for: 2, 3, or 0 = three when the launch alone has >= 3 per SIMD
  int tag_bits = 0;          // 12: the 12-tag-bit layout (pointer dialect 2) also for sequences of up to 2048 residues (tests); 0 = by length
  int tag_solo = 0;          // 1: pairs of 1025..2048 columns run in dp_affine_solo (one wave per pair, no barriers; wants >= 2048 pairs in flight)
  int tag_segments = 0;      // tagged kernel: (pair, row segment) work items handed out by a queue (dp_affine_tag.hip "Segment queue"):
                             // K in 2..8 = long pairs are cut into K segments when the batch alone fills the GPU (>= 512 pairs), -K = whenever
                             // pairs are long, 0 = one workgroup per pair
  int dp_nw = 0, dp_r = 0, dp_x = 0;   // force a row-sweep variant (waves per pair, groups per lane, columns per lane and group); 0 = auto
  int exact_tiles = 1;       // 0: dp_exact_blocked instead of dp_exact_tiled where both apply
  int exact_literal = 0;     // 1: the literal O(n^3) kernel everywhere
  int exact_alt_prio = 2;    // tiled exact kernel: 1 = priority rotation over the 4 resident waves, 2 = priority by progress (waves ahead of the launch's average yield), 0 = none
  int exact_wavefront = 1;   // tiled exact kernel: 64-column tiles, the four waves of a pair on four row blocks (one barrier per tile); 0: 256-column tiles, one barrier per row
  int exact_prune = 1;       // tiled exact kernel: skip far chunks that provably cannot matter (bit-exact; dp_exact_blocked.hip)
  int exact_debug = 0;       // 1: the tiled kernel counts tested / skipped far chunks (aln_batch_last_exact_stats); 2: the same four words hold
                             // how long its waves ran (sum, longest, ~shortest, count; units of 1024 s_memtime ticks)
  int score_packed = 1;      // 0: one query per wave in aln_score_all_vs_all
  int plane_row_align = 8;   // cells a plane row is padded to when a batch is created (8, 16, 32 or 64)
  int64_t enum_node_cap = 0; // trie nodes of aln_batch_enumerate (0 = default)
  int enum_keep_pools = 1;   // 1: aln_batch_enumerate_all keeps its device pools with the batch (freed with it); 0: frees them when it returns
  int enum_debug = 0;        // 1: aln_batch_enumerate_all reports every group of pairs it searches on stderr
  int enum_waves = 0;        // cw / ucw search: waves per pair (enumerate_par.hip); 0 = by the number of pairs, 1 = the one-wave kernel
  int enum_heavy_first = 1;  // aln_batch_enumerate_all: launch the pairs in the order of what the batch's previous search used, heaviest first
  int enum_pool_retries = 2; // aln_batch_enumerate_all: times a pair whose pools overflowed is searched again with 4 x the capacity
};

struct aln_comm;

struct aln_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipStream_t copy_stream = nullptr;   // device -> host copies that must not hold up the launch stream (created on first use)
  std::string last_error;
  aln_hints hints;
};

struct aln_batch {
  aln_ctx* ctx;
  int32_t n_pairs;
  bool score_only;
  std::vector<aln::PairDesc> h_pairs;
  std::vector<int64_t> q_offsets, t_offsets;   // host copies of the pools' offsets
  std::string q_res, t_res;                    // host copies of residues (for host-side helpers / lowering)
  int64_t q_total, t_total;
  int32_t maxQ, maxT;
  int32_t maxld = 0;                           // largest row stride of a pair
  int64_t plane_elems;
  int64_t cells;                               // sum (Q-2)(T-2)
  // device
  aln::PairDesc* d_pairs;
  uint8_t* d_qcodes; uint8_t* d_tcodes;        // residue codes of the last SUBMATRIX dp
  float* d_H; uint32_t* d_P; float* d_S;       // planes (d_S only for SIM_MATRIX / HMAP2)
  float* d_sabs = nullptr;                     // per pair max |S|, left by hmap2_apply_kernel (the exact-order kernel's rounding margin)
  bool sabs_valid = false;                     // ... and whether it describes the resident d_S
  aln::PairResult* d_res;
  int32_t* d_table32;                          // 32x32 int substitution table (fast path)
  float* d_tablef;                             // 32x32 float table (exact path / getSim)
  float* d_tgi; float* d_tge;                  // AFFINE_TPOS_MIN arrays (template pool positions)
  float* d_tcn = nullptr; float* d_deltab = nullptr; int64_t* d_deltab_off = nullptr;   // DEL_TABLE_INS_TPOS (Gn2Eval), TABLES
  float* d_instab = nullptr;                                                            // TABLES
  float* d_deltabR = nullptr; bool deltabR_valid = false;     // DEL_TABLE_INS_TPOS, reverse builds: flipped transposes of the tables
  int64_t* d_pair_deloff = nullptr;                           // ... first element of every PAIR's table
  int32_t n_tseqs = 0;
  int32_t* d_path;                             // traceback output, n_pairs x path_stride x 2
  int32_t path_stride;
  int32_t* d_bounds;
  float* d_xscratch = nullptr; size_t xscratch_floats = 0;   // far-insertion scratch of dp_exact_blocked
  int* d_tagq = nullptr; size_t tagq_bytes = 0;              // segment queue of dp_affine_tag (counters, error word, item slots)
  uint32_t* d_tagstate = nullptr; size_t tagstate_bytes = 0; // ... and its hand-off slots
  bool tag_segmented = false;                                // the last tagged launch used them
  // state of the last dp
  bool have_dp, have_sub;
  int32_t sim_kind, direction, algo, bug_b4;
  aln_gap gap;                                 // host copy (pointers not retained beyond dp call)
  aln::GapDev gapdev;
  bool islocal;
  bool gap_ext_nonneg = true;                  // every gap extension coefficient of the last description is >= 0 (gaps grow with distance)
  unsigned long long exact_stats[4] = {0, 0, 0, 0};   // hint exact_debug: far chunks tested / skipped, deletions then insertions
  bool simplane_integral = false;              // ALN_SIM_MATRIX planes of the last dp were all small integers (kept for reevaluate)
  int32_t ptr_mode;                            // encoding of the P plane words (aln_device.h decode_ptr)
  int32_t h_mode;                              // score plane element type: 0 fp32, 1 uint16 (aln_device.h load_score)
  std::string kernel_name;
  hipEvent_t ev0, ev1;                          // around the DP kernel(s) of the latest build
  static const int kEvRing = 64;               // ... and of the builds before it (aln_batch_dp_ms_history)
  hipEvent_t ring0[kEvRing] = {}, ring1[kEvRing] = {};
  long n_builds = 0;
  float enum_search_ms = 0.f, enum_unroll_ms = 0.f;   // last aln_batch_enumerate_all
  // device pools of aln_batch_enumerate_all, kept between calls (hint enum_keep_pools): a hipMalloc of tens of GB costs seconds
  uint8_t* h_stage_pin = nullptr; size_t h_stage_bytes = 0;   // pinned staging of residue codes + table (upload_submatrix)
  hipEvent_t stage_ev = nullptr;                              // ... behind the last upload's copies out of it
  bool pairs_dirty = false;                                   // d_pairs differs from the full-rectangle descriptors in h_pairs
  struct Scratch { void* p = nullptr; size_t bytes = 0; };
  Scratch enum_scratch[9];
  std::vector<int32_t> enum_usage;                    // ... and what every pair's search used of its pools
  // aln_batch_optimal_enqueue / _collect: two pinned result slots
  aln::PairResult* h_slot[2] = {nullptr, nullptr};
  hipEvent_t slot_ev[2] = {nullptr, nullptr};
  bool slot_local[2] = {false, false};
  int slot_head = 0, slot_count = 0;
  // aln_batch_optimal_strings_enqueue / _collect (gapped_strings.hip): residue characters on the device, two device + two pinned slots
  char* d_qchars = nullptr; char* d_tchars = nullptr;
  char* d_str_lines[2] = {nullptr, nullptr}; char* h_str_lines[2] = {nullptr, nullptr};
  void* d_str_out[2] = {nullptr, nullptr}; void* h_str_out[2] = {nullptr, nullptr};
  hipEvent_t str_ev[2] = {nullptr, nullptr}, str_kernel_ev[2] = {nullptr, nullptr};
  int32_t str_stride = 0;
  int str_head = 0, str_count = 0;
  std::vector<int32_t> h_bounds;
  // retained similarity description for reevaluate()
  std::vector<float> h_table; int32_t alpha_n; std::string alphabet;
};

#define ALN_HIP_CHECK(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);           \
      return ALN_E_HIP;                                                                 \
    }                                                                                   \
  } while (0)

namespace aln {

// aln_hints.hip
void hints_from_env(aln_hints* h);
// dp_affine_int.hip
int launch_dp_affine_int(aln_batch* b, bool use_simplane);
bool fast_path_legal(const aln_batch* b, const float* table, int n, const aln_gap* gap, bool simplane_integral);
// dp_affine_tag.hip
int launch_dp_affine_tag(aln_batch* b);
bool tag_path_legal(const aln_batch* b, const float* table, int n, const aln_gap* gap);
int tag_path_bits(const aln_batch* b, const float* table, int n, const aln_gap* gap);   // 0 (not legal), 11 or 12 tag bits
bool tag_h16_legal(const aln_batch* b);
// dp_affine_solo.hip
bool dp_affine_solo_legal(const aln_batch* b);
int launch_dp_affine_solo(aln_batch* b);
// dp_corner.hip
int launch_dp_corner(aln_batch* b);
// traceback.hip
int launch_traceback(aln_batch* b, bool subali);
// gapped_strings.hip
void free_string_buffers(aln_batch* b);
// dp_exact.hip
int launch_dp_exact(aln_batch* b);
// dp_exact_blocked.hip
bool dp_exact_blocked_legal(const aln_batch* b);
int launch_dp_exact_blocked(aln_batch* b);
// sim_hmap2.hip
int launch_sim_hmap2(aln_batch* b, const aln_sim* sim);

}  // namespace aln
