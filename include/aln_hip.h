/*
 * aln_hip.h — C ABI of the MI355X (gfx950) alignment engine, libalnhip.so.
 *
 * The reference (christang/alignment-algos) has no ABI boundary: its hot path is the header
 * template DPMatrix<S1,S2,Etype> driven by CRTP Evaluator plugins (evaluator.h:20-97) and
 * walked by Enumerator classes (enumerator.h:20-25).  This header is the boundary a
 * maintainer binds instead (INTEGRATION.md shows the C++ shim): each entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; the caller owns every host
 * buffer, the library owns device memory inside aln_ctx / aln_batch.  Calls on one ctx are
 * serialised by the caller; different ctx objects may be used from different threads.
 *
 * Index conventions are the reference's: a sequence includes its '^' head and '$' tail
 * sentinels (sequence.cpp:15-16, fastaio.h:126,137), Q = query size, T = template size,
 * matrices are Q x T row-major, DPCell::null is -1 (dpmatrix.cpp:15).
 *
 * Return codes: 0 = ok; negative values mirror the reference's throw sites so that a C++
 * wrapper can rethrow the same std::string (aln_error_string()).
 */
#ifndef ALN_HIP_H
#define ALN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values equal the reference's) ------------------------------------------- */
enum aln_align_t {            /* alib.h:20-26 */
  ALN_GLOBAL_LOCAL = 0, ALN_GLOBAL = 1, ALN_LOCAL_GLOBAL = 2, ALN_LOCAL = 3, ALN_SEMI_LOCAL = 4
};
enum aln_direction_t { ALN_FWD = 1, ALN_REV = 2 };   /* dpmatrix.h:23-26 */

enum aln_gap_model {
  ALN_GAP_AFFINE_CONST = 0,     /* AASubstitutionEval::deletion/insertion, aasubalib.h:27-77 */
  ALN_GAP_AFFINE_TPOS_MIN = 1,  /* Hmap2Eval / HMAPaliEval, hmap2_eval.h:41-95 (min of the two template positions) */
  ALN_GAP_DEL_TABLE_INS_TPOS = 2, /* Gn2Eval, gn2_eval.h:100-165: deletion(t1,t2) is a per-template table (vv_gi/vv_ge/vv_cd + the
                                   8100 distance rule, materialised by the caller), insertion = (gi[t1] + ge[t1]*(di-2)) + cn[t1] */
  ALN_GAP_TABLES = 3            /* any Evaluator whose deletion() depends on (t1,t2) and whose insertion() depends on (t1, q2-q1)
                                   away from the query ends: both functions materialised by the caller (aln_lowering.h does it
                                   for an arbitrary evaluator.h plugin; covers GnoaliEval, gnoalib.h:90-185) */
};

enum aln_sim_kind {
  ALN_SIM_SUBMATRIX = 0,        /* residue codes + substitution table, aasubalib.h:17-25 + submatrix.h:36-38 */
  ALN_SIM_MATRIX = 1,           /* caller-materialised SimilarityMatrix planes (any Evaluator::similarity + post_process) */
  ALN_SIM_HMAP2 = 2             /* Hmap2Eval::similarity + post_process computed on the device, hmap2_eval.h:27-39,98-101 */
};

enum aln_dp_algo {
  ALN_DP_AUTO = 0,              /* O(n^2) running-max kernel when every value is a small integer, else exact */
  ALN_DP_EXACT = 1,             /* exact-order O(n^3) kernel: literal dpmatrix.h:447-486 arithmetic */
  ALN_DP_FAST = 2               /* force the O(n^2) kernel; ALN_E_NOT_INTEGRAL if the proof fails */
};

enum aln_enum_kind {
  ALN_ENUM_CW = 0,              /* ConstrainedNearOptimal, cw.h:68-284 */
  ALN_ENUM_UCW = 1,             /* UnconstrainedNearOptimal, ucw.h:64-236 */
  ALN_ENUM_KSCW = 2,            /* KSConstrainedNearOptimal, kscw.h:109-351 (parity unpinned: the reference header does not
                                   compile on LP64) */
  ALN_ENUM_CRCW = 3             /* CRConstrainedNearOptimal, crcw.h:134-594 (parity unpinned, same reason) */
};

enum aln_status {
  ALN_OK = 0,
  ALN_E_BOUNDS = -1,            /* "Illegal bounds building DPM"   dpmatrix.h:361,544,699,885 */
  ALN_E_GAPSTYLE = -2,          /* "Illegal gap style"             aasubalib.h:46,72 */
  ALN_E_STARTPAIR = -3,         /* "Illegal alignment start pair"  optimal.h:74 */
  ALN_E_RESIDUE = -4,           /* residue outside the substitution alphabet (UB in submatrix.h:36-38) */
  ALN_E_ARG = -5,               /* bad argument */
  ALN_E_HIP = -6,               /* HIP runtime error, see aln_last_error() */
  ALN_E_NOMEM = -7,
  ALN_E_TOO_LONG = -8,          /* sequence longer than this build's kernels handle */
  ALN_E_NOT_INTEGRAL = -9,      /* ALN_DP_FAST requested but scores/gaps are not small integers */
  ALN_E_STATE = -10,            /* call order: dp before optimal/enumerate/get_cells */
  ALN_E_OVERFLOW = -11          /* an output buffer or the enumeration pool is too small */
};

typedef struct aln_ctx aln_ctx;
typedef struct aln_batch aln_batch;

/* ---- plain descriptors ---------------------------------------------------------------- */

/* A pool of sequences; replaces Sequence<elem_t>/AASequence (sequence.h:41-64, aa_seq.h:10-25).
 * Sequence s is residues[offsets[s] .. offsets[s+1]) and INCLUDES '^' and '$'. */
typedef struct {
  int32_t n_seqs;
  const int64_t* offsets;       /* n_seqs + 1 */
  const char* residues;         /* one-letter codes (SequenceElem::olc) */
} aln_seqs;

/* SubstitutionMatrix / BlosumMatrix (submatrix.h:19-48): n x n row-major over `alphabet`. */
typedef struct {
  int32_t n;                    /* <= 30 */
  const char* alphabet;
  const float* table;
} aln_submatrix;

/* Per-position profile records for Hmap2Eval (HMAPElem, hmapalib_seq.h:28-99), one per residue of
 * the pool, sentinels included: aa[20] (aa_profile), sse[3] (sse_values), conf (sse_confid). */
typedef struct {
  const float* aa;              /* total_residues x 20 */
  const float* sse;             /* total_residues x 3  */
  const float* conf;            /* total_residues      */
} aln_profiles;

/* Gap model + AliParams (alib.h:28-46). */
typedef struct {
  int32_t model;                /* aln_gap_model */
  int32_t align_type;           /* aln_align_t */
  float gap_init, gap_extn;     /* AFFINE_CONST: AliParams::gap_init_penalty / gap_extn_penalty */
  const float* t_gap_init;      /* AFFINE_TPOS_MIN: per residue of the TEMPLATE pool (HMAPElem::gap_init()) */
  const float* t_gap_extn;
  int32_t dp_local;             /* the DPMatrix constructor's own `type == local` (dpmatrix.h:155) when it differs from the
                                   evaluator's align_type: 0 = same as align_type, 1 = not local, 2 = local */
  /* DEL_TABLE_INS_TPOS only.  t_gap_init / t_gap_extn / t_gap_cn are Gn2Eval's v_gi / v_ge / v_cn per residue of the TEMPLATE
   * pool (gn2_eval.cpp:113-130).  del_table: template sequence s (T residues) owns T*T floats at del_table[del_table_off[s]],
   * entry [t1*T + t2] = exactly what Evaluator::deletion(q,t,.,.,t1,t2) returns for t1 < t2 (end rules included). */
  const float* t_gap_cn;
  const float* del_table;       /* DEL_TABLE_INS_TPOS and TABLES */
  const int64_t* del_table_off; /* n template sequences */
  /* TABLES only: pair p (Q x T) owns three T x Q planes at ins_tables[ins_table_off[p]]: [0][t1*Q + d] = insertion(q1,q1+d,t1,t1+1)
   * for interior q1 and q1+d; [1][t1*Q + q2] = insertion(0,q2,t1,t1+1); [2][t1*Q + q1] = insertion(q1,Q-1,t1,t1+1).  No
   * end-gap rule is applied by the library in this model: the tables are the evaluator's own values. */
  const float* ins_tables;
  const int64_t* ins_table_off; /* n pairs */
} aln_gap;

/* Similarity source. */
typedef struct {
  int32_t kind;                 /* aln_sim_kind */
  aln_submatrix sub;            /* ALN_SIM_SUBMATRIX */
  const float* planes;          /* ALN_SIM_MATRIX: pair p's Q x T row-major plane starts at planes[plane_off[p]] */
  const int64_t* plane_off;
  aln_profiles q_prof, t_prof;  /* ALN_SIM_HMAP2 */
  float alpha;                  /* HMAPaliParams::alpha   (hmap_eval.cpp:4)  */
  float zero_shift;             /* HMAPaliParams::zero_shift (hmap_eval.cpp:7) */
  int32_t normalize;            /* run post_process (z-normalise + shift) */
} aln_sim;

/* NOaliParams subset used by cw.h / ucw.h (noalib.h:18-45). */
typedef struct {
  int32_t kind;                 /* aln_enum_kind */
  int32_t number_suboptimal;    /* NUM_SUBOPT; sortSet(max) */
  float delta_ratio;            /* DELTA_RATIO */
  uint32_t user_limit;          /* 0 = the reference's hard-coded 1000000 (cw.h:76) / 100000 (ucw.h:72) */
  int32_t n_existing;           /* < 0: seed the set with the pair's Optimal alignment like the drivers (aa_ali.cpp:83);
                                   >= 0: the caller's AlignmentSet already holds this many alignments (enumerate() appends) */
  const float* existing_scores; /* their scores (they take part in sortSet); such entries come back with n_pairs = -1
                                   and pair_off = their old index */
  uint32_t k_limit;             /* KSCW / CRCW: operations a branch node keeps (NOaliParams::k_limit, default 16); their user_limit
                                   is NOaliParams::user_limit (0 = the default 100000, noalib.cpp:20) */
  uint32_t sort_limit;          /* CRCW: operations a branch node sorts and follows (NOaliParams::sort_limit; 0 = the default 100) */
  float max_overlap;            /* CRCW: share of an accepted sub-path a later one may repeat (NOaliParams::max_overlap, default 0.30) */
} aln_noa;

/* One alignment as the enumerators return it (AlignedPairList, alignment.h:52-113). */
typedef struct {
  float score;
  float identity;               /* calcIdentity, alignment.h:856-865 */
  int32_t uid;
  int32_t n_pairs;
  int64_t pair_off;             /* into the caller's pairs buffer, in (q,t) int32 units of 2 */
} aln_alignment;

/* ---- context ---------------------------------------------------------------------------- */
/* stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream) or NULL for a private stream. */
int aln_ctx_create(int device_id, void* stream, aln_ctx** out);
void aln_ctx_destroy(aln_ctx* ctx);
const char* aln_error_string(int status);
const char* aln_last_error(const aln_ctx* ctx);
int aln_ctx_synchronize(aln_ctx* ctx);
/* 1 if the shared object this function lives in carries a gfx950 code object (the offload-bundle id
 * "hipv4-amdgcn-amd-amdhsa--gfx950" is looked up in the file itself; needs no GPU), else 0. */
int aln_has_gfx950(void);
/* Tuning / kernel-selection hints of ONE context.  A context reads its defaults from the environment once, when it is created
 * (ALN_NO_TAG_KERNEL, ALN_NO_H16, ALN_NO_KEY16, ALN_TAG_ALT_PRIO, ALN_TAG_SEGMENTS, ALN_DP_VARIANT="NW,R[,X]", ALN_EXACT_NO_TILES,
 * ALN_EXACT_LITERAL, ALN_EXACT_ALT_PRIO, ALN_SCORE_NO_PACKED, ALN_ENUM_NODE_CAP, ALN_TAG_LAG, ALN_TAG_SOLO, ALN_TAG_BITS, ALN_TAG_OCCUPANCY, ALN_PLANE_ROW_ALIGN, ALN_ENUM_POOL_RETRIES, ALN_ENUM_WAVES, ALN_ENUM_DEBUG, ALN_ENUM_KEEP_POOLS); launches never read the environment.  Keys:
 *   "tag_kernel" "h16" "key16"     1/0: tagged-key kernel / uint16 score plane / 16-bit key layout allowed (results identical)
 *   "tag_alt_prio"                  1/0: row-alternating wave priority in the tagged kernel (a scheduling hint; pays when launches
 *                                   follow each other on one stream, loses when launches of several contexts overlap);
 *                                   2, 3 and 0x100 | four 2-bit levels: experimental schedules (dp_affine_tag.hip, row loop)
 *   "tag_segments"                  tagged kernel: 0 (default) one workgroup per pair; K in 2..8: long pairs are cut into K row
 *                                   segments handed out by a device queue when the batch alone fills the GPU; -K: whenever pairs are long
 *   "tag_occupancy"                 tagged kernel: 2 or 3 waves per SIMD (two builds of the 16-cells-per-lane instantiations); 0 (default):
 *                                   by the waves per SIMD the launch alone brings (rounds of 3 against rounds of 2).  A caller
 *                                   that overlaps launches of several contexts sets 3 (bench.py: -8 % per step).
 *   "tag_bits"                      12: the 12-tag-bit key layout (pointer dialect 2, sequences up to 4096) also for shorter sequences;
 *                                   0 (default): by length.  Results do not depend on it.
 *   "tag_solo"                      1: templates of up to 2048 columns run in the one-wave-per-pair kernel (dp_affine_solo.hip: no
 *                                   barriers; wants >= 2048 pairs in flight, i.e. two 1024-pair launches on two streams)
 *   "dp_variant_nw" "dp_variant_r" "dp_variant_x"   force a row-sweep instantiation (0 = automatic)
 *   "exact_tiles" "exact_literal" "score_packed" "enum_node_cap" "tag_lag"
 *   "exact_wavefront"               tiled exact-order kernel: 1 (default) = 64-column tiles, the four waves of a pair on four row
 *                                   blocks, no workgroup barrier; 0 = 256-column tiles with a barrier per row.  Same planes.
 *   "exact_alt_prio"                tiled exact-order kernel, wave priorities: 2 (default) = by progress (a wave ahead of the
 *                                   launch's average yields), 1 = rotation over the resident waves, 0 = none
 *   "exact_prune"                   1 (default): far candidate chunks that provably cannot matter are skipped (bit-exact); 0: off
 *   "exact_debug"                   1: aln_batch_last_exact_stats reports tested / skipped far chunks; 2: the waves' run times
 *   "enum_waves"                    ConstrainedNearOptimal / UnconstrainedNearOptimal / KSConstrainedNearOptimal search: waves per pair
 *                                   (2..16, enumerate_par.hip);
 *                                   1 = the one-wave kernel; 0 (default) = 16.  Same sets, same order.
 *   "enum_debug"                    1: aln_batch_enumerate_all reports every group of pairs it searches (pools, times, retries) on stderr
 *   "enum_keep_pools"               1 (default): aln_batch_enumerate_all keeps its device pools with the batch until the batch is destroyed
 *                                   (allocating tens of GB costs seconds); 0: frees them when it returns
 *   "enum_pool_retries"             aln_batch_enumerate_all: how often a pair whose pools overflowed is searched again with four
 *                                   times the capacity (default 2)
 * Unknown key -> ALN_E_ARG.  No hint changes any result. */
int aln_ctx_set_hint(aln_ctx* ctx, const char* key, int64_t value);
int aln_ctx_get_hint(const aln_ctx* ctx, const char* key, int64_t* value);

/* ---- batch = many DPMatrix objects resident in HBM -------------------------------------- */
/* Replaces N x `DPMatrix(query, templ, ...)` construction up to initMtxMem (dpmatrix.h:250-259):
 * uploads both pools and the pair list (pair p aligns queries[q_idx[p]] with templates[t_idx[p]]).
 * score_only != 0 keeps no cell planes (all-vs-all scoring): only corner / best scores are produced. */
int aln_batch_create(aln_ctx* ctx, const aln_seqs* queries, const aln_seqs* templates,
                     int32_t n_pairs, const int32_t* q_idx, const int32_t* t_idx,
                     int32_t score_only, aln_batch** out);
void aln_batch_destroy(aln_batch* b);
int32_t aln_batch_n_pairs(const aln_batch* b);
/* bytes of HBM held by the batch (planes + sequences + results) */
int64_t aln_batch_device_bytes(const aln_batch* b);

/* DPMatrix::build (dpmatrix.h:291-317) for every pair: pre_calculate/SimilarityMatrix as `sim` says,
 * then the four builders (:356-1030) per direction and islocal = (align_type == local) (:155).
 * Asynchronous on the ctx stream.  bug_b4 != 0 reproduces dpmatrix.h:868 in reverse global builds. */
int aln_batch_dp(aln_batch* b, const aln_sim* sim, const aln_gap* gap,
                 int32_t direction, int32_t algo, int32_t bug_b4);
/* DPMatrix::reevaluate (dpmatrix.h:213-218): rebuild with the parameters of the last aln_batch_dp. */
int aln_batch_reevaluate(aln_batch* b);
/* Replace the gap description of the resident batch (constants, per-position arrays, deletion / insertion tables are uploaded
 * again; the similarity source stays resident); the next aln_batch_reevaluate builds with it.  The engine-side half of the
 * reference's refinement rounds — crcno.enumerate -> templ.updateCore -> dpm.reevaluate (gn2.cpp:146-185), where pre_calculate
 * derives new gap tables (gn2_eval.cpp:113-158) — for callers whose similarity does not change between rounds. */
int aln_batch_set_gap(aln_batch* b, const aln_gap* gap);
/* name of the DP kernel the last aln_batch_dp / _reevaluate launched ("dp_affine_tag_kernel<NW=2,R=2,X=8,local,h16,key16>",
 * "dp_affine_int_kernel<...>", "dp_exact_tiled_kernel<tpos,global,fwd>", "dp_exact_kernel<...>" ...) */
const char* aln_batch_dp_kernel_name(const aln_batch* b);

/* 7-argument DPMatrix ctor / build_subdpm (dpmatrix.h:169-189, :319-353) on one bounds rectangle per
 * pair: bounds[4*p..] = (q1_end, t1_end, q2_beg, t2_beg) in the order of the DEFINITION (B8). */
int aln_batch_dp_sub(aln_batch* b, const aln_sim* sim, const aln_gap* gap,
                     int32_t direction, const int32_t* bounds);

/* DPMatrix::getCell for a whole pair (dpmatrix.h:230-232): score, prev_query_idx, prev_template_idx
 * planes, Q x T row-major, untouched cells read 0 / -1 / -1 (dpmatrix.cpp:17-25).  Any pointer may be NULL. */
int aln_batch_get_cells(aln_batch* b, int32_t pair, float* score, int32_t* prev_q, int32_t* prev_t);
/* DPMatrix::getSim (dpmatrix.h:72-73) for a whole pair. */
int aln_batch_get_sim(aln_batch* b, int32_t pair, float* sim);
/* score of cell (Q-1,T-1) for forward, (0,0) for reverse builds, per pair (what global Optimal reports) */
int aln_batch_get_corner_scores(aln_batch* b, float* scores);

/* Optimal / Optimal_Rev ::enumerate (optimal.h:48-124, optimal_rev.h:44-131) for every pair:
 * find_max + pointer traceback on the device.  pairs: n_pairs x pair_stride x 2 int32, (q,t) in list
 * order; n[p] = list length; status[p] = 0 or ALN_E_STARTPAIR.  scores/n/status/pairs may be NULL. */
int aln_batch_optimal(aln_batch* b, float* scores, int32_t* n, int32_t* pairs, int32_t pair_stride,
                      int32_t* status);
/* The same, split so that a caller can keep the device busy: _enqueue launches find_max + traceback and the copy of the
 * per-pair results into a pinned host slot (two slots; ALN_E_STATE when both are waiting) and returns at once; _collect waits
 * for the OLDEST enqueued slot and hands out its scores / list lengths / status.  A loop `reevaluate; enqueue; collect(previous)`
 * overlaps the host's launch and copy latency of one step with the kernels of the next (bench.py). */
int aln_batch_optimal_enqueue(aln_batch* b);
int aln_batch_optimal_collect(aln_batch* b, float* scores, int32_t* n, int32_t* status);
/* Optimal + AlignmentSet::assignIdentity + SequenceGaps for every pair: what a driver prints for
 * `AlignmentSet as(dpm, optimal); as.assignIdentity(); cout << FastaOut(len) << as` (aa_ali.cpp:83-92, fastaio.h:51-76,
 * gstrings.h:84-164, alignment.h:856-865), without the FASTA framing.  Device: find_max + traceback, then one wave per pair lays
 * the template / query lines out and counts the identities (csrc/gapped_strings.hip); only the lines travel to the host.
 * tlines / qlines: n_pairs x stride chars, NUL-terminated (stride > T + Q covers any alignment); lengths[p] = line length
 * (0: status[p] != 0, or a list SequenceGaps cannot print).  scores/identity/status/lengths may be NULL.  Returns ALN_E_OVERFLOW when
 * a line does not fit the stride.  This is the end-to-end readout of config 2; aln_batch_optimal_enqueue/_collect is the score-only one. */
int aln_batch_optimal_strings(aln_batch* b, float* scores, float* identity, int32_t* status, char* tlines, char* qlines,
                              int32_t stride, int32_t* lengths);
/* The same, split like aln_batch_optimal_enqueue / _collect: _enqueue launches find_max + traceback + the string kernel on the
 * context's stream and the copy of the lines into one of two pinned slots on a separate copy stream, and returns at once
 * (ALN_E_STATE when both slots are waiting); _collect waits for the OLDEST slot and fills the caller's buffers (same `stride`).
 * `dp; strings_enqueue; strings_collect(previous)` overlaps step k's copy and host work with step k+1's kernels. */
int aln_batch_optimal_strings_enqueue(aln_batch* b, int32_t stride);
int aln_batch_optimal_strings_collect(aln_batch* b, float* scores, float* identity, int32_t* status, char* tlines, char* qlines,
                                      int32_t stride, int32_t* lengths);
/* Optimal_Subali::enumerate (optimal_subali.h:60-84) on the rectangles of the last aln_batch_dp_sub. */
int aln_batch_optimal_subali(aln_batch* b, float* scores, int32_t* n, int32_t* pairs,
                             int32_t pair_stride, int32_t* status);

/* ConstrainedNearOptimal / UnconstrainedNearOptimal ::enumerate (cw.h:68-92, ucw.h:64-85) for one pair
 * of the resident batch.  The set is seeded with the pair's Optimal alignment exactly like the drivers do
 * (aa_ali.cpp:83-89), the enumerator pushes its own uid-0 seed on top (B16), and the result is
 * sortSet(number_suboptimal)'ed.  flags: T bytes (SuboptFlags, sflags.h:23-37), ignored for UCW.
 * Outputs: up to max_alignments records + their pairs; *n_out = set size after sortSet. */
int aln_batch_enumerate(aln_batch* b, int32_t pair, const aln_noa* noa, const uint8_t* flags,
                        aln_alignment* out, int32_t max_alignments,
                        int32_t* pairs, int64_t pairs_capacity, int32_t* n_out);

/* The same enumeration for EVERY pair of the resident batch in one launch (BASELINE config 4: top-K near-optimal
 * tracebacks per pair from the GPU-resident matrices): one workgroup per pair runs its search (cw / ucw / kscw: 16 waves share
 * it, csrc/enumerate_par.hip; crcw: one wave), the host runs sortSet(number_suboptimal) per pair on (score,
 * index) keys in the reference's set order, survivors are unrolled on the device.  Every set is seeded with the pair's Optimal
 * alignment (noa->n_existing is ignored).
 *   flags: SuboptFlags rows, pair p's row at flags + p * flags_stride (flags_stride 0: one shared row of max T bytes);
 *   node_cap_per_pair / ali_cap_per_pair: trie nodes / alignments one pair's pools hold at first (0 = 1 Mi nodes / 64 Ki
 *   alignments).  A trie node is one aligned pair for the one-wave kernels and one diagonal run of up to 64 aligned pairs for
 *   the several-wave kernel, whose node pool is moreover shared by the pairs of a launch (n_pairs x node_cap_per_pair nodes in
 *   all: a 2000-residue homolog at DELTA_RATIO 0.01 needs 0.03-0.8 M of them).  Pairs that need more are searched again with 4 x larger pools, in groups sized to a device-memory budget,
 *   "enum_pool_retries" times (context hint, default 2), and only then report ALN_E_OVERFLOW;  K: slots per pair in the outputs (>= min(number_suboptimal, set size)).
 * Outputs (slot k of pair p at index p*K + k, set order): n_out[p] = set size after sortSet, scores, lengths,
 * pairs (NULL = not wanted) as (q,t) int32 at (p*K + k) * pair_stride * 2, status[p] = 0 / ALN_E_STARTPAIR /
 * ALN_E_OVERFLOW.  Returns the worst per-pair status. */
int aln_batch_enumerate_all(aln_batch* b, const aln_noa* noa, const uint8_t* flags, int32_t flags_stride,
                            uint32_t node_cap_per_pair, uint32_t ali_cap_per_pair, int32_t K,
                            int32_t* n_out, float* scores, int32_t* lengths, int32_t* pairs, int32_t pair_stride,
                            int32_t* status);
/* What the last aln_batch_enumerate_all used of every pair's pools: alignments created before sortSet (the reference's
 * as.size(), cw.h:91) and trie nodes (the several-wave kernel: runs, reserved 512 at a time by each wave) — also for pairs that reported ALN_E_OVERFLOW (the count at which they stopped), so that a
 * caller can size node_cap_per_pair / ali_cap_per_pair.  Either pointer may be NULL. */
int aln_batch_last_enum_usage(aln_batch* b, int32_t* alignments, int32_t* nodes);
/* Milliseconds the device spent in the search kernel / the unroll kernel of the last aln_batch_enumerate_all. */
int aln_batch_last_enum_ms(aln_batch* b, float* search_ms, float* unroll_ms);

/* ---- all-vs-all scoring without planes (BASELINE config 5) ------------------------------------ */
/* The score Optimal reports — find_max for local alignments (optimal.h:90-93,108-124), the final cell's score for the four
 * other align types (optimal.h:56-74) — for queries[q_begin..q_end) against EVERY template:
 * scores[(q - q_begin) * templates->n_seqs + t].  Replaces that many DPMatrix(q, t, AASubstitutionEval, fwd, align_type) +
 * Optimal constructions; nothing per cell is written to HBM.  A rank of a multi-GPU job calls it with its own block of query
 * rows.  ALN_GAP_AFFINE_CONST.  Integer table and gaps, templates up to 2046 residues: register-resident kernels
 * (local alignments whose values fit 15 bits run two queries per wave in packed 16-bit lanes).  Anything else the reference would
 * score — longer templates, fractional values — goes through resident batches of full builds (aln_batch_dp + Optimal) inside the
 * same call, ~12 GB of planes at a time; sequences beyond 65534 residues: ALN_E_TOO_LONG. */
int aln_score_all_vs_all(aln_ctx* ctx, const aln_seqs* queries, const aln_seqs* templates, const aln_submatrix* sub,
                         const aln_gap* gap, int32_t q_begin, int32_t q_end, float* scores);

/* ---- multi-GPU: a length-sorted deal of independent units + ONE collective (RCCL all-gather of scores) ----------- */
/* The reference is one thread, one DPMatrix at a time; pairs are independent (dpmatrix.h:104-111), so ranks own disjoint pair
 * lists and the only exchange is the final gather of the per-pair scores (BASELINE north_star). */
typedef struct aln_comm aln_comm;
#define ALN_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
/* Length-sorted deal: units sorted by work[u] (= Q*T of a pair, or the residues of a query row) descending, ties by index, dealt
 * over n_ranks in boustrophedon order (0..n-1, n-1..0, ...).  owner[u] = rank; slot[u] (may be NULL) = position of u in that
 * rank's local list, which keeps the sorted order (long units first).  Host arithmetic only. */
int aln_deal_units(const int64_t* work, int64_t n_units, int32_t n_ranks, int32_t* owner, int32_t* slot);
/* One process per GPU: rank 0 obtains an id, ships its ALN_COMM_ID_BYTES bytes to the other processes by the job's own
 * transport, and every process calls aln_comm_create(&its_ctx, 1, id, n_ranks, its_rank, ...).
 * One process driving n GPUs: aln_comm_create(ctxs, n, NULL, n, 0, ...) — or aln_ctx_create_multi, which also makes the contexts.
 * librccl.so is dlopen()ed on first use.  Collective calls run on each context's stream. */
int aln_comm_unique_id(void* id_out);
int aln_comm_create(aln_ctx* const* ctxs, int32_t n_local, const void* id, int32_t n_ranks, int32_t first_rank, aln_comm** out);
int aln_ctx_create_multi(const int32_t* device_ids, int32_t n, aln_ctx** ctxs_out, aln_comm** comm_out);
void aln_comm_destroy(aln_comm* comm);
int32_t aln_comm_n_ranks(const aln_comm* comm);
const char* aln_comm_last_error(const aln_comm* comm);
/* All ranks call it together.  Local rank k of this process contributes n_local[k] (<= n_max, the same n_max on every rank)
 * scores; global_index[k][e] is the position of local score e in the job's pair list.  On return global_out[0..n_total) holds
 * every contributed score on every rank.  One ncclAllGather of (index, score) records over xGMI. */
int aln_gather_scores(aln_comm* comm, const float* const* local_scores, const int32_t* const* global_index,
                      const int32_t* n_local, int32_t n_max, float* global_out, int64_t n_total);

/* ---- host-side helpers with no device work (alignment.h / gstrings.h) -------------------- */
/* AlignedPairList::calcIdentity (alignment.h:856-865). qstr/tstr include sentinels. */
float aln_identity(const char* qstr, int32_t Q, const char* tstr, int32_t T,
                   const int32_t* pairs, int32_t n_pairs);
/* SequenceGaps (gstrings.h:84-164, gstrings.cpp:17-29): gapped template line and one gapped query line
 * per alignment.  aln_gapped_length() gives the length of every line (without NUL).  qlines may be NULL
 * (template line only); tline may be NULL. */
int32_t aln_gapped_length(int32_t T, const aln_alignment* alis, int32_t n_alis, const int32_t* pairs);
int aln_gapped_strings(const char* qstr, int32_t Q, const char* tstr, int32_t T,
                       const aln_alignment* alis, int32_t n_alis, const int32_t* pairs,
                       char* tline, char* qlines, int32_t stride);

/* Hmap2Eval::pre_calculate (hmap2_eval.cpp:17-25): per template residue gap coefficients
 * t_gap_init = gap_init * Pi, t_gap_extn = gap_extn * Pi, Pi = exp(beta * (1 - 1.25 * p_coil)), p_coil = sse[3*i+2].
 * Host arithmetic with the host libm, like the reference; feeds aln_gap.t_gap_init / t_gap_extn. */
int aln_hmap2_gap_arrays(const float* t_sse, int64_t n, float gap_init, float gap_extn, float beta,
                         float* t_gap_init, float* t_gap_extn);

/* ---- measurement hooks ---------------------------------------------------------------------- */
/* Context hint "exact_debug" = 1: the exact-order tiled kernel (config 3) counts, per wave, the far candidate chunks it tested
 * against their bounds and the ones it could skip: out4 = {deletion chunks tested, skipped, insertion chunks tested, skipped}.
 * "exact_debug" = 2: the same four words hold how long the kernel's waves ran instead — {sum, longest, bitwise NOT of the
 * shortest, number of waves}, in units of 1024 s_memtime ticks — and the library lists the slowest pairs on stderr (that is how
 * a degenerate all-NaN pair was found to bound a 1024-pair launch).
 * (No reference counterpart: the reference scans every candidate, dpmatrix.h:453-480.) */
int aln_batch_last_exact_stats(const aln_batch* b, uint64_t* out4);
/* Milliseconds the device spent in the DP kernel(s) of the last aln_batch_dp, from HIP events recorded on
 * the ctx stream around those launches; synchronises the stream. */
int aln_batch_last_dp_ms(aln_batch* b, float* ms);
/* The same for up to max_n of the latest builds (ms[0] = the latest; the library keeps 64 event pairs), so that a pipelined
 * caller can read the kernel times after its loop instead of synchronising inside it.  Returns how many it wrote, -1 on error. */
int aln_batch_dp_ms_history(aln_batch* b, float* ms, int32_t max_n);
/* Bytes one DP launch MUST write to HBM with the plane layout the last aln_batch_dp chose: per cell of the Q x T matrices
 * the score element (fp32, or uint16 in local tagged builds) + the pointer element (32-bit, or 16-bit tagged words):
 * 8, 6 or 4 B/cell.  This is what a roofline fraction is computed from.  _contract_bytes is SURVEY.md 8(d)'s figure for the
 * reference's layout, 8 B per cell (fp32 score + 32-bit packed pointer), whatever was chosen; _plane_bytes_per_cell the factor. */
int64_t aln_batch_dp_algorithmic_bytes(const aln_batch* b);
int64_t aln_batch_dp_contract_bytes(const aln_batch* b);
int32_t aln_batch_plane_bytes_per_cell(const aln_batch* b);
int64_t aln_batch_cells(const aln_batch* b);      /* sum over pairs of |q|*|t| = (Q-2)(T-2) */

#ifdef __cplusplus
}
#endif
#endif /* ALN_HIP_H */
