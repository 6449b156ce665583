/* TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference algorithm (christang/alignment-algos) for the hot
 * path named by BASELINE.json: DPMatrix build -> Optimal traceback -> near-optimal
 * enumeration -> gapped strings, behind plain-array arguments.  Every function cites
 * the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (libalnhip.so) never does.
 *
 * Pinning: the AA/substitution-matrix path is checked against the real reference
 * compiled in place (oracle/_ref/ref_harness) and against tests/golden/ fixtures that
 * harness produced (oracle/gen_golden.py).  The profile (Hmap2Eval) arithmetic cannot
 * be built from the reference (it needs the absent Troll library): its math primitives
 * (hmath.h) and the DP with position-dependent gaps are pinned through
 * oracle/_ref/ref_profile (real hmath.h + real DPMatrix driven by a plugin evaluator),
 * the glue formula of hmap2_eval.h:27-95 itself is "parity unpinned".
 *
 * Conventions: matrices are row-major Q x T (Q = |query|+2, T = |template|+2, both
 * including the '^' head and '$' tail sentinels, sequence.cpp:15-16).  Head = index 0,
 * tail = index size-1.  All arithmetic is IEEE fp32 in the reference's operation order
 * (built with -ffp-contract=off, no -ffast-math).
 */
#ifndef ALN_ORACLE_H
#define ALN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* align_t, alib.h:20-26 */
enum { ORC_GLOBAL_LOCAL = 0, ORC_GLOBAL = 1, ORC_LOCAL_GLOBAL = 2, ORC_LOCAL = 3, ORC_SEMI_LOCAL = 4 };
/* direction_t, dpmatrix.h:23-26 */
enum { ORC_FWD = 1, ORC_REV = 2 };
/* gap models */
enum { ORC_GAP_AFFINE_CONST = 0,    /* AASubstitutionEval, aasubalib.h:27-77 */
       ORC_GAP_AFFINE_TPOS_MIN = 1, /* Hmap2Eval / HMAPaliEval, hmap2_eval.h:41-95 */
       ORC_GAP_CALLBACK = 3,        /* any deletion(q1,q2,t1,t2) / insertion(q1,q2,t1,t2) supplied by the test (arbitrary plugins) */
       ORC_GAP_GN2 = 2              /* Gn2Eval, gn2_eval.h:100-165 — parity UNPINNED (its inputs come from the absent Troll library) */ };

typedef struct {
  int model;
  int align_type;
  float gi, ge;            /* model 0 */
  const float* tgi;        /* model 1: per template position, length T; model 2: Gn2Eval's v_gi / v_ge */
  const float* tge;
  /* model 2 (gn2_eval.cpp:113-158): v_cn per template position; distance, vv_gi, vv_ge, vv_cd as T x T arrays [p2*T + p1] */
  const float* tcn;
  const float* dist;
  const float* vvgi;
  const float* vvge;
  const float* vvcd;
  /* model 3: the test's own gap functions */
  float (*del_cb)(int q1, int q2, int t1, int t2);
  float (*ins_cb)(int q1, int q2, int t1, int t2);
} orc_gap;

/* error codes mirror the reference's throw sites */
enum { ORC_OK = 0,
       ORC_E_BOUNDS = -1,       /* "Illegal bounds building DPM"  dpmatrix.h:361,544,699,885 */
       ORC_E_GAPSTYLE = -2,     /* "Illegal gap style"            aasubalib.h:46,72 */
       ORC_E_STARTPAIR = -3,    /* "Illegal alignment start pair" optimal.h:74 */
       ORC_E_RESIDUE = -4,      /* residue outside the matrix alphabet (UB in submatrix.h:36-38) */
       ORC_E_ARG = -5 };

float orc_deletion(const orc_gap* g, int Q, int T, int q1, int q2, int t1, int t2, int* err);
float orc_insertion(const orc_gap* g, int Q, int T, int q1, int q2, int t1, int t2, int* err);

/* SimilarityMatrix for AASubstitutionEval: simmatrix.h:51-72 + aasubalib.h:17-25.
 * qres/tres are the full strings including '^' and '$'.  table is n x n row-major over `alphabet`. */
int orc_sim_submatrix(int Q, int T, const char* qres, const char* tres,
                      const char* alphabet, int n, const float* table, float* S);

/* Hmap2Eval::similarity (hmap2_eval.h:27-39) over plain profile arrays, borders zero
 * (simmatrix.h:58-66); q_aa: Q x 20, q_sse: Q x 3, q_conf: Q. */
int orc_sim_hmap2(int Q, int T, const float* q_aa, const float* q_sse, const float* q_conf,
                  const float* t_aa, const float* t_sse, const float* t_conf, float alpha, float* S);
/* Hmap2Eval::post_process (hmap2_eval.h:98-101; hmath.h:43-92): z-normalise the interior then add -zero_shift */
int orc_norm_shift(int Q, int T, float* S, float zero_shift);
/* Hmap2Eval::pre_calculate (hmap2_eval.cpp:17-25): per template position gi/ge from p_coil */
void orc_hmap2_precalc(int T, const float* t_pcoil, float gi, float ge, float beta, float* tgi, float* tge);
/* hmath.h primitives, exposed for pinning */
float orc_dot(const float* a, const float* b, int n);       /* hmath.h:18-26 */
float orc_pearson(const float* a, const float* b, int n);   /* hmath.h:94-103 */

/* DPMatrix init (dpmatrix.cpp:17-25): score 0, prev -1 */
void orc_dp_init(int Q, int T, float* D, int* PQ, int* PT);
/* DPMatrix::build / build_subdpm (dpmatrix.h:291-353) dispatching to the four builders
 * (:356-1030).  Full build: q0=t0=0, q1=Q-1, t1=T-1.  bug_b4 reproduces dpmatrix.h:868. */
int orc_dp_build(int Q, int T, const float* S, const orc_gap* gap, int direction, int islocal,
                 int q0, int q1, int t0, int t1, int bug_b4, float* D, int* PQ, int* PT);

/* Optimal::enumerate / enumerate_local / find_max (optimal.h:48-124).
 * pairs: out, 2*(Q+T) ints as (q,t) in list order; returns ORC_E_STARTPAIR like the throw. */
int orc_optimal(int Q, int T, const float* D, const int* PQ, const int* PT, int islocal,
                int* pairs, int* npairs, float* score);
/* Optimal_Rev (optimal_rev.h:44-131) */
int orc_optimal_rev(int Q, int T, const float* D, const int* PQ, const int* PT, int islocal,
                    int* pairs, int* npairs, float* score);
/* Optimal_Subali (optimal_subali.h:60-84) */
int orc_optimal_subali(int Q, int T, const float* D, const int* PQ, const int* PT,
                       int q1_end, int t1_end, int q2_beg, int t2_beg,
                       int* pairs, int* npairs, float* score);

/* Alignment sets (alignment.h:876-949) as an opaque handle */
typedef struct orc_set orc_set;
orc_set* orc_set_new(void);
void orc_set_free(orc_set*);
int orc_set_size(const orc_set*);
/* push an alignment given in list order */
void orc_set_push(orc_set*, const int* pairs, int npairs, float score, int uid);
int orc_set_npairs(const orc_set*, int k);
void orc_set_get(const orc_set*, int k, int* pairs, float* score, float* identity, int* uid);
/* AlignmentSet::sortSet (alignment.h:922-932): std::sort / std::partial_sort, "higher score first" */
void orc_set_sort(orc_set*, int max);
/* AlignmentSet::assignIdentity (alignment.h:942-949, :856-865) */
void orc_set_identity(orc_set*, const char* qstr, const char* tstr);

/* ConstrainedNearOptimal (cw.h:68-284; kind=0, flags = T bytes 0/1) and
 * UnconstrainedNearOptimal (ucw.h:64-236; kind=1, flags ignored).  Appends to `as`
 * exactly like the reference (seed alignment uid 0 pushed on top of what is there), then sortSet. */
int orc_enumerate(int kind, int Q, int T, const float* D, const int* PQ, const int* PT,
                  const float* S, const orc_gap* gap, const unsigned char* flags,
                  int number_suboptimal, float delta_ratio, unsigned user_limit, orc_set* as);

/* KSConstrainedNearOptimal (kscw.h:109-351): per branch node only the k_limit best operations continue (std::sort /
 * std::partial_sort on the operation scores), limits halve down the tree.  Parity unpinned (the header does not build on LP64). */
int orc_enumerate_ks(int Q, int T, const float* D, const int* PQ, const int* PT, const float* S, const orc_gap* gap,
                     const unsigned char* flags, int number_suboptimal, float delta_ratio, unsigned k_limit, unsigned user_limit,
                     orc_set* as);

/* CRConstrainedNearOptimal (crcw.h:134-594): operations sorted per branch node (at most sort_limit kept), each followed along the
 * stored pointers to the end of the current flag region, operations sharing more than max_overlap of an accepted better one's
 * sub-path dropped, at most the node's limit survive.  Parity unpinned (the header does not build on LP64).  *oob_reads: how
 * often the reference would have read regions[-1] (crcw.h:387); the restatement gives that read its own state class. */
int orc_enumerate_cr(int Q, int T, const float* D, const int* PQ, const int* PT, const float* S, const orc_gap* gap,
                     const unsigned char* flags, int number_suboptimal, float delta_ratio, unsigned k_limit, unsigned sort_limit,
                     unsigned user_limit, float max_overlap, orc_set* as, long* oob_reads);

/* SequenceGaps (gstrings.h:84-164, gstrings.cpp:17-29): template line and one query line
 * per alignment.  Every line is orc_gapped_len() chars (+NUL); qlines is n x stride, stride > that. */
int orc_gapped_len(const orc_set* as, int T);
int orc_gapped_strings(const orc_set* as, int Q, int T, const char* qstr, const char* tstr,
                       char* tline, char* qlines, int stride);

#ifdef __cplusplus
}
#endif
#endif
