// aln_lowering.h — how a host-side Evaluator becomes the plain arrays the C ABI takes (include/aln_hip.h).
//
// The reference calls the evaluator's similarity/deletion/insertion per DP cell (dpmatrix.h:447-486).  Here
// DPMatrix::build() runs pre_calculate() once, then asks aln::Lowering<S1,S2,Etype> for
//   * a similarity source:  residue codes + substitution table | a materialised SimilarityMatrix | HMAP profiles
//   * a gap model from the closed set of include/aln_hip.h (AFFINE_CONST, AFFINE_TPOS_MIN, DEL_TABLE_INS_TPOS)
// and hands both to aln_batch_dp().  The evaluator families of the reference (AASubstitutionEval, Hmap2Eval /
// HMAPaliEval, Gn2Eval) have specialisations below / in hmap_eval.h / gn2_eval.h.  Any other evaluator works UNCHANGED through
// the generic path: its similarity() + post_process() are evaluated once per cell on the host into a SimilarityMatrix plane
// and its deletion()/insertion() are tabulated (ALN_GAP_TABLES).  An evaluator that knows its gaps are one of the closed
// forms can say so and get the fast kernels:
//
//     void aln_describe_gaps(const S1& q, const S2& t, aln::GapDescription& g) const;
#ifndef ALN_HOST_LOWERING_H
#define ALN_HOST_LOWERING_H
#include <string>
#include <vector>

#include "aasubalib.h"
#include "aln_hip.h"
#include "alib.h"
#include "evaluator.h"
#include "simmatrix.h"

namespace aln {

struct GapDescription {
  int model;                    // ALN_GAP_AFFINE_CONST or ALN_GAP_AFFINE_TPOS_MIN
  int align_type;               // the align_t the evaluator's free-end rules follow
  float gap_init, gap_extn;     // AFFINE_CONST
  std::vector<float> t_gap_init, t_gap_extn;   // AFFINE_TPOS_MIN, one per template position (sentinels included)
  GapDescription() : model(ALN_GAP_AFFINE_CONST), align_type(semi_local), gap_init(0.f), gap_extn(0.f) {}
};

// Everything aln_batch_dp needs, with the storage that backs the pointers.
struct Lowered {
  aln_sim sim;
  aln_gap gap;
  GapDescription gd;
  std::string alphabet;
  std::vector<float> table;
  std::vector<float> plane;
  int64_t plane_off0;
  std::vector<float> q_aa, q_sse, q_conf, t_aa, t_sse, t_conf;
  std::vector<float> t_gap_cn, del_table;     // ALN_GAP_DEL_TABLE_INS_TPOS (Gn2Eval)
  std::vector<float> ins_tables;              // ALN_GAP_TABLES (any plugin)
  int64_t del_off0, ins_off0;
  Lowered() : plane_off0(0), del_off0(0), ins_off0(0) { sim = aln_sim(); gap = aln_gap(); }
  void finish_gap() {
    gap.model = gd.model; gap.align_type = gd.align_type; gap.gap_init = gd.gap_init; gap.gap_extn = gd.gap_extn;
    gap.t_gap_init = gd.t_gap_init.empty() ? 0 : gd.t_gap_init.data();
    gap.t_gap_extn = gd.t_gap_extn.empty() ? 0 : gd.t_gap_extn.data();
  }
};

// does the evaluator name a closed-form gap model itself?
template <class E>
struct has_describe_gaps {
  template <class U> static char test(decltype(&U::aln_describe_gaps));
  template <class U> static long test(...);
  enum { value = sizeof(test<E>(0)) == sizeof(char) };
};
template <bool B> struct bool_tag {};

// generic path: SimilarityMatrix on the host, then either the evaluator's own gap description (aln_describe_gaps) or — for a
// plugin written against evaluator.h and nothing else — its deletion()/insertion() tabulated (ALN_GAP_TABLES):
//   deletion(q,t,1,2,t1,t2) for every t1 < t2; insertion(q,t,q1,q2,t1,t1+1) for interior query positions by distance, for
//   q1 = head and for q2 = tail.  The one assumption — away from the query ends insertion() depends on q2-q1, not on q1 — is
//   checked on a second offset and throws if the evaluator violates it.  (All evaluators of the reference satisfy it.)
template <class S1, class S2, class Etype>
struct Lowering {
  static void lower(const S1& q, const S2& t, const Etype& e, Lowered& L) {
    SimilarityMatrix sm(q, t, static_cast<const Evaluator<S1, S2, Etype>&>(e));
    L.plane.assign(sm.data(), sm.data() + sm.size());
    L.sim.kind = ALN_SIM_MATRIX;
    L.sim.planes = L.plane.data();
    L.plane_off0 = 0;
    L.sim.plane_off = &L.plane_off0;
    gaps(q, t, e, L, bool_tag<has_describe_gaps<Etype>::value>());
  }
  static void gaps(const S1& q, const S2& t, const Etype& e, Lowered& L, bool_tag<true>) {
    e.aln_describe_gaps(q, t, L.gd);
    L.finish_gap();
  }
  static void gaps(const S1& q, const S2& t, const Etype& e, Lowered& L, bool_tag<false>) {
    const int Q = (int)q.size(), T = (int)t.size();
    const int qa = Q >= 3 ? 1 : 0;                          // the reference passes (i-1, i) as the query positions of a deletion
    L.del_table.assign((size_t)T * T, 0.f);
    for (int t1 = 0; t1 < T; ++t1)
      for (int t2 = t1 + 1; t2 < T; ++t2) L.del_table[(size_t)t1 * T + t2] = e.deletion(q, t, qa, qa + 1, t1, t2);
    L.ins_tables.assign((size_t)3 * T * Q, 0.f);
    float* in0 = L.ins_tables.data();
    float* in1 = in0 + (size_t)T * Q;
    float* in2 = in1 + (size_t)T * Q;
    for (int t1 = 0; t1 + 1 < T; ++t1) {
      for (int d = 1; 1 + d <= Q - 2; ++d) in0[(size_t)t1 * Q + d] = e.insertion(q, t, 1, 1 + d, t1, t1 + 1);
      for (int q2 = 1; q2 < Q; ++q2) in1[(size_t)t1 * Q + q2] = e.insertion(q, t, 0, q2, t1, t1 + 1);
      for (int q1 = 0; q1 < Q - 1; ++q1) in2[(size_t)t1 * Q + q1] = e.insertion(q, t, q1, Q - 1, t1, t1 + 1);
      for (int d = 1; 2 + d <= Q - 2; d += 3)               // translation check on a second offset
        if (e.insertion(q, t, 2, 2 + d, t1, t1 + 1) != in0[(size_t)t1 * Q + d])
          throw std::string("Evaluator::insertion depends on the query position: it cannot be tabulated for the device");
    }
    L.gd.model = ALN_GAP_TABLES;
    L.gd.align_type = global;                               // unused by this model: the tables carry the end rules
    L.finish_gap();
    L.del_off0 = 0; L.ins_off0 = 0;
    L.gap.del_table = L.del_table.data(); L.gap.del_table_off = &L.del_off0;
    L.gap.ins_tables = L.ins_tables.data(); L.gap.ins_table_off = &L.ins_off0;
  }
};

// AASubstitutionEval: codes + table, constant affine gaps (aasubalib.h:17-77)
template <class S1, class S2>
struct Lowering<S1, S2, AASubstitutionEval<S1, S2> > {
  static void lower(const S1&, const S2&, const AASubstitutionEval<S1, S2>& e, Lowered& L) {
    const SubstitutionMatrix* m = e.subMatrix();
    L.alphabet = m->letters();
    const size_t n = L.alphabet.size();
    L.table.assign(m->table(), m->table() + n * n);
    L.sim.kind = ALN_SIM_SUBMATRIX;
    L.sim.sub.n = (int32_t)n;
    L.sim.sub.alphabet = L.alphabet.c_str();
    L.sim.sub.table = L.table.data();
    const AliParams* p = e.aliParams();
    if (p->align_type < 0 || p->align_type > 4) throw std::string("Illegal gap style");
    L.gd.model = ALN_GAP_AFFINE_CONST;
    L.gd.align_type = p->align_type;
    L.gd.gap_init = p->gap_init_penalty;
    L.gd.gap_extn = p->gap_extn_penalty;
    L.finish_gap();
  }
};

inline void check(int rc, aln_ctx* ctx = 0) {
  if (rc == ALN_OK) return;
  std::string msg = aln_error_string(rc);
  if (rc == ALN_E_HIP && ctx) msg += std::string(": ") + aln_last_error(ctx);
  throw msg;                   // the reference throws std::string everywhere (dpmatrix.h:361, optimal.h:74, ...)
}
// the process-wide context the host classes use (device 0 unless ALN_DEVICE is set)
inline aln_ctx* default_ctx() {
  static aln_ctx* ctx = 0;
  if (!ctx) {
    const char* d = getenv("ALN_DEVICE");
    check(aln_ctx_create(d ? atoi(d) : 0, 0, &ctx));
  }
  return ctx;
}

}  // namespace aln
#endif
