// aasubalib.h — substitution-matrix evaluator with constant affine gaps and per-mode free end gaps.
// Same class and semantics as the reference's AASubstitutionEval (aasubalib.h:8-87); on this engine it lowers
// to (residue codes + table, ALN_GAP_AFFINE_CONST) and never runs per cell on the host.
#ifndef ALN_HOST_AASUBALIB_H
#define ALN_HOST_AASUBALIB_H
#include <string>
#include "alib.h"
#include "evaluator.h"
#include "sequence.h"
#include "submatrix.h"

template <class S1, class S2>
class AASubstitutionEval : public Evaluator<S1, S2, AASubstitutionEval<S1, S2> > {
 public:
  AASubstitutionEval(AliParams& p, SubstitutionMatrix& m) : params(&p), sub_matrix(&m) {}
  float similarity(const S1& q, const S2& t, int qi, int ti) const {
    if (q[qi]->isHead() || q[qi]->isTail() || t[ti]->isHead() || t[ti]->isTail()) return 0.f;
    return sub_matrix->score(q[qi]->olc, t[ti]->olc);
  }
  float deletion(const S1&, const S2& t, int, int, int t1, int t2) const {
    return gap(t2 - t1 - 1, free_del() && (t[t1]->isHead() || t[t2]->isTail()));
  }
  float insertion(const S1& q, const S2&, int q1, int q2, int, int) const {
    return gap(q2 - q1 - 1, free_ins() && (q[q1]->isHead() || q[q2]->isTail()));
  }
  void pre_calculate(const S1&, const S2&) const {}
  void post_process(SimilarityMatrix&) const {}
  const AliParams* aliParams() const { return params; }
  const SubstitutionMatrix* subMatrix() const { return sub_matrix; }
 private:
  bool free_del() const { check(); return params->align_type == local || params->align_type == semi_local || params->align_type == local_global; }
  bool free_ins() const { check(); return params->align_type == local || params->align_type == semi_local || params->align_type == global_local; }
  void check() const { if (params->align_type < 0 || params->align_type > 4) throw std::string("Illegal gap style"); }
  float gap(int len, bool free_end) const {
    if (len < 1) return 0.f;
    if (free_end) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * (len - 1);
  }
  AliParams* params;
  SubstitutionMatrix* sub_matrix;
};
#endif
