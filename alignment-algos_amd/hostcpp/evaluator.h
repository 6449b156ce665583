// evaluator.h — the scoring plugin surface, unchanged from the reference (evaluator.h:20-147): a CRTP base whose
// five methods forward to the derived class without virtual calls.
//   similarity(q,t,i,j)                     match score of query position i with template position j
//   deletion(q,t,q1,q2,t1,t2)               penalty for skipping template positions t1+1..t2-1 (q1->t1, q2->t2 aligned)
//   insertion(q,t,q1,q2,t1,t2)              penalty for extra query positions q1+1..q2-1
//   pre_calculate(q,t)                      called once before the similarity matrix is built
//   post_process(SimilarityMatrix&)         called once after it is built
// On this engine the DP does not call these per cell: DPMatrix lowers the evaluator to device arrays through
// aln::Lowering<Etype> (aln_lowering.h); evaluators of the reference's families lower automatically.
#ifndef ALN_HOST_EVALUATOR_H
#define ALN_HOST_EVALUATOR_H
#include <string>

class SimilarityMatrix;

template <class S1, class S2, class Etype>
class Evaluator {
 public:
  float similarity(const S1& q, const S2& t, int qi, int ti) const { return Derived().similarity(q, t, qi, ti); }
  float deletion(const S1& q, const S2& t, int q1, int q2, int t1, int t2) const { return Derived().deletion(q, t, q1, q2, t1, t2); }
  float insertion(const S1& q, const S2& t, int q1, int q2, int t1, int t2) const { return Derived().insertion(q, t, q1, q2, t1, t2); }
  void post_process(SimilarityMatrix& s) const { Derived().post_process(s); }
  void pre_calculate(const S1& q, const S2& t) const { Derived().pre_calculate(q, t); }
  const Etype& Derived() const { return static_cast<const Etype&>(*this); }
 protected:
  Evaluator() {}
};
#endif
