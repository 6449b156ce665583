// simmatrix.h — host-side similarity matrix: zero borders, evaluator.similarity() inside, then post_process
// (reference simmatrix.h:18-72).  Used by the generic lowering path and by evaluators' post_process hooks.
#ifndef ALN_HOST_SIMMATRIX_H
#define ALN_HOST_SIMMATRIX_H
#include "evaluator.h"
#include "matrix.h"

class SimilarityMatrix : public matrix<float> {
 public:
  SimilarityMatrix(int rows, int cols) : matrix<float>(rows, cols) {}
  template <class S1, class S2, class Etype>
  SimilarityMatrix(const S1& qs, const S2& ts, const Evaluator<S1, S2, Etype>& eval) : matrix<float>((int)qs.size(), (int)ts.size()) {
    const int ql = rows() - 1, tl = cols() - 1;
    for (int i = 0; i <= ql; ++i) for (int j = 0; j <= tl; ++j) (*this)(i, j) = 0.f;
    for (int i = 1; i < ql; ++i) for (int j = 1; j < tl; ++j) (*this)(i, j) = eval.similarity(qs, ts, i, j);
    eval.post_process(*this);
  }
};
#endif
