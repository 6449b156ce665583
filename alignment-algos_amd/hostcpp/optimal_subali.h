// optimal_subali.h — Optimal_Subali: traceback inside a sub-rectangle build from (q2_beg,t2_beg) back to
// (q1_end,t1_end) (reference optimal_subali.h:23-84).
#ifndef ALN_HOST_OPTIMAL_SUBALI_H
#define ALN_HOST_OPTIMAL_SUBALI_H
#include "optimal.h"
// standard headers the reference's optimal_subali.h hands on to its includers
#include <iostream>

template <class S1, class S2, class Etype>
class Optimal_Subali : public Enumerator<S1, S2, Etype> {
 public:
  Optimal_Subali(int q1, int t1, int q2, int t2) : q1_end(q1), t1_end(t1), q2_beg(q2), t2_beg(t2) {}
  int estimateSize() const { return 1; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) {
    if (!dpm.isSub()) throw std::string("Optimal_Subali needs a sub-rectangle DP matrix");
    aln::run_optimal(dpm, as, false, fwd, true);
  }
 private:
  int q1_end, t1_end, q2_beg, t2_beg;
};
#endif
