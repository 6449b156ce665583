// optimal_rev.h — Optimal_Rev: traceback of a reverse-built matrix from (0,0) towards the tail (reference
// optimal_rev.h:23-133).  The reference class is abstract as shipped (its enumerate is const on a const matrix and
// does not override Enumerator::enumerate); this one is usable.
#ifndef ALN_HOST_OPTIMAL_REV_H
#define ALN_HOST_OPTIMAL_REV_H
#include "optimal.h"
// standard headers the reference's optimal_rev.h hands on to its includers
#include <iostream>

template <class S1, class S2, class Etype>
class Optimal_Rev : public Enumerator<S1, S2, Etype> {
 public:
  Optimal_Rev(align_t type = global) : islocal(type == local) {}
  int estimateSize() const { return 1; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) { aln::run_optimal(dpm, as, islocal, rev, false); }
 private:
  bool islocal;
};
#endif
