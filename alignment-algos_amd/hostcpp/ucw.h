// ucw.h — UnconstrainedNearOptimal: branching allowed at every cell (reference ucw.h:25-236).  See cw.h.
#ifndef ALN_HOST_UCW_H
#define ALN_HOST_UCW_H
#include "cw.h"
// standard headers the reference's ucw.h hands on to its includers
#include <string>
using namespace std;   // as the reference's ucw.h does at header scope: sources written against it name string, vector, cerr ... unqualified

template <class S1, class S2, class Etype>
class UnconstrainedNearOptimal : public Enumerator<S1, S2, Etype> {
 public:
  typedef AlignedPairList<S1, S2> SingleAlignment;
  typedef AlignedPair<S1, S2> SinglePair;
  explicit UnconstrainedNearOptimal(const NOaliParams& p) : user_limit(100000), params(&p) {}
  unsigned int user_limit;
  int estimateSize() const { return params->number_suboptimal; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) {
    user_limit = 100000;                                // ucw.h:72
    aln::run_enumeration(ALN_ENUM_UCW, *params, (const unsigned char*)0, dpm, as, user_limit);
  }
 private:
  const NOaliParams* params;
};
#endif
