// optimal.h — Optimal: the single best traceback (reference optimal.h:22-124), run on the device by
// aln_batch_optimal (find_max with the seed rule + pointer traceback), returned as an AlignedPairList.
#ifndef ALN_HOST_OPTIMAL_H
#define ALN_HOST_OPTIMAL_H
#include <vector>
#include "alib.h"
#include "alignment.h"
#include "enumerator.h"
// standard headers the reference's optimal.h hands on to its includers
#include <iostream>

namespace aln {
// shared by Optimal / Optimal_Rev / Optimal_Subali: run the device traceback and append the list to `as`
template <class S1, class S2, class Etype>
void run_optimal(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as, bool enumerator_local, direction_t want, bool subali) {
  if (!subali && dpm.getDirection() != want) throw std::string("Enumerator direction does not match the DP matrix");
  if (!subali && enumerator_local != dpm.isLocal()) throw std::string("Enumerator align type does not match the DP matrix");
  const int stride = std::min(dpm.getQuerySize(), dpm.getTemplateSize()) + 3;
  std::vector<int32_t> pairs((size_t)stride * 2);
  float score = 0.f; int32_t n = 0, status = 0;
  int rc = subali ? aln_batch_optimal_subali(dpm.batch(), &score, &n, pairs.data(), stride, &status)
                  : aln_batch_optimal(dpm.batch(), &score, &n, pairs.data(), stride, &status);
  check(rc, default_ctx());
  check(status);                                   // "Illegal alignment start pair" (optimal.h:74)
  size_t k = as.size();
  as.resize(k + 1);
  as[k].score = score;
  for (int i = 0; i < n; ++i) as[k].append(pairs[2 * i], pairs[2 * i + 1]);
}
}  // namespace aln

template <class S1, class S2, class Etype>
class Optimal : public Enumerator<S1, S2, Etype> {
 public:
  Optimal(align_t type = global) : islocal(type == local) {}
  int estimateSize() const { return 1; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) { aln::run_optimal(dpm, as, islocal, fwd, false); }
 private:
  bool islocal;
};
#endif
