// crcw.h — CRConstrainedNearOptimal: constrained enumeration with "controlled redundancy" (reference crcw.h:26-594): a branch
// node follows its sort_limit best operations to the end of the template's flag region and keeps at most k_limit of them whose
// sub-paths do not repeat more than max_overlap of a better one's; run on the device-resident matrix by
// aln_batch_enumerate(ALN_ENUM_CRCW).  Same calling pattern as cw.h / kscw.h.  The reference header itself does not compile
// on LP64 (crcw.h:242), so this class is checked against the oracle's restatement only (parity unpinned); nalign2 and gn2
// reach it with -crcw (nalign2.cpp:114-130, gn2.cpp:139-185).
#ifndef ALN_HOST_CRCW_H
#define ALN_HOST_CRCW_H
#include "cw.h"
// standard headers the reference's crcw.h hands on to its includers
#include <string>
#include <iomanip>
using namespace std;   // as the reference's crcw.h does at header scope: sources written against it name string, vector, cerr ... unqualified

template <class S1, class S2, class Etype>
class CRConstrainedNearOptimal : public Enumerator<S1, S2, Etype> {
 public:
  typedef AlignedPairList<S1, S2> SingleAlignment;
  typedef AlignedPair<S1, S2> SinglePair;
  CRConstrainedNearOptimal(const NOaliParams& p, const SuboptFlags& f) : params(&p), subopt(&f) {}
  int estimateSize() const { return params->number_suboptimal; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) {
    if ((int)subopt->size() != dpm.getTemplateSize()) throw std::string("SuboptFlags length differs from the template");
    aln::run_enumeration(ALN_ENUM_CRCW, *params, subopt->data(), dpm, as, params->user_limit);   // params->user_limit, crcw.h:224
    std::cerr << "Number of alignments after sorting: " << as.size() << "." << std::endl;
  }
 private:
  const NOaliParams* params;
  const SuboptFlags* subopt;
};
#endif
