// kscw.h — KSConstrainedNearOptimal: constrained enumeration whose branch nodes keep only their k_limit best operations
// (reference kscw.h:26-351), run on the device-resident matrix by aln_batch_enumerate(ALN_ENUM_KSCW).  Same calling
// pattern as cw.h.  The reference header itself does not compile on LP64 (kscw.h:188), so this class is checked against
// the oracle's restatement only (parity unpinned); nalign2 reaches it with -kscw (nalign2.cpp:100-113).
#ifndef ALN_HOST_KSCW_H
#define ALN_HOST_KSCW_H
#include "cw.h"
// standard headers the reference's kscw.h hands on to its includers
#include <string>
using namespace std;   // as the reference's kscw.h does at header scope: sources written against it name string, vector, cerr ... unqualified

template <class S1, class S2, class Etype>
class KSConstrainedNearOptimal : public Enumerator<S1, S2, Etype> {
 public:
  typedef AlignedPairList<S1, S2> SingleAlignment;
  typedef AlignedPair<S1, S2> SinglePair;
  KSConstrainedNearOptimal(const NOaliParams& p, const SuboptFlags& f) : params(&p), subopt(&f) {}
  int estimateSize() const { return params->number_suboptimal; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) {
    if ((int)subopt->size() != dpm.getTemplateSize()) throw std::string("SuboptFlags length differs from the template");
    aln::run_enumeration(ALN_ENUM_KSCW, *params, subopt->data(), dpm, as, params->user_limit);   // params->user_limit, :168
    std::cerr << "Ali#=" << as.size() << std::endl;                                             // :134
  }
 private:
  const NOaliParams* params;
  const SuboptFlags* subopt;
};
#endif
