// cw.h / ucw.h — near-optimal alignment enumerators (reference cw.h:26-284, ucw.h:25-236) run on the device-resident
// matrix by aln_batch_enumerate.  enumerate() APPENDS to whatever the set already holds (the drivers put the Optimal
// alignment there first, aa_ali.cpp:83), pushes its own uid-0 seed on top (cw.h:82-83) and finally
// sortSet(number_suboptimal)s the whole set: the engine gets the existing scores, sorts keys with the same
// std::sort / std::partial_sort, and the set is rebuilt in that order.
#ifndef ALN_HOST_CW_H
#define ALN_HOST_CW_H
#include <vector>
#include "alignment.h"
#include "enumerator.h"
#include "noalib.h"
#include "sflags.h"
// standard headers the reference's cw.h hands on to its includers
#include <algorithm>
#include <string>
using namespace std;   // as the reference's cw.h does at header scope: sources written against it name string, vector, cerr ... unqualified

namespace aln {
template <class S1, class S2, class Etype>
void run_enumeration(int kind, const NOaliParams& params, const unsigned char* flags, DPMatrix<S1, S2, Etype>& dpm,
                     AlignmentSet<S1, S2, Etype>& as, unsigned int user_limit) {
  typedef AlignedPairList<S1, S2> Alignment;
  const int n_existing = (int)as.size();
  std::vector<float> ex(std::max(n_existing, 1));
  for (int k = 0; k < n_existing; ++k) ex[k] = as[k].score;
  aln_noa noa = aln_noa();
  noa.kind = kind;
  noa.number_suboptimal = params.number_suboptimal;
  noa.delta_ratio = params.delta_ratio;
  noa.user_limit = user_limit;
  noa.n_existing = n_existing;
  noa.existing_scores = ex.data();
  noa.k_limit = params.k_limit;
  noa.sort_limit = params.sort_limit;
  noa.max_overlap = params.max_overlap;
  const int per = std::min(dpm.getQuerySize(), dpm.getTemplateSize()) + 3;
  int32_t cap = std::max(params.number_suboptimal, 1) + n_existing + 1, n_out = 0;
  std::vector<aln_alignment> out;
  std::vector<int32_t> pairs;
  int rc;
  for (int attempt = 0;; ++attempt) {                   // sortSet(max <= 0) keeps everything: grow once to the reported size
    out.assign(cap, aln_alignment());
    pairs.assign((size_t)cap * per * 2, 0);
    rc = aln_batch_enumerate(dpm.batch(), 0, &noa, flags, out.data(), cap, pairs.data(), (int64_t)cap * per, &n_out);
    if (rc == ALN_E_OVERFLOW && n_out > cap && attempt == 0) { cap = n_out; continue; }
    break;
  }
  check(rc, default_ctx());
  std::vector<Alignment> sorted((size_t)n_out);
  for (int k = 0; k < n_out; ++k) {
    if (out[k].n_pairs < 0) { sorted[k] = as[(size_t)out[k].pair_off]; continue; }     // one of the caller's own alignments
    Alignment& a = sorted[k];
    a.score = out[k].score;
    a.uid = out[k].uid;
    const int32_t* p = pairs.data() + 2 * out[k].pair_off;
    for (int i = 0; i < out[k].n_pairs; ++i) a.append(p[2 * i], p[2 * i + 1]);
  }
  as.assign(sorted.begin(), sorted.end());
}
}  // namespace aln

template <class S1, class S2, class Etype>
class ConstrainedNearOptimal : public Enumerator<S1, S2, Etype> {
 public:
  typedef AlignedPairList<S1, S2> SingleAlignment;
  typedef AlignedPair<S1, S2> SinglePair;
  ConstrainedNearOptimal(const NOaliParams& p, const SuboptFlags& f) : user_limit(1000000), params(&p), subopt(&f) {}
  unsigned int user_limit;
  int estimateSize() const { return params->number_suboptimal; }
  void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) {
    user_limit = 1000000;                               // hard-wired in the reference (cw.h:76)
    if ((int)subopt->size() != dpm.getTemplateSize()) throw std::string("SuboptFlags length differs from the template");
    aln::run_enumeration(ALN_ENUM_CW, *params, subopt->data(), dpm, as, user_limit);
    std::cerr << "Ali#=" << as.size() << std::endl;
  }
 private:
  const NOaliParams* params;
  const SuboptFlags* subopt;
};
#endif
