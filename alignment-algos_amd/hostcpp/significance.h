// significance.h — score -> significance functor interface used by AlignedPairList::calcSignificance.
#ifndef ALN_HOST_SIGNIFICANCE_H
#define ALN_HOST_SIGNIFICANCE_H
template <class Stype>
class Significance {
 public:
  float significance(float score) const { return static_cast<const Stype&>(*this).significance(score); }
};
#endif
