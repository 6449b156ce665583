// significance.h — static-polymorphic "score to significance" functor that AlignedPairList::calcSignificance accepts
// (reference significance.h).  A model derives from Significance<Model> and provides significance(float).
#ifndef ALN_HOST_SIGNIFICANCE_H
#define ALN_HOST_SIGNIFICANCE_H

template <class Model>
struct Significance {
  const Model& model() const { return *static_cast<const Model*>(this); }
  float significance(float raw_score) const { return model().significance(raw_score); }
};
#endif
