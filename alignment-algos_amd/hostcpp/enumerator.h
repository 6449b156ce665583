// enumerator.h — abstract traceback / enumeration strategy (reference enumerator.h:20-25).
#ifndef ALN_HOST_ENUMERATOR_H
#define ALN_HOST_ENUMERATOR_H
template <class S1, class S2, class Etype> class DPMatrix;
template <class S1, class S2, class Etype> class AlignmentSet;

template <class S1, class S2, class Etype>
class Enumerator {
 public:
  virtual ~Enumerator() {}
  virtual int estimateSize() const = 0;
  virtual void enumerate(DPMatrix<S1, S2, Etype>& dpm, AlignmentSet<S1, S2, Etype>& as) = 0;
};
#endif
