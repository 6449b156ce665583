// enumerator.h — the strategy interface every traceback / enumeration class implements (reference enumerator.h:20-25):
// Optimal*, ConstrainedNearOptimal, UnconstrainedNearOptimal, KSConstrainedNearOptimal.  An AlignmentSet is built by handing
// it a finished DPMatrix and one of these.
#ifndef ALN_HOST_ENUMERATOR_H
#define ALN_HOST_ENUMERATOR_H

template <class QuerySeq, class TemplSeq, class Eval> class AlignmentSet;
template <class QuerySeq, class TemplSeq, class Eval> class DPMatrix;

template <class QuerySeq, class TemplSeq, class Eval>
struct Enumerator {
  typedef DPMatrix<QuerySeq, TemplSeq, Eval> Matrix;
  typedef AlignmentSet<QuerySeq, TemplSeq, Eval> Set;
  virtual ~Enumerator() {}
  // read alignments out of `matrix` and append them to `out` (what is already there stays and takes part in sortSet)
  virtual void enumerate(Matrix& matrix, Set& out) = 0;
  // how many alignments enumerate() expects to add (a reserve() hint)
  virtual int estimateSize() const = 0;
};
#endif
