// alignment.h — alignment containers with the reference's names and behaviour (alignment.h:34-113, :847-949):
//   AlignedPair      (query_idx, template_idx)
//   AlignedPairList  list of pairs + score / identity / significance / uid; operator< = "higher score first"
//   AlignmentSet     vector of lists bound to a DPMatrix and an Enumerator; sortSet(max), assignIdentity()
// Identity is computed by the engine's host helper aln_identity (calcIdentity, reference :856-865).
#ifndef ALN_HOST_ALIGNMENT_H
#define ALN_HOST_ALIGNMENT_H
#include <algorithm>
#include <list>
#include <string>
#include <utility>
#include <vector>

#include "aln_hip.h"
#include "dpmatrix.h"
#include "enumerator.h"
#include "sequence.h"
#include "sflags.h"
#include "significance.h"
// standard headers the reference's alignment.h hands on to its includers
#include <cassert>
#include <iostream>
using namespace std;   // as the reference's alignment.h does at header scope: sources written against it name string, vector, cerr ... unqualified

template <class S1, class S2>
class AlignedPair : public std::pair<int, int> {
 public:
  AlignedPair() : std::pair<int, int>(-1, -1) {}
  AlignedPair(int i, int j) : std::pair<int, int>(i, j) {}
  int query_idx() const { return first; }
  int template_idx() const { return second; }
};

template <class S1, class S2>
class AlignedPairList : public std::list<AlignedPair<S1, S2> > {
  typedef AlignedPair<S1, S2> Pair;
 public:
  AlignedPairList() : score(0.f), identity(0.f), significance(9999.f), SSE_CO(0.f), coverage(0.f), uid(-1) {}
  AlignedPairList(const Pair& a, float s) : score(s), identity(0.f), significance(9999.f), SSE_CO(0.f), coverage(0.f), uid(-1) { this->push_back(a); }
  // two gapped sequence objects of equal length (reference alignment.h:147-171; note the order it appends: (index in s2, index in s1))
  AlignedPairList(const S1& s1, const S2& s2) : score(0.f), identity(0.f), significance(9999.f), SSE_CO(0.f), coverage(0.f), uid(-1) {
    int at1 = -1, at2 = -1;
    for (unsigned int c = 0; c < s1.size(); ++c) {
      const bool r1 = s1.olc(c) != '-', r2 = s2.olc(c) != '-';
      at1 += r1; at2 += r2;
      if (r1 && r2) append(at2, at1);
    }
  }
  // rebuilds the list from two gapped display lines (reference alignment.h:116-145): a column with a residue in both lines is
  // an aligned pair; identity = identical / aligned columns (markers excluded) in percent; throws a C string on unequal lengths
  void readFrom(const std::string& query, const std::string& templ) {
    score = 0.f; significance = 9999.f; uid = -1;
    this->clear();
    if (query.size() != templ.size()) throw ("readFrom error: query and templ not equal length");
    int qi = -1, ti = -1;
    float same = 0.f, aligned = 0.f;
    for (size_t c = 0; c < query.size(); ++c) {
      const char a = query[c], b = templ[c];
      if (a != '-') ++qi;
      if (b != '-') ++ti;
      if (a == '-' || b == '-') continue;
      append(qi, ti);
      if (a == '^' || a == '$' || b == '^' || b == '$') continue;
      aligned += 1.f;
      if (a == b) same += 1.f;
    }
    identity = same;
    identity /= aligned;
    identity *= 100.f;
  }
  void append(int i, int j) { this->push_back(Pair(i, j)); }
  void prepend(int i, int j) { this->push_front(Pair(i, j)); }
  void calcIdentity(const std::string& query, const std::string& templ) {
    std::vector<int32_t> flat;
    flatten(flat);
    identity = aln_identity(query.c_str(), (int32_t)query.size(), templ.c_str(), (int32_t)templ.size(), flat.data(), (int32_t)this->size());
  }
  template <class Stype>
  void calcSignificance(const Significance<Stype>& s) { significance = s.significance(score); }
  bool operator<(const AlignedPairList& a) const { return score > a.score; }
  void flatten(std::vector<int32_t>& out) const {
    out.clear();
    out.reserve(2 * this->size());
    for (typename std::list<Pair>::const_iterator it = this->begin(); it != this->end(); ++it) { out.push_back(it->first); out.push_back(it->second); }
  }
  void print_pairs() const {
    for (typename std::list<Pair>::const_iterator it = this->begin(); it != this->end(); ++it) std::cerr << "(" << it->first << "," << it->second << ") ";
    std::cerr << std::endl;
  }
  float score;
  float identity;
  float significance;
  float SSE_CO;
  float coverage;
  int uid;
};

template <class S1, class S2, class Etype>
class AlignmentSet : public std::vector<AlignedPairList<S1, S2> > {
  typedef AlignedPairList<S1, S2> Alignment;
 public:
  // a set that holds one given alignment and is bound to no matrix (reference alignment.h:882)
  AlignmentSet(const Alignment& apl) : std::vector<Alignment>(1, apl), dpmatrix(0), enumerator(0) {}
  // (the reference's copy constructor, :908-911, does not compile when instantiated — SURVEY B9; this one copies the bindings)
  AlignmentSet(const AlignmentSet& as) : std::vector<Alignment>(as), dpmatrix(as.dpmatrix), enumerator(as.enumerator) {}
  AlignmentSet(DPMatrix<S1, S2, Etype>& dpm, Enumerator<S1, S2, Etype>& en) : dpmatrix(&dpm), enumerator(&en) { build(); }
  const S1* getQuerySequence() const { return dpmatrix->getQuerySequence(); }
  const S2* getTemplateSequence() const { return dpmatrix->getTemplateSequence(); }
  const DPMatrix<S1, S2, Etype>* getDPMatrix() const { return dpmatrix; }
  // std::sort when everything is kept, std::partial_sort + erase otherwise — the very calls of the reference
  // (alignment.h:922-932): with an unstable sort the order of equal scores is part of the observable result
  void sortSet(int max) {
    if (max >= (int)this->size()) std::sort(this->begin(), this->end());
    else if (max > 0) {
      std::partial_sort(this->begin(), this->begin() + max, this->end());
      this->erase(this->begin() + max, this->end());
    }
  }
  void assignIdentity() {
    const std::string& q = *dpmatrix->getQuerySequence()->getString();
    const std::string& t = *dpmatrix->getTemplateSequence()->getString();
    for (size_t k = 0; k < this->size(); ++k) (*this)[k].calcIdentity(q, t);
  }
  template <class Stype>
  void assignSignificance(const Significance<Stype>& s) { for (size_t k = 0; k < this->size(); ++k) (*this)[k].calcSignificance(s); }
 private:
  void build() {
    this->reserve(enumerator->estimateSize());
    enumerator->enumerate(*dpmatrix, *this);
    assignIdentity();
  }
  DPMatrix<S1, S2, Etype>* dpmatrix;
  Enumerator<S1, S2, Etype>* enumerator;
};

struct IdentityComparator {
  template <class S1, class S2>
  bool operator()(const AlignedPairList<S1, S2>& a, const AlignedPairList<S1, S2>& b) { return a.identity > b.identity; }
};
struct ScoreComparator {
  template <class S1, class S2>
  bool operator()(const AlignedPairList<S1, S2>& a, const AlignedPairList<S1, S2>& b) { return a.score > b.score; }
};
#endif
