// aa_seq.h — plain amino-acid sequence (reference aa_seq.h:10-25, aa_seq.cpp:5-35).  Own implementation.
#ifndef ALN_HOST_AA_SEQ_H
#define ALN_HOST_AA_SEQ_H
#include <string>
#include "sequence.h"
// standard headers the reference's aa_seq.h hands on to its includers
#include <vector>

class AASequence : public Sequence<SequenceElem*> {
 public:
  AASequence() {}
  ~AASequence() { for (size_t i = 0; i < size(); ++i) delete (*this)[i]; }
  void append(const std::string& s) {
    for (size_t k = 0; k < s.size(); ++k) push_back(new SequenceElem((int)size(), s[k]));
    seq_string.clear();
  }
  void append(const char* cs) { append(std::string(cs)); }
  void cleargaps(char c) {
    size_t w = 0;
    for (size_t r = 0; r < size(); ++r) {
      if ((*this)[r]->olc == c) delete (*this)[r];
      else (*this)[w++] = (*this)[r];
    }
    resize(w);
    seq_string.clear();
  }
 private:
  AASequence(const AASequence&);
  AASequence& operator=(const AASequence&);
};
#endif
