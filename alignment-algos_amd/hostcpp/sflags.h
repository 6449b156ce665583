// sflags.h — per-template-position "suboptimal region" flags (reference sflags.h:23-37, sflags.cpp:15-58).
// Note the constructor order (bool, length): aa_ali.cpp:86 passes them swapped.
#ifndef ALN_HOST_SFLAGS_H
#define ALN_HOST_SFLAGS_H
#include <string>
#include <vector>
#include "sequence.h"
// standard headers the reference's sflags.h hands on to its includers
#include <iostream>
#include <valarray>
using namespace std;   // as the reference's sflags.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class SuboptFlags : public Sequence<SequenceElem*> {
 public:
  SuboptFlags(bool f, size_t len) : bits_(len, f ? 1 : 0), next_(0) {
    seq_name = "Flags=suboptimal region";
    seq_string.assign(len, f ? '1' : '0');
  }
  bool operator[](unsigned int i) const { return bits_[i] != 0; }
  void append(const std::string& s) {
    for (size_t k = 0; k < s.size(); ++k) {
      if (next_ >= bits_.size()) throw std::string("Sequence flags longer than template!");
      Set((unsigned int)next_++, s[k] != '0');
    }
  }
  void append(const char* cs) { append(std::string(cs)); }
  void Set(unsigned int i, bool b) {
    if (i >= bits_.size()) throw std::string("Subopt index out of range");
    bits_[i] = b ? 1 : 0;
    seq_string[i] = b ? '1' : '0';
  }
  size_t size() const { return bits_.size(); }
  const unsigned char* data() const { return bits_.data(); }
 private:
  std::vector<unsigned char> bits_;
  size_t next_;
};
#endif
