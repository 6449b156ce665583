// sequence.h — sequences as vectors of element pointers with '^' head and '$' tail sentinels.
// Same surface as the reference (sequence.h:22-64, sequence.cpp:15-36): SequenceElem{index,olc,isHead,isTail},
// Sequence<elem_t>{seq_length, seq_name, olc(i), getString()}.  Own implementation.
#ifndef ALN_HOST_SEQUENCE_H
#define ALN_HOST_SEQUENCE_H
#include <string>
#include <vector>
// standard headers the reference's sequence.h hands on to its includers
#include <sstream>
using namespace std;   // as the reference's sequence.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class SequenceElem {
 public:
  int index;
  char olc;
  SequenceElem() : index(-1), olc(' ') {}
  SequenceElem(int i, char o) : index(i), olc(o) {}
  bool isHead() const { return olc == Head; }
  bool isTail() const { return olc == Tail; }
  static const char Head = '^';
  static const char Tail = '$';
};

template <class elem_t>
class Sequence : public std::vector<elem_t> {
 public:
  Sequence() : seq_length(0) {}
  unsigned int seq_length;      // length without head/tail
  std::string seq_name;
  char olc(int i) const { return std::vector<elem_t>::at(i)->olc; }
  // one-letter string including the sentinels, built lazily
  const std::string* getString() const {
    if (seq_string.empty()) {
      seq_string.reserve(this->size());
      for (typename std::vector<elem_t>::const_iterator it = this->begin(); it != this->end(); ++it) seq_string.push_back((*it)->olc);
    }
    return &seq_string;
  }
 protected:
  mutable std::string seq_string;
};
#endif
