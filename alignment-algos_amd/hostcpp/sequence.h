// sequence.h — sequences as vectors of element pointers framed by a '^' head and a '$' tail marker.
// The surface the reference's drivers and evaluators use (sequence.h:22-64, sequence.cpp:15-36): SequenceElem {index, olc,
// isHead, isTail, Head, Tail}, Sequence<elem_t> {seq_length, seq_name, olc(i), getString()}.  Own implementation; the engine
// works on the marker-framed one-letter string (aln_seqs takes exactly what getString() returns).
#ifndef ALN_HOST_SEQUENCE_H
#define ALN_HOST_SEQUENCE_H
#include <sstream>      // (the reference's header hands <sstream>, <string>, <vector> and the std names on to its includers)
#include <string>
#include <vector>
using namespace std;

struct SequenceElem {
  static const char Head = '^';
  static const char Tail = '$';
  SequenceElem(int position = -1, char letter = ' ') : index(position), olc(letter) {}
  bool isHead() const { return olc == Head; }
  bool isTail() const { return olc == Tail; }
  bool isMarker() const { return isHead() || isTail(); }
  int index;       // position in the framed sequence
  char olc;        // one-letter code
};

template <class elem_t>
class Sequence : public std::vector<elem_t> {
  typedef std::vector<elem_t> Elems;

 public:
  Sequence() : seq_length(0) {}
  char olc(int i) const { return Elems::at(i)->olc; }
  // the framed one-letter string; rendered on first use (a derived class that edits its elements clears seq_string)
  const std::string* getString() const {
    if (seq_string.empty() && !this->empty()) {
      std::string letters(this->size(), ' ');
      for (size_t k = 0; k < letters.size(); ++k) letters[k] = (*this)[k]->olc;
      seq_string.swap(letters);
    }
    return &seq_string;
  }
  std::string seq_name;
  unsigned int seq_length;      // residues, markers not counted

 protected:
  mutable std::string seq_string;
};
#endif
