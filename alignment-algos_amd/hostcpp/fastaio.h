// fastaio.h — FASTA reader / alignment writer with the reference's stream syntax (fastaio.h:12-205, fastaio.cpp:3-30).
// Writer: "> template_name" + gapped template line(s), then per alignment "> query_name_k (sc=..,ev=..,id=..%)" and its
// gapped query line(s), wrapped at line_length.  Reader: first record whose header contains `find_me`; '^' and '$'
// are added when head_tail.  Unlike the reference the reader does not duplicate the last line of a file that ends
// with a newline.
#ifndef ALN_HOST_FASTAIO_H
#define ALN_HOST_FASTAIO_H
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "formats.h"
#include "sequence.h"

class FastaWrite {
 public:
  FastaWrite(std::ostream& o, int len) : output(&o), line_length(len) {}
  void write(const std::string& seq) {
    for (size_t i = 0; i < seq.size(); i += (size_t)line_length) *output << seq.substr(i, line_length) << std::endl;
  }
  template <class elem_t>
  void write(Sequence<elem_t>& seq) {
    *output << "> " << seq.seq_name << std::endl;
    write(*seq.getString());
  }
  template <class S1, class S2>
  void makeAnnotation(AlignedPairList<S1, S2>& ali, std::string& s) {
    std::stringstream buff("");
    buff << "(sc=" << ali.score << ",ev=" << ali.significance << ",id=" << ali.identity << "%)";
    s = buff.str();
  }
  template <class S1, class S2, class Etype>
  void write(AlignmentSet<S1, S2, Etype>& as) {
    // every line of the set in one call of the engine's SequenceGaps helper
    const std::string& q = *as.getQuerySequence()->getString();
    const std::string& t = *as.getTemplateSequence()->getString();
    std::vector<aln_alignment> alis(as.size());
    std::vector<int32_t> pairs;
    for (size_t k = 0; k < as.size(); ++k) {
      std::vector<int32_t> flat;
      as[k].flatten(flat);
      alis[k] = aln_alignment();
      alis[k].n_pairs = (int32_t)as[k].size();
      alis[k].pair_off = (int64_t)(pairs.size() / 2);
      pairs.insert(pairs.end(), flat.begin(), flat.end());
    }
    if (pairs.empty()) pairs.push_back(0);
    const int n = (int)alis.size();
    const int len = aln_gapped_length((int32_t)t.size(), n ? alis.data() : 0, n, pairs.data());
    std::vector<char> tb(len + 1), qb((size_t)(len + 1) * (n ? n : 1));
    int rc = aln_gapped_strings(q.c_str(), (int32_t)q.size(), t.c_str(), (int32_t)t.size(), n ? alis.data() : 0, n, pairs.data(),
                                tb.data(), qb.data(), len + 1);
    if (rc != ALN_OK) throw std::string(aln_error_string(rc));
    *output << "> " << as.getTemplateSequence()->seq_name << std::endl;
    write(std::string(tb.data()));
    for (int k = 0; k < n; ++k) {
      *output << "> " << as.getQuerySequence()->seq_name << "_" << k;
      std::string annot;
      makeAnnotation(as[k], annot);
      if (annot != "") *output << " " << annot;
      *output << std::endl;
      write(std::string(qb.data() + (size_t)(len + 1) * k));
    }
  }
  std::ostream* output;
  int line_length;
};

inline FastaWrite operator<<(std::ostream& o, Formats::FastaOut p) { return FastaWrite(o, p.line_length); }
template <class S1, class S2, class Etype>
std::ostream& operator<<(FastaWrite w, AlignmentSet<S1, S2, Etype>& as) { w.write(as); return *w.output; }
template <class elem_t>
std::ostream& operator<<(FastaWrite w, Sequence<elem_t>& seq) { w.write(seq); return *w.output; }

class FastaRead {
 public:
  FastaRead(std::istream& i, const std::string& s = "", bool flag = true) : input(&i), head_tail(flag), find_me(s) {}
  template <class S>
  void readInto(S& s) {
    std::string line, name;
    bool found = false;
    while (!found && std::getline(*input, line)) {
      if (!line.empty() && line[0] == '>') {
        std::string h = line.substr(1);
        if (!h.empty() && h[0] == ' ') h.erase(0, h.find_first_not_of(' '));
        if (find_me.empty() || h.find(find_me) != std::string::npos) { name = h; found = true; }
      }
    }
    if (!found) {
      if (find_me.empty()) throw std::string("Error reading fasta file");
      throw std::string("Could not find search string: ") + find_me;
    }
    s.seq_name = name;
    if (head_tail) s.append("^");
    while (input->good()) {
      int c = input->peek();
      if (c == '>' || c == EOF) break;
      std::getline(*input, line);
      if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
      s.append(line);
    }
    if (head_tail) s.append("$");
  }
  std::istream* input;
  bool head_tail;
  std::string find_me;
};

inline FastaRead operator>>(std::istream& i, Formats::FastaIn p) { return FastaRead(i, p.find_me, p.head_tail); }
template <class S>
std::istream& operator>>(FastaRead r, S& s) { r.readInto(s); return *r.input; }
#endif
