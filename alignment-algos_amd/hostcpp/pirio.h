// pirio.h — PIR alignment writer with the reference's stream syntax: `out << Formats::PIROut(60) << set`
// (reference pirio.h:17-71, pirio.cpp:6-33).  One "#start ... #end" block per alignment: the template line gapped for
// THAT alignment only (SequenceGaps with a one-hot mask, pirio.h:39-41), ">P1;name", "structureN:name::::" /
// "sequence:name::::", '^' and '$' stripped from the ends (fix_ends), lines wrapped at line_length.
#ifndef ALN_HOST_PIRIO_H
#define ALN_HOST_PIRIO_H
#include <iostream>
#include <string>
#include <vector>
#include "formats.h"
#include "sequence.h"

class PIRWrite {
 public:
  PIRWrite(std::ostream& o, int len) : output(&o), line_length(len) {}
  void write(const std::string& seq) {
    for (size_t i = 0; i < seq.size(); i += (size_t)line_length) *output << seq.substr(i, line_length) << std::endl;
  }
  static void fix_ends(std::string& seq) {
    if (!seq.empty() && seq[0] == '^') seq.erase(0, 1);
    if (!seq.empty() && seq[seq.size() - 1] == '$') seq.erase(seq.size() - 1);
  }
  template <class S1, class S2, class Etype>
  void write(AlignmentSet<S1, S2, Etype>& as) {
    const std::string& q = *as.getQuerySequence()->getString();
    const std::string& t = *as.getTemplateSequence()->getString();
    for (size_t k = 0; k < as.size(); ++k) {
      std::vector<int32_t> flat;
      as[k].flatten(flat);
      aln_alignment one = aln_alignment();
      one.n_pairs = (int32_t)as[k].size();
      one.pair_off = 0;
      if (flat.empty()) flat.push_back(0);
      const int len = aln_gapped_length((int32_t)t.size(), &one, 1, flat.data());
      std::vector<char> tb(len + 1), qb(len + 1);
      int rc = aln_gapped_strings(q.c_str(), (int32_t)q.size(), t.c_str(), (int32_t)t.size(), &one, 1, flat.data(), tb.data(), qb.data(),
                                  len + 1);
      if (rc != ALN_OK) throw std::string(aln_error_string(rc));
      std::string ts(tb.data()), qs(qb.data());
      *output << "#start" << std::endl << std::endl;
      *output << ">P1;" << as.getTemplateSequence()->seq_name << std::endl;
      *output << "structureN:" << as.getTemplateSequence()->seq_name << "::::" << std::endl;
      fix_ends(ts);
      write(ts);
      *output << std::endl;
      *output << ">P1;" << as.getQuerySequence()->seq_name << std::endl;
      *output << "sequence:" << as.getQuerySequence()->seq_name << "::::" << std::endl;
      fix_ends(qs);
      write(qs);
      *output << std::endl;
      *output << "#end" << std::endl;
    }
  }
  std::ostream* output;
  int line_length;
};

inline PIRWrite operator<<(std::ostream& o, Formats::PIROut p) { return PIRWrite(o, p.line_length); }
template <class S1, class S2, class Etype>
std::ostream& operator<<(PIRWrite w, AlignmentSet<S1, S2, Etype>& as) { w.write(as); return *w.output; }
#endif
