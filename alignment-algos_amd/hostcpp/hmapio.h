// hmapio.h — HMAP alignment writer with the reference's stream syntax: `out << Formats::HMAPOut(submatrix, 60) << set`
// (reference hmapio.h:19-164, hmapio.cpp:6-41).  Per alignment: ">query_k (sc=..,ev=..,id=..%)  UID=u", the two lengths, then
// blocks of five lines — template SSE string, "model:" residues, match marks ('|' identical, ':' positive substitution score
// when a matrix file is given, '.' positive similarity, ' ' otherwise), "query:" residues, query SSE string — each gapped for
// THAT alignment only (one-hot SequenceGaps mask), '^' / '$' stripped, wrapped at line_length.
// The reference header includes the Troll-dependent sequence types and cannot be built here: this writer restates the source
// and has no golden (parity unpinned); the gapped strings underneath are the pinned SequenceGaps helpers.
#ifndef ALN_HOST_HMAPIO_H
#define ALN_HOST_HMAPIO_H
#include <iostream>
#include <sstream>
#include <string>
#include <valarray>
#include "formats.h"
#include "gstrings.h"
#include "submatrix.h"

class HMAPWrite {
 public:
  HMAPWrite(std::ostream& o, const char* sm, int len) : output(&o), line_length(len), submatrix_fn(sm) {}
  template <class S1, class S2, class Etype>
  void write(AlignmentSet<S1, S2, Etype>& as) {
    int count = 0;
    std::string gapped_templ_sse, gapped_templ, gapped_marks, gapped_query, gapped_query_sse;
    std::valarray<bool> mask(false, as.size());
    for (typename AlignmentSet<S1, S2, Etype>::iterator it = as.begin(); it != as.end(); ++it) {
      mask[count] = true;
      SequenceGaps gaps(as, mask);
      *output << ">" << as.getQuerySequence()->seq_name << "_" << count;
      std::string annot;
      makeAnnotation(*it, annot);
      if (annot != "") *output << " " << annot;
      *output << std::endl << std::endl;
      *output << "model: length " << as.getTemplateSequence()->size() - 2 << std::endl;
      *output << "query: length " << as.getQuerySequence()->size() - 2 << std::endl;
      gaps.build(*as.getTemplateSequence()->getSSEString(), gapped_templ_sse, ' ');
      fix_ends(gapped_templ_sse);
      gaps.build(*as.getTemplateSequence()->getString(), gapped_templ);
      fix_ends(gapped_templ);
      generateMarks(*it, as, gapped_marks);
      {
        std::string src = gapped_marks;
        if (src.size() < as.getQuerySequence()->size()) src.resize(as.getQuerySequence()->size(), ' ');   // lists end at the tail pair; be safe
        gaps.build(src, *it, gapped_marks, ' ');
      }
      fix_ends(gapped_marks);
      gaps.build(*as.getQuerySequence()->getString(), *it, gapped_query);
      fix_ends(gapped_query);
      gaps.build(*as.getQuerySequence()->getSSEString(), *it, gapped_query_sse, ' ');
      fix_ends(gapped_query_sse);
      write(gapped_templ_sse, gapped_templ, gapped_marks, gapped_query, gapped_query_sse);
      *output << std::endl;
      mask[count++] = false;
    }
  }
  void write(const std::string& templ_sse, const std::string& templ, const std::string& marks, const std::string& query,
             const std::string& query_sse) {
    const int size = (int)templ.size();
    for (int i = 0; i < size; i += line_length) {
      *output << std::endl;
      *output << "       " << sub(templ_sse, i) << std::endl;
      *output << "model: " << sub(templ, i) << std::endl;
      *output << "       " << sub(marks, i) << std::endl;
      *output << "query: " << sub(query, i) << std::endl;
      *output << "       " << sub(query_sse, i) << std::endl;
    }
  }
  template <class S1, class S2>
  void makeAnnotation(AlignedPairList<S1, S2>& ali, std::string& s) {
    std::stringstream buff("");
    buff << "(sc=" << ali.score << ",ev=" << ali.significance << ",id=" << ali.identity << "%)" << "  UID=" << ali.uid;
    s.assign(buff.str());
  }
  template <class S1, class S2, class Etype>
  void generateMarks(AlignedPairList<S1, S2>& ali, AlignmentSet<S1, S2, Etype>& as, std::string& marks) {
    BlosumMatrix* bm = 0;
    if (submatrix_fn != "") bm = new BlosumMatrix(submatrix_fn.c_str());
    int qp = -1;
    std::stringstream buffer("");
    const std::string* q_seq = as.getQuerySequence()->getString();
    const std::string* t_seq = as.getTemplateSequence()->getString();
    for (typename AlignedPairList<S1, S2>::iterator it = ali.begin(); it != ali.end(); ++it) {
      const int qi = it->first, ti = it->second;
      const char qc = (*q_seq)[qi], tc = (*t_seq)[ti];
      const float s = as.getDPMatrix()->getSim(qi, ti);
      buffer << std::string(qi - qp - 1, ' ');
      qp = qi;
      if (qc == SequenceElem::Head || qc == SequenceElem::Tail) buffer << qc;
      else if (qc == tc) buffer << '|';
      else if (bm && bm->score(qc, tc) > 0) buffer << ':';
      else if (s > 0) buffer << '.';
      else buffer << ' ';
    }
    marks = buffer.str();
    delete bm;                                   // (the reference leaks it)
  }
  static void fix_ends(std::string& seq) {
    if (!seq.empty() && seq[0] == SequenceElem::Head) seq.erase(0, 1);
    if (!seq.empty() && seq[seq.size() - 1] == SequenceElem::Tail) seq.erase(seq.size() - 1);
  }
  std::ostream* output;
  int line_length;
  std::string submatrix_fn;

 private:
  std::string sub(const std::string& s, int i) const { return i < (int)s.size() ? s.substr(i, line_length) : std::string(); }
};

inline HMAPWrite operator<<(std::ostream& o, Formats::HMAPOut p) { return HMAPWrite(o, p.submatrix.c_str(), p.line_length); }
template <class S1, class S2, class Etype>
std::ostream& operator<<(HMAPWrite w, AlignmentSet<S1, S2, Etype>& as) { w.write(as); return *w.output; }
#endif
