// hmapio.h — HMAP alignment report: `out << Formats::HMAPOut(submatrix, 60) << set` (stream syntax and bytes of the
// reference's writer, hmapio.h:19-164 / hmapio.cpp:6-41; own construction).
//
// What a record looks like:  ">query_k (sc=..,ev=..,id=..%)  UID=u", blank, "model: length N", "query: length M", then the
// display cut into line_length-wide slices of five rows each — template SSE, "model:" residues, marks, "query:" residues,
// query SSE.  Each alignment is displayed on its own gap columns ('^' / '$' trimmed).
//
// How it is built here: the reference renders each row separately through a one-hot masked SequenceGaps; on a one-alignment
// mask the gap columns are the alignment's own insert lengths, so this writer walks the flattened pair list ONCE and appends
// to all five rows column group by column group:
//     aligned pair (q,t) followed by (q',t'):  template residue + (q'-q-1) gap columns over query residues q..q'-1 (the
//                                              residues after the first lower-cased when both indices jump: a zig-zag),
//     template positions t+1..t'-1:            one column each, query side gapped.
// Marks: '|' identical, ':' positive substitution score (when a matrix file was named), '.' positive similarity, ' ' otherwise.
// The reference writer drags in the Troll-dependent sequence types and cannot be compiled here: no golden, parity unpinned;
// tests check the rows against SequenceGaps renderings (pinned) and the format's invariants.
#ifndef ALN_HOST_HMAPIO_H
#define ALN_HOST_HMAPIO_H
#include <cctype>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include "formats.h"
#include "gstrings.h"
#include "submatrix.h"

namespace aln_hmapio {

// the five display rows of one alignment
struct Display {
  std::string t_sse, t_res, marks, q_res, q_sse;
  void trim_markers() {
    std::string* rows[5] = {&t_sse, &t_res, &marks, &q_res, &q_sse};
    for (int r = 0; r < 5; ++r) {
      std::string& s = *rows[r];
      if (!s.empty() && s[0] == SequenceElem::Head) s.erase(0, 1);
      if (!s.empty() && s[s.size() - 1] == SequenceElem::Tail) s.erase(s.size() - 1);
    }
  }
};

// what the walk needs to know about the two sequences
struct Strands {
  const std::string *q_res, *q_sse, *t_res, *t_sse;
};

// query-side piece q0..q1-1 of a row; `fold` lower-cases everything after the first character
inline void put_query_piece(std::string& row, const std::string& src, int q0, int q1, bool fold) {
  for (int q = q0; q < q1; ++q) {
    char c = q < (int)src.size() ? src[q] : ' ';
    row.push_back(fold && q > q0 ? (char)tolower((unsigned char)c) : c);
  }
}

// One walk over `pairs` (n × (q,t), ascending, ending at the tail pair): fills all rows.  `mark[k]` is the mark of pair k.
inline void lay_out(const Strands& s, const std::vector<int32_t>& pairs, const std::string& mark, Display& d) {
  const int n = (int)(pairs.size() / 2);
  const int T = (int)s.t_res->size(), Q = (int)s.q_res->size();
  int t_next = 0;                                            // next template position without a column yet
  for (int k = 0; k + 1 < n; ++k) {
    const int q0 = pairs[2 * k], t0 = pairs[2 * k + 1], q1 = pairs[2 * k + 2], t1 = pairs[2 * k + 3];
    for (; t_next < t0 && t_next < T - 1; ++t_next) {        // template residues no pair touches
      d.t_sse.push_back((*s.t_sse)[t_next]); d.t_res.push_back((*s.t_res)[t_next]);
      d.marks.push_back(' '); d.q_res.push_back('-'); d.q_sse.push_back(' ');
    }
    if (t0 != t_next || t0 >= T - 1) continue;
    const int span = q1 - q0;                                // columns of this group (>= 1 on monotone lists)
    const bool zigzag = (t1 - t0 != 1) && (span != 1);
    d.t_sse.push_back((*s.t_sse)[t0]); d.t_res.push_back((*s.t_res)[t0]); d.marks.push_back(mark[k]);
    if (span > 1) { d.t_sse.append(span - 1, ' '); d.t_res.append(span - 1, '-'); d.marks.append(span - 1, ' '); }
    put_query_piece(d.q_res, *s.q_res, q0, q1, zigzag);
    put_query_piece(d.q_sse, *s.q_sse, q0, q1, zigzag);
    ++t_next;
  }
  for (; t_next < T - 1; ++t_next) {
    d.t_sse.push_back((*s.t_sse)[t_next]); d.t_res.push_back((*s.t_res)[t_next]);
    d.marks.push_back(' '); d.q_res.push_back('-'); d.q_sse.push_back(' ');
  }
  // the closing column always shows the last character of every row's source
  d.t_sse.push_back((*s.t_sse)[T - 1]); d.t_res.push_back((*s.t_res)[T - 1]);
  d.q_res.push_back((*s.q_res)[Q - 1]); d.q_sse.push_back((*s.q_sse)[s.q_sse->size() - 1]);
  d.marks.push_back(n > 0 && pairs[2 * (n - 1)] == Q - 1 ? mark[n - 1] : ' ');
}

}  // namespace aln_hmapio

class HMAPWrite {
 public:
  HMAPWrite(std::ostream& o, int len, const std::string& sfn) : output(&o), line_length(len), submatrix_fn(sfn) {}

  template <class S1, class S2, class Etype>
  void write(AlignmentSet<S1, S2, Etype>& as) {
    aln_hmapio::Strands s = {as.getQuerySequence()->getString(), as.getQuerySequence()->getSSEString(),
                             as.getTemplateSequence()->getString(), as.getTemplateSequence()->getSSEString()};
    std::vector<int32_t> flat;
    for (size_t k = 0; k < as.size(); ++k) {
      std::string note;
      makeAnnotation(as[k], note);
      *output << ">" << as.getQuerySequence()->seq_name << "_" << k;
      if (!note.empty()) *output << " " << note;
      *output << std::endl << std::endl
              << "model: length " << as.getTemplateSequence()->size() - 2 << std::endl
              << "query: length " << as.getQuerySequence()->size() - 2 << std::endl;
      as[k].flatten(flat);
      aln_hmapio::Display d;
      aln_hmapio::lay_out(s, flat, pair_marks(flat, as), d);
      d.trim_markers();
      write(d.t_sse, d.t_res, d.marks, d.q_res, d.q_sse);
      *output << std::endl;
    }
  }

  // the display in line_length-wide slices; the model row decides how many
  void write(const std::string& templ_sse, const std::string& templ, const std::string& marks, const std::string& query,
             const std::string& query_sse) {
    const std::string* rows[5] = {&templ_sse, &templ, &marks, &query, &query_sse};
    static const char* const label[5] = {"       ", "model: ", "       ", "query: ", "       "};
    for (size_t at = 0; at < templ.size(); at += (size_t)line_length) {
      *output << std::endl;
      for (int r = 0; r < 5; ++r) {
        *output << label[r];
        if (at < rows[r]->size()) output->write(rows[r]->data() + at, (std::streamsize)std::min((size_t)line_length, rows[r]->size() - at));
        *output << std::endl;
      }
    }
  }

  void fix_ends(std::string& s) {
    if (!s.empty() && s[0] == SequenceElem::Head) s.erase(0, 1);
    if (!s.empty() && s[s.size() - 1] == SequenceElem::Tail) s.erase(s.size() - 1);
  }

  template <class S1, class S2>
  void makeAnnotation(AlignedPairList<S1, S2>& ali, std::string& s) {
    std::ostringstream os;
    os << "(sc=" << ali.score << ",ev=" << ali.significance << ",id=" << ali.identity << "%)  UID=" << ali.uid;
    s = os.str();
  }

  // the marks as a string over query positions (blank where the query residue is not aligned)
  template <class S1, class S2, class Etype>
  void generateMarks(AlignedPairList<S1, S2>& ali, AlignmentSet<S1, S2, Etype>& as, std::string& s) {
    std::vector<int32_t> flat;
    ali.flatten(flat);
    const std::string m = pair_marks(flat, as);
    s.assign(flat.empty() ? 0 : (size_t)flat[flat.size() - 2] + 1, ' ');
    for (size_t k = 0; k < m.size(); ++k) s[flat[2 * k]] = m[k];
  }

  std::ostream* output;
  int line_length;

 private:
  // one mark per aligned pair
  template <class S1, class S2, class Etype>
  std::string pair_marks(const std::vector<int32_t>& flat, AlignmentSet<S1, S2, Etype>& as) {
    if (!submatrix_fn.empty() && !table) table.reset(new BlosumMatrix(submatrix_fn.c_str()));
    const std::string& qs = *as.getQuerySequence()->getString();
    const std::string& ts = *as.getTemplateSequence()->getString();
    std::string m(flat.size() / 2, ' ');
    for (size_t k = 0; k < m.size(); ++k) {
      const int qi = flat[2 * k], ti = flat[2 * k + 1];
      const char a = qs[qi], b = ts[ti];
      if (a == SequenceElem::Head || a == SequenceElem::Tail) m[k] = a;
      else if (a == b) m[k] = '|';
      else if (table && table->score(a, b) > 0) m[k] = ':';
      else if (as.getDPMatrix()->getSim(qi, ti) > 0) m[k] = '.';
    }
    return m;
  }
  std::string submatrix_fn;
  std::shared_ptr<BlosumMatrix> table;          // read once per writer
};

inline HMAPWrite operator<<(std::ostream& o, Formats::HMAPOut p) { return HMAPWrite(o, p.line_length, p.submatrix); }
template <class S1, class S2, class Etype>
std::ostream& operator<<(HMAPWrite w, AlignmentSet<S1, S2, Etype>& as) { w.write(as); return *w.output; }
#endif
