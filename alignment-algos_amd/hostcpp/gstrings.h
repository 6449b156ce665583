// gstrings.h — SequenceGaps: gapped display strings for a set of alignments (reference gstrings.h:25-164,
// gstrings.cpp:15-29), computed by the engine's host helpers aln_gapped_length / aln_gapped_strings.
// The constructor records the (optionally masked) set, whose longest insertions define the gap columns; build()
// then renders the template line or one alignment's query line against those columns.
#ifndef ALN_HOST_GSTRINGS_H
#define ALN_HOST_GSTRINGS_H
#include <string>
#include <valarray>
#include <vector>
#include "alignment.h"
// standard headers the reference's gstrings.h hands on to its includers
#include <algorithm>
#include <cassert>
#include <cctype>
using namespace std;   // as the reference's gstrings.h does at header scope: sources written against it name string, vector, cerr ... unqualified

class SequenceGaps {
 public:
  template <class S1, class S2, class Etype>
  SequenceGaps(const AlignmentSet<S1, S2, Etype>& as, std::valarray<bool> mask = std::valarray<bool>(0))
      : query_len((int)as.getQuerySequence()->size()), template_len((int)as.getTemplateSequence()->size()) { collect(as, mask); }
  template <class S1, class S2, class Etype>
  SequenceGaps(const AlignmentSet<S1, S2, Etype>& as, int qs, int ts, std::valarray<bool> mask = std::valarray<bool>(0))
      : query_len(qs), template_len(ts) { collect(as, mask); }

  // the template line (gstrings.cpp:17-29)
  void build(const std::string& sequence, std::string& result, char gc = '-') {
    const std::string dummy_query(query_len, 'X');
    std::vector<aln_alignment> alis(alis_);
    std::vector<int32_t> pairs(pairs_);
    render(dummy_query, sequence, alis, pairs, &result, 0);
    regap(result, gc);
  }
  // one alignment's query line (gstrings.h:118-164)
  template <class S1, class S2>
  void build(const std::string& sequence, const AlignedPairList<S1, S2>& a, std::string& result, char gc = '-') {
    std::vector<aln_alignment> alis(alis_);
    std::vector<int32_t> pairs(pairs_);
    std::vector<int32_t> flat;
    a.flatten(flat);
    aln_alignment one = aln_alignment();
    one.n_pairs = (int32_t)a.size();
    one.pair_off = (int64_t)(pairs.size() / 2);
    pairs.insert(pairs.end(), flat.begin(), flat.end());
    alis.push_back(one);                      // a member of the recorded set adds nothing to its gap columns
    const std::string dummy_templ(template_len, 'X');
    render(sequence, dummy_templ, alis, pairs, 0, &result);
    regap(result, gc);
  }

 private:
  template <class S1, class S2, class Etype>
  void collect(const AlignmentSet<S1, S2, Etype>& as, const std::valarray<bool>& mask) {
    for (size_t k = 0; k < as.size(); ++k) {
      if (mask.size() != 0 && !mask[k]) continue;
      aln_alignment a = aln_alignment();
      a.n_pairs = (int32_t)as[k].size();
      a.pair_off = (int64_t)(pairs_.size() / 2);
      std::vector<int32_t> flat;
      as[k].flatten(flat);
      pairs_.insert(pairs_.end(), flat.begin(), flat.end());
      alis_.push_back(a);
    }
  }
  // renders every line of `alis`; hands back the template line or the LAST query line
  void render(const std::string& q, const std::string& t, std::vector<aln_alignment>& alis, std::vector<int32_t>& pairs,
              std::string* tline, std::string* last_qline) {
    if (pairs.empty()) pairs.push_back(0);
    const int n = (int)alis.size();
    const int len = aln_gapped_length(template_len, n ? alis.data() : 0, n, pairs.data());
    const int stride = len + 1;
    std::vector<char> tb(stride), qb(last_qline ? (size_t)stride * (n ? n : 1) : 1);
    int rc = aln_gapped_strings(q.c_str(), query_len, t.c_str(), template_len, n ? alis.data() : 0, n, pairs.data(), tb.data(),
                                last_qline ? qb.data() : 0, stride);
    if (rc != ALN_OK) throw std::string(aln_error_string(rc));
    if (tline) *tline = std::string(tb.data());
    if (last_qline) *last_qline = std::string(qb.data() + (size_t)stride * (n - 1));
  }
  static void regap(std::string& s, char gc) {
    if (gc == '-') return;
    for (size_t k = 0; k < s.size(); ++k) if (s[k] == '-') s[k] = gc;
  }
  int query_len, template_len;
  std::vector<aln_alignment> alis_;
  std::vector<int32_t> pairs_;
};
#endif
