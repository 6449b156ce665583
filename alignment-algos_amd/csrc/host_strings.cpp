// host_strings.cpp — host-only helpers of the C ABI: percent identity and the gapped alignment strings.
// No device work.  Semantics of AlignedPairList::calcIdentity (reference alignment.h:856-865) and
// SequenceGaps (gstrings.h:84-164, gstrings.cpp:17-29): the template line carries, after template
// position j, as many '-' as the longest query insertion any alignment of the set places there
// ("anchors"); each query line spells the query residues consumed between consecutive aligned pairs,
// lower-casing the extra residues of a zig-zag jump (both indices advance by more than one).
#include <algorithm>
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include "aln_hip.h"

namespace {

// longest insertion after each template position over the whole set
std::vector<int> insertion_widths(int T, const aln_alignment* alis, int n_alis, const int32_t* pairs) {
  std::vector<int> width(T > 1 ? T - 1 : 0, 0);
  for (int a = 0; a < n_alis; ++a) {
    const int32_t* p = pairs + 2 * alis[a].pair_off;
    for (int k = 1; k < alis[a].n_pairs; ++k) {
      int dq = p[2 * k] - p[2 * (k - 1)];
      int at = p[2 * (k - 1) + 1];
      if (dq != 1 && at >= 0 && at < T - 1) width[at] = std::max(width[at], dq - 1);
    }
  }
  return width;
}

}  // namespace

extern "C" {

float aln_identity(const char* qstr, int32_t Q, const char* tstr, int32_t T, const int32_t* pairs, int32_t n_pairs) {
  int same = -2;                                   // the head and tail pairs always match themselves
  int total = std::min(Q, T) - 2;
  for (int k = 0; k < n_pairs; ++k) {
    int q = pairs[2 * k], t = pairs[2 * k + 1];
    if (q >= 0 && q < Q && t >= 0 && t < T && qstr[q] == tstr[t]) ++same;
  }
  return float(same) / float(total) * 100.f;
}

int32_t aln_gapped_length(int32_t T, const aln_alignment* alis, int32_t n_alis, const int32_t* pairs) {
  std::vector<int> w = insertion_widths(T, alis, n_alis, pairs);
  int total = T;
  for (int v : w) total += v;
  return total;
}

int aln_gapped_strings(const char* qstr, int32_t Q, const char* tstr, int32_t T, const aln_alignment* alis,
                       int32_t n_alis, const int32_t* pairs, char* tline, char* qlines, int32_t stride) {
  if (!qstr || !tstr || Q < 2 || T < 2 || (n_alis > 0 && (!alis || !pairs))) return ALN_E_ARG;
  std::vector<int> w = insertion_widths(T, alis, n_alis, pairs);
  int total = T;
  for (int v : w) total += v;
  if (total >= stride) return ALN_E_OVERFLOW;
  if (tline) {
    std::string s;
    s.reserve(total);
    for (int j = 0; j < T - 1; ++j) { s.push_back(tstr[j]); s.append(w[j], '-'); }
    s.push_back(tstr[T - 1]);
    memcpy(tline, s.c_str(), s.size() + 1);
  }
  for (int a = 0; a < n_alis && qlines; ++a) {   // qlines == NULL: template line only
    const int32_t* p = pairs + 2 * alis[a].pair_off;
    const int np = alis[a].n_pairs;
    std::string s;
    s.reserve(total);
    int k = 0;                                    // next aligned pair to place
    for (int j = 0; j < T - 1; ++j) {
      int room = w[j] + 1;                        // columns available at template position j
      if (k < np && p[2 * k + 1] == j) {
        if (k + 1 >= np) return ALN_E_ARG;        // the list must end at the tail pair (T-1 is never visited)
        int x = p[2 * k], y = p[2 * (k + 1)];
        int dt = p[2 * (k + 1) + 1] - j, dq = y - x;
        if (x < 0 || y > Q || dq < 0) return ALN_E_ARG;
        size_t at = s.size();
        s.append(qstr + x, (size_t)dq);
        if (!(dt == 1 || dq == 1))                // zig-zag: residues after the aligned one are shown in lower case
          for (size_t c = at + 1; c < s.size(); ++c) s[c] = (char)tolower((unsigned char)s[c]);
        room -= dq;
        ++k;
      }
      if (room < 0) return ALN_E_ARG;
      s.append((size_t)room, '-');
    }
    int rest = total - (int)s.size();
    if (rest > 1) s.append((size_t)(rest - 1), '-');
    s.push_back(qstr[Q - 1]);
    if ((int)s.size() >= stride) return ALN_E_OVERFLOW;
    memcpy(qlines + (size_t)a * stride, s.c_str(), s.size() + 1);
  }
  return ALN_OK;
}

}  // extern "C"
