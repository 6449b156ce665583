// host_unit_test.cpp — host-only checks of the mirrored headers (no GPU call is made; links libalnhip.so for the host string
// helpers).  Run by tests/test_host_headers.py in the CPU suite.
//   layout     hmapio.h's one-walk five-row layout == the rows SequenceGaps renders one by one through a one-hot mask
//              (the route the reference's HMAPWrite takes, hmapio.h:48-92) on random monotone pair lists with zig-zag jumps
//   readfrom   AlignedPairList::readFrom (alignment.h:116-145) inverts the gapped display lines; AlignmentSet(const Alignment&)
//   names      sources written against the reference's headers use string / vector / cerr unqualified: they must compile here
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "aa_seq.h"
#include "aasubalib.h"
#include "alignment.h"
#include "cw.h"
#include "formats.h"
#include "gstrings.h"
#include "hmapio.h"
#include "matrix.h"
#include "pstore.h"

typedef AASubstitutionEval<AASequence, AASequence> Eval;
typedef AlignedPairList<AASequence, AASequence> List;
typedef AlignmentSet<AASequence, AASequence, Eval> Set;

// `string` and `vector` unqualified, as in the reference's drivers (aa_ali.cpp:37)
static string random_residues(std::mt19937& g, int n, const char* alphabet) {
  string s = "^";
  vector<char> pool(alphabet, alphabet + strlen(alphabet));
  for (int k = 0; k < n; ++k) s.push_back(pool[g() % pool.size()]);
  s.push_back('$');
  return s;
}

// a monotone list from (0,0) to (Q-1,T-1); steps: diagonal, template jump, query jump, or both (zig-zag)
static List random_list(std::mt19937& g, int Q, int T) {
  List l;
  int q = 0, t = 0;
  l.append(0, 0);
  while (q < Q - 2 && t < T - 2) {
    int dq = 1, dt = 1;
    switch (g() % 8) {
      case 0: dq = 1 + (int)(g() % 4); break;
      case 1: dt = 1 + (int)(g() % 4); break;
      case 2: dq = 2 + (int)(g() % 3); dt = 2 + (int)(g() % 3); break;
      default: break;
    }
    q = std::min(q + dq, Q - 2); t = std::min(t + dt, T - 2);
    l.append(q, t);
  }
  l.append(Q - 1, T - 1);
  return l;
}

static int check_layout(unsigned seed, int rounds) {
  std::mt19937 g(seed);
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    const int Q = 3 + (int)(g() % 60), T = 3 + (int)(g() % 60);
    const string q_res = random_residues(g, Q - 2, "ACDEFGHIKLMNPQRSTVWY"), t_res = random_residues(g, T - 2, "ACDEFGHIKLMNPQRSTVWY");
    const string q_sse = random_residues(g, Q - 2, "HEC"), t_sse = random_residues(g, T - 2, "HEC");
    List l = random_list(g, Q, T);
    vector<int32_t> flat;
    l.flatten(flat);
    string mark(flat.size() / 2, ' ');
    for (size_t k = 0; k < mark.size(); ++k) mark[k] = " |:."[g() % 4];
    mark[0] = '^'; mark[mark.size() - 1] = '$';
    // the reference's route: marks as a string over query positions, then five separate renderings
    string marks_by_q(Q, ' ');
    for (size_t k = 0; k < mark.size(); ++k) marks_by_q[flat[2 * k]] = mark[k];
    Set as(l);
    std::valarray<bool> one(true, 1);
    SequenceGaps gaps(as, Q, T, one);
    string w_tsse, w_tres, w_marks, w_qres, w_qsse;
    gaps.build(t_sse, w_tsse, ' ');
    gaps.build(t_res, w_tres);
    gaps.build(marks_by_q, l, w_marks, ' ');
    gaps.build(q_res, l, w_qres);
    gaps.build(q_sse, l, w_qsse, ' ');
    aln_hmapio::Strands s = {&q_res, &q_sse, &t_res, &t_sse};
    aln_hmapio::Display d;
    aln_hmapio::lay_out(s, flat, mark, d);
    if (d.t_sse != w_tsse || d.t_res != w_tres || d.marks != w_marks || d.q_res != w_qres || d.q_sse != w_qsse) {
      if (++bad < 4) fprintf(stderr, "layout differs (round %d)\n got  %s\n want %s\n got  %s\n want %s\n got  [%s]\n want [%s]\n", r, d.t_res.c_str(), w_tres.c_str(),
                             d.q_res.c_str(), w_qres.c_str(), d.marks.c_str(), w_marks.c_str());
    }
    // the gapped lines invert to the list
    List back;
    back.readFrom(w_qres, w_tres);
    if (back.size() != l.size() || !std::equal(back.begin(), back.end(), l.begin())) {
      // residues of a zig-zag or an insertion sit over gap columns, so only lists without query jumps invert exactly
      bool jumps = false;
      for (size_t k = 1; k < mark.size(); ++k) if (flat[2 * k] - flat[2 * k - 2] != 1) jumps = true;
      if (!jumps) ++bad;
    }
  }
  printf("LAYOUT rounds %d mismatches %d\n", rounds, bad);
  return bad;
}

static int check_readfrom() {
  int bad = 0;
  List l;
  //              q: ^AC-DE$   t: ^A-GDE$   pairs (0,0)(1,1)(3,3)(4,4)(5,5); aligned residue columns A/A D/D E/E -> identity 100
  l.readFrom("^AC-DE$", "^A-GDE$");
  const int want[][2] = {{0, 0}, {1, 1}, {3, 3}, {4, 4}, {5, 5}};
  if (l.size() != 5) ++bad;
  int k = 0;
  for (List::const_iterator it = l.begin(); it != l.end() && k < 5; ++it, ++k)
    if (it->query_idx() != want[k][0] || it->template_idx() != want[k][1]) ++bad;
  if (l.identity != 100.f || l.score != 0.f || l.uid != -1 || l.significance != 9999.f) ++bad;
  l.readFrom("^AC$", "^AD$");
  if (l.identity != 50.f) ++bad;
  bool threw = false;
  try { l.readFrom("^A$", "^AC$"); } catch (const char*) { threw = true; }
  if (!threw) ++bad;
  Set one(l);                                  // a set of one given alignment
  if (one.size() != 1 || one[0].size() != l.size()) ++bad;
  Set copy(one);
  if (copy.size() != 1) ++bad;
  printf("READFROM mismatches %d\n", bad);
  return bad;
}

int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 7u;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2000;
  int bad = check_layout(seed, rounds) + check_readfrom();
  matrix<float> m(2, 3);                       // matrix.h hands <valarray> and the std names on
  valarray<bool> v(false, 2);
  (void)m; (void)v;
  printf(bad ? "HOST UNIT FAILED\n" : "HOST UNIT OK\n");
  return bad ? 1 : 0;
}
