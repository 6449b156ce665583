// TEST INFRASTRUCTURE — not part of the product.
//
// Harness around the REAL reference implementation (christang/alignment-algos),
// compiled from the sources where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/ref_harness.  Nothing from the reference is
// copied here: this file only #includes the reference's public headers and
// drives its public API (DPMatrix ctor, getCell, getSim, Optimal,
// ConstrainedNearOptimal, UnconstrainedNearOptimal, SequenceGaps,
// AlignmentSet::assignIdentity) the way aa_ali.cpp:56-92 does.
//
// It is used only (a) by oracle/gen_golden.py in the build container to
// produce tests/golden/*.json, (b) by tests that cross-check oracle/
// (our restatement) when /root/reference exists, and (c) optionally as the
// "reference" CPU baseline in bench.py (the binary travels, the sources do not).
//
// Usage: ref_harness <cmd> ...   (all results on stdout, floats as hex bit patterns)
//   aa <blosum> <align_t> <gi> <ge> <fwd|rev> <query> <templ> [dump] [opt] [cw N delta flags] [ucw N delta] [cwcount delta flags]
//   sub <blosum> <align_t> <gi> <ge> <fwd|rev> <query> <templ> q1 t1 q2 t2   (7-arg ctor + Optimal_Subali)
//   time <blosum> <align_t> <gi> <ge> <len> <seed> <npairs>    (times the DPMatrix ctor)
//   block <blosum> <align_t> <gi> <ge> <seqfile> r0 r1 c0 c1   (score Optimal reports for every (row, col) of a sequence set)
// extra ops of `aa`: bin <path> (planes as raw little-endian int32: Q T, H bits, PQ, PT — for full-size sha256 goldens)

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <sstream>
#include <iostream>
#include <chrono>
#include <fstream>
#include <vector>

#include "aa_seq.h"
#include "aasubalib.h"
#include "cw.h"
#include "ucw.h"
#include "dpmatrix.h"
#include "gstrings.h"
#include "noalib.h"
#include "optimal.h"
#include "optimal_rev.h"
#include "optimal_subali.h"
#include "sequence.h"
#include "sflags.h"

typedef AASubstitutionEval<AASequence, AASequence> AAEval;
typedef DPMatrix<AASequence, AASequence, AAEval> AADpm;
typedef AlignmentSet<AASequence, AASequence, AAEval> AASet;

// Optimal_Rev is abstract as shipped (optimal_rev.h:29-30 does not override
// enumerator.h:23); this subclass only forwards to its const methods.
struct OptimalRevUsable : public Optimal_Rev<AASequence, AASequence, AAEval> {
  OptimalRevUsable(align_t t) : Optimal_Rev<AASequence, AASequence, AAEval>(t) {}
  void enumerate(AADpm& dpm, AASet& as) {
    const AADpm& c = dpm;
    Optimal_Rev<AASequence, AASequence, AAEval>::enumerate(c, as);
  }
};

static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static void fill_seq(AASequence& s, const std::string& name, const std::string& res) {
  s.seq_name = name;
  s.append("^");
  s.append(res);
  s.append("$");
}

static void dump_matrix(const AADpm& dpm) {
  int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
  printf("DIM %d %d\n", Q, T);
  printf("H");
  for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score));
  printf("\nPQ");
  for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_query_idx);
  printf("\nPT");
  for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_template_idx);
  printf("\nS");
  for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getSim(i, j)));
  printf("\n");
}

static void dump_bin(const AADpm& dpm, const char* path) {
  int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
  FILE* f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
  int hdr[2] = {Q, T};
  fwrite(hdr, 4, 2, f);
  std::vector<int> row(T);
  for (int pl = 0; pl < 3; ++pl)
    for (int i = 0; i < Q; ++i) {
      for (int j = 0; j < T; ++j) {
        const DPCell* c = dpm.getCell(i, j);
        row[j] = pl == 0 ? (int)fbits(c->score) : pl == 1 ? c->prev_query_idx : c->prev_template_idx;
      }
      fwrite(row.data(), 4, T, f);
    }
  fclose(f);
}

static void dump_set(const char* tag, AASet& as) {
  as.assignIdentity();
  printf("%s %d\n", tag, (int)as.size());
  // SequenceGaps (gstrings.h:118-164) walks past the list end unless every alignment ends with
  // the tail pair; Optimal_Rev::enumerate_local can produce such lists.  Guard instead of crashing.
  bool printable = true;
  int Q = (int)as.getQuerySequence()->size(), T = (int)as.getTemplateSequence()->size();
  for (size_t a = 0; a < as.size(); ++a)
  {
    if (as[a].empty() || as[a].back().query_idx() != Q - 1 || as[a].back().template_idx() != T - 1) printable = false;
    // a repeated pair (Optimal_Rev local seeds (0,0) twice) makes gstrings.h:150 run transform() on begin()+1 of an empty string
    std::list<AlignedPair<AASequence, AASequence> >::const_iterator it = as[a].begin(), nx = it;
    if (it != as[a].end()) for (++nx; nx != as[a].end(); ++it, ++nx) if (*it == *nx) printable = false;
  }
  std::string ts;
  if (printable) {
    SequenceGaps gaps(as);
    gaps.build(*as.getTemplateSequence()->getString(), ts);
    printf("TSTR %s\n", ts.c_str());
  } else printf("NOSTR\n");
  for (size_t a = 0; a < as.size(); ++a) {
    printf("ALI %08x %08x %d %d", fbits(as[a].score), fbits(as[a].identity), as[a].uid, (int)as[a].size());
    for (std::list<AlignedPair<AASequence, AASequence> >::const_iterator it = as[a].begin(); it != as[a].end(); ++it)
      printf(" %d %d", it->query_idx(), it->template_idx());
    printf("\n");
    if (printable) {
      SequenceGaps gaps(as);
      std::string qs;
      gaps.build(*as.getQuerySequence()->getString(), as[a], qs);
      printf("QSTR %s\n", qs.c_str());
    }
    // FASTA annotation exactly as fastaio.h:79-91 formats it (default stream precision)
    std::stringstream buff("");
    buff << "(sc=" << as[a].score << ",ev=" << as[a].significance << ",id=" << as[a].identity << "%)";
    printf("ANNOT %s\n", buff.str().c_str());
  }
}

static std::string synth(std::mt19937& g, int n) {
  static const char* A = "ARNDCQEGHILKMFPSTWYV";
  std::string s(n, 'A');
  for (int i = 0; i < n; ++i) s[i] = A[g() % 20];
  return s;
}

int main(int argc, char** argv) {
  try {
    if (argc < 2) { fprintf(stderr, "usage\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "time") {
      BlosumMatrix blosum(argv[2]);
      AliParams p;
      p.align_type = (align_t)atoi(argv[3]);
      p.gap_init_penalty = (float)atof(argv[4]);
      p.gap_extn_penalty = (float)atof(argv[5]);
      int len = atoi(argv[6]); unsigned seed = (unsigned)atoi(argv[7]); int np = atoi(argv[8]);
      AAEval ev(p, blosum);
      double total = 0; double cells = 0;
      for (int k = 0; k < np; ++k) {
        std::mt19937 g(seed + k);
        AASequence q, t;
        fill_seq(q, "q", synth(g, len));
        fill_seq(t, "t", synth(g, len));
        auto t0 = std::chrono::steady_clock::now();
        AADpm dpm(q, t, ev, fwd, p.align_type);
        auto t1 = std::chrono::steady_clock::now();
        total += std::chrono::duration<double>(t1 - t0).count();
        cells += (double)len * len;
        printf("PAIR %d corner %08x\n", k, fbits(dpm.getCell(len + 1, len + 1)->score));
      }
      printf("TIME %.6f CELLS %.0f\n", total, cells);
      return 0;
    }
    if (cmd == "block") {
      BlosumMatrix blosum(argv[2]);
      AliParams p;
      p.align_type = (align_t)atoi(argv[3]);
      p.gap_init_penalty = (float)atof(argv[4]);
      p.gap_extn_penalty = (float)atof(argv[5]);
      std::ifstream in(argv[6]);
      std::vector<std::string> seqs;
      std::string line;
      while (std::getline(in, line)) seqs.push_back(line);   // an empty line is an empty sequence
      int r0 = atoi(argv[7]), r1 = atoi(argv[8]), c0 = atoi(argv[9]), c1 = atoi(argv[10]);
      AAEval ev(p, blosum);
      for (int r = r0; r < r1; ++r) {
        printf("ROW %d", r);
        for (int c = c0; c < c1; ++c) {
          AASequence q, t;
          fill_seq(q, "q", seqs[r]);
          fill_seq(t, "t", seqs[c]);
          AADpm dpm(q, t, ev, fwd, p.align_type);
          Optimal<AASequence, AASequence, AAEval> opt(p.align_type);
          AASet as(dpm, opt);
          printf(" %08x", fbits(as[0].score));
        }
        printf("\n");
        fflush(stdout);
      }
      return 0;
    }
    if (cmd != "aa" && cmd != "sub") { fprintf(stderr, "unknown cmd\n"); return 2; }
    BlosumMatrix blosum(argv[2]);
    AliParams p;
    p.align_type = (align_t)atoi(argv[3]);
    p.gap_init_penalty = (float)atof(argv[4]);
    p.gap_extn_penalty = (float)atof(argv[5]);
    direction_t dir = strcmp(argv[6], "rev") == 0 ? rev : fwd;
    AASequence q, t;
    fill_seq(q, "query", argv[7]);
    fill_seq(t, "templ", argv[8]);
    AAEval ev(p, blosum);
    int a = 9;
    if (cmd == "sub") {
      int q1 = atoi(argv[a]), t1 = atoi(argv[a + 1]), q2 = atoi(argv[a + 2]), t2 = atoi(argv[a + 3]);
      AADpm dpm(q, t, ev, q1, t1, q2, t2, dir, p.align_type);
      dump_matrix(dpm);
      if (dir == fwd) {
        Optimal_Subali<AASequence, AASequence, AAEval> os(q1, t1, q2, t2);
        AASet as(dpm, os);
        printf("SUBALI %08x %d", fbits(as[0].score), (int)as[0].size());
        for (std::list<AlignedPair<AASequence, AASequence> >::const_iterator it = as[0].begin(); it != as[0].end(); ++it)
          printf(" %d %d", it->query_idx(), it->template_idx());
        printf("\n");
      }
      return 0;
    }
    auto tc0 = std::chrono::steady_clock::now();
    AADpm dpm(q, t, ev, dir, p.align_type);
    auto tc1 = std::chrono::steady_clock::now();
    for (; a < argc; ++a) {
      std::string op = argv[a];
      if (op == "dump") dump_matrix(dpm);
      else if (op == "bin") dump_bin(dpm, argv[++a]);
      else if (op == "ctime") printf("CTIME %.6f\n", std::chrono::duration<double>(tc1 - tc0).count());
      else if (op == "corner") {
        printf("CORNER %08x %08x\n", fbits(dpm.getCell(dpm.getQuerySize() - 1, dpm.getTemplateSize() - 1)->score),
               fbits(dpm.getCell(0, 0)->score));
      } else if (op == "opt") {
        if (dir == fwd) {
          Optimal<AASequence, AASequence, AAEval> opt(p.align_type);
          AASet as(dpm, opt);
          dump_set("OPT", as);
        } else {
          OptimalRevUsable opt(p.align_type);
          AASet as(dpm, opt);
          dump_set("OPT", as);
        }
      } else if (op == "cwcount") {   // cwcount delta flags: alignments ConstrainedNearOptimal creates before sortSet cuts the set
        NOaliParams noa;
        noa.number_suboptimal = 2000000000;
        noa.delta_ratio = (float)atof(argv[a + 1]);
        Optimal<AASequence, AASequence, AAEval> opt(p.align_type);
        AASet as(dpm, opt);
        std::string fl = argv[a + 2];
        SuboptFlags sf(true, t.size());
        for (size_t i = 0; i < fl.size(); ++i) sf.Set(i, fl[i] != '0');
        ConstrainedNearOptimal<AASequence, AASequence, AAEval> c(noa, sf);
        c.enumerate(dpm, as);
        size_t pairs_total = 0;
        for (size_t k = 0; k < as.size(); ++k) pairs_total += as[k].size();
        printf("CWCOUNT %d %zu\n", (int)as.size(), pairs_total);
        a += 2;
      } else if (op == "cw" || op == "ucw") {
        NOaliParams noa;
        noa.number_suboptimal = atoi(argv[a + 1]);
        noa.delta_ratio = (float)atof(argv[a + 2]);
        Optimal<AASequence, AASequence, AAEval> opt(p.align_type);
        AASet as(dpm, opt);   // drivers seed the set with the optimal (aa_ali.cpp:83)
        if (op == "cw") {
          std::string fl = argv[a + 3];
          SuboptFlags sf(true, t.size());   // (bool,len) as nalign.cpp:84; NOT aa_ali.cpp:86's swapped form
          if (fl.size() != t.size()) { fprintf(stderr, "flags length\n"); return 2; }
          for (size_t i = 0; i < fl.size(); ++i) sf.Set(i, fl[i] != '0');
          ConstrainedNearOptimal<AASequence, AASequence, AAEval> c(noa, sf);
          c.enumerate(dpm, as);
          dump_set("CW", as);
          a += 3;
        } else {
          UnconstrainedNearOptimal<AASequence, AASequence, AAEval> c(noa);
          c.enumerate(dpm, as);
          dump_set("UCW", as);
          a += 2;
        }
      }
    }
    return 0;
  } catch (std::string e) {
    printf("THROW %s\n", e.c_str());
    return 0;
  }
}
