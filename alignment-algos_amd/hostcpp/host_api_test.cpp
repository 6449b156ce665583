// host_api_test.cpp — exercises the C++ host mirror end to end on the GPU (used by tests/test_gpu_hostcpp.py):
//   profile <mode> <q.hmap> <t.hmap>      DPMatrix<HMAPSequence,SMAPSequence,Hmap2Eval> + Optimal -> score bits, pairs, probes
//   aa <mode> <gi> <ge> <dir> <q> <t> <blosum>   DPMatrix<AASequence,...> fwd/rev + Optimal/Optimal_Rev + ucw, cells via getCell
//   set <mode> <gi> <ge> <blosum> <q1> <t1> [<q2> <t2> ...]   DPMatrixSet (all pairs in one launch) against one DPMatrix per pair:
//                                         cells, Optimal, ConstrainedNearOptimal — with AASubstitutionEval and with a plugin
//   gn2 <mode> <q.hmap> <t.hmap> <seed>   DPMatrix<HMAPSequence,SMAPSequence,Gn2Eval> with synthetic structural members; prints the
//                                         tables pre_calculate built (TAB name n v...) so a test can feed them to its checker
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include "aa_seq.h"
#include "aasubalib.h"
#include "cw.h"
#include "dpmatrix.h"
#include "dpmatrix_set.h"
#include "gn2_eval.h"
#include "hmap2_eval.h"
#include "optimal.h"
#include "optimal_rev.h"
#include "ucw.h"

// A plugin written against evaluator.h and nothing else (no aln_describe_gaps, no engine types): BLOSUM similarities, a
// gap cost that is NOT affine (square-root growth), the usual free-end rules.  It must run unchanged.
template <class S1, class S2>
class SqrtGapEval : public Evaluator<S1, S2, SqrtGapEval<S1, S2> > {
 public:
  SqrtGapEval(const AliParams& p, const SubstitutionMatrix& m) : params(&p), sub(&m) {}
  float similarity(const S1& q, const S2& t, int qi, int ti) const {
    if (q[qi]->isHead() || q[qi]->isTail() || t[ti]->isHead() || t[ti]->isTail()) return 0.f;
    return sub->score(q[qi]->olc, t[ti]->olc);
  }
  float deletion(const S1&, const S2& t, int, int, int t1, int t2) const {
    int len = t2 - t1 - 1;
    if (len < 1) return 0.f;
    if (free_del() && (t[t1]->isHead() || t[t2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * std::sqrt((float)len);
  }
  float insertion(const S1& q, const S2&, int q1, int q2, int, int) const {
    int len = q2 - q1 - 1;
    if (len < 1) return 0.f;
    if (free_ins() && (q[q1]->isHead() || q[q2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * std::sqrt((float)len);
  }
  void pre_calculate(const S1&, const S2&) const {}
  void post_process(SimilarityMatrix&) const {}
 private:
  bool free_del() const { return params->align_type == local || params->align_type == semi_local || params->align_type == local_global; }
  bool free_ins() const { return params->align_type == local || params->align_type == semi_local || params->align_type == global_local; }
  const AliParams* params;
  const SubstitutionMatrix* sub;
};

static unsigned fbits(float f);

// A plugin that names its gap model (aln_describe_gaps): half-BLOSUM similarities (fractional -> a SimilarityMatrix plane per
// pair, the exact-order kernels), constant affine gaps.
template <class S1, class S2>
class HalfBlosumEval : public Evaluator<S1, S2, HalfBlosumEval<S1, S2> > {
 public:
  HalfBlosumEval(const AliParams& p, const SubstitutionMatrix& m) : params(&p), sub(&m) {}
  float similarity(const S1& q, const S2& t, int qi, int ti) const {
    if (q[qi]->isHead() || q[qi]->isTail() || t[ti]->isHead() || t[ti]->isTail()) return 0.f;
    return 0.5f * sub->score(q[qi]->olc, t[ti]->olc) + 0.25f;
  }
  float deletion(const S1&, const S2& t, int, int, int t1, int t2) const {
    int len = t2 - t1 - 1;
    if (len < 1) return 0.f;
    if (free_end() && (t[t1]->isHead() || t[t2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * (float)(len - 1);
  }
  float insertion(const S1& q, const S2&, int q1, int q2, int, int) const {
    int len = q2 - q1 - 1;
    if (len < 1) return 0.f;
    if (free_end() && (q[q1]->isHead() || q[q2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * (float)(len - 1);
  }
  void pre_calculate(const S1&, const S2&) const {}
  void post_process(SimilarityMatrix&) const {}
  void aln_describe_gaps(const S1&, const S2&, aln::GapDescription& g) const {
    g.model = ALN_GAP_AFFINE_CONST; g.align_type = params->align_type;
    g.gap_init = params->gap_init_penalty; g.gap_extn = params->gap_extn_penalty;
  }
 private:
  bool free_end() const { return params->align_type == local || params->align_type == semi_local; }
  const AliParams* params;
  const SubstitutionMatrix* sub;
};

// A plugin that keeps per-pair state IN ITSELF: pre_calculate stores a shift derived from the two lengths, similarity() uses
// it.  The reference calls pre_calculate right before it builds a pair (dpmatrix.h:298); a set that called it for all pairs
// first would build every pair but the last with the wrong shift.
template <class S1, class S2>
class PairShiftEval : public Evaluator<S1, S2, PairShiftEval<S1, S2> > {
 public:
  PairShiftEval(const AliParams& p, const SubstitutionMatrix& m) : params(&p), sub(&m), shift(-100.f) {}
  float similarity(const S1& q, const S2& t, int qi, int ti) const {
    if (q[qi]->isHead() || q[qi]->isTail() || t[ti]->isHead() || t[ti]->isTail()) return 0.f;
    return sub->score(q[qi]->olc, t[ti]->olc) + shift;
  }
  float deletion(const S1&, const S2& t, int, int, int t1, int t2) const {
    const int len = t2 - t1 - 1;
    if (len < 1) return 0.f;
    if (free_end() && (t[t1]->isHead() || t[t2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * (float)(len - 1);
  }
  float insertion(const S1& q, const S2&, int q1, int q2, int, int) const {
    const int len = q2 - q1 - 1;
    if (len < 1) return 0.f;
    if (free_end() && (q[q1]->isHead() || q[q2]->isTail())) return 0.f;
    return params->gap_init_penalty + params->gap_extn_penalty * (float)(len - 1);
  }
  void pre_calculate(const S1& q, const S2& t) const { shift = 0.125f * (float)((q.size() * 7 + t.size() * 3) % 9) - 0.5f; }
  void post_process(SimilarityMatrix&) const {}
  void aln_describe_gaps(const S1&, const S2&, aln::GapDescription& g) const {
    g.model = ALN_GAP_AFFINE_CONST; g.align_type = params->align_type;
    g.gap_init = params->gap_init_penalty; g.gap_extn = params->gap_extn_penalty;
  }
 private:
  bool free_end() const { return params->align_type == local || params->align_type == semi_local; }
  const AliParams* params;
  const SubstitutionMatrix* sub;
  mutable float shift;
};

// DPMatrixSet against one DPMatrix per pair: every cell, the Optimal alignment, the ConstrainedNearOptimal set
template <class Eval>
static int compare_set(const char* tag, const std::vector<const AASequence*>& qv, const std::vector<const AASequence*>& tv, const Eval& ev, align_t type) {
  DPMatrixSet<AASequence, AASequence, Eval> set(qv, tv, ev, fwd, type);
  int bad = 0;
  NOaliParams noa;
  noa.number_suboptimal = 12; noa.delta_ratio = 0.2f;
  for (size_t p = 0; p < set.size(); ++p) {
    DPMatrix<AASequence, AASequence, Eval> dpm(*qv[p], *tv[p], ev, fwd, type);
    const int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < T; ++j) {
        const DPCell* a = dpm.getCell(i, j);
        const DPCell b = set.getCell(p, i, j);
        if (fbits(a->score) != fbits(b.score) || a->prev_query_idx != b.prev_query_idx || a->prev_template_idx != b.prev_template_idx) ++bad;
      }
    Optimal<AASequence, AASequence, Eval> opt(type);
    AlignmentSet<AASequence, AASequence, Eval> as(dpm, opt);
    typename DPMatrixSet<AASequence, AASequence, Eval>::Alignment o = set.optimal(p);
    if (fbits(o.score) != fbits(as[0].score) || o.size() != as[0].size() || !std::equal(o.begin(), o.end(), as[0].begin())) ++bad;
    if (fbits(set.scores()[p]) != fbits(as[0].score)) ++bad;
    SuboptFlags fl(true, (size_t)T);
    for (int k = T / 3; k < 2 * T / 3; ++k) fl.Set((unsigned)k, false);
    ConstrainedNearOptimal<AASequence, AASequence, Eval> cno(noa, fl);
    cno.enumerate(dpm, as);                     // (returns the sorted set)
    std::vector<typename DPMatrixSet<AASequence, AASequence, Eval>::Alignment> es = set.enumerate(p, noa, &fl);
    if (es.size() != as.size()) ++bad;
    for (size_t k = 0; k < es.size() && k < as.size(); ++k)
      if (fbits(es[k].score) != fbits(as[k].score) || es[k].size() != as[k].size() || !std::equal(es[k].begin(), es[k].end(), as[k].begin())) ++bad;
  }
  printf("SET %s pairs %d mismatches %d\n", tag, (int)set.size(), bad);
  return bad;
}

static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

template <class Set>
static void dump(const char* tag, Set& as) {
  printf("%s %d\n", tag, (int)as.size());
  for (size_t a = 0; a < as.size(); ++a) {
    printf("ALI %08x %08x %d %d", fbits(as[a].score), fbits(as[a].identity), as[a].uid, (int)as[a].size());
    for (typename Set::value_type::const_iterator it = as[a].begin(); it != as[a].end(); ++it) printf(" %d %d", it->first, it->second);
    printf("\n");
  }
}

int main(int argc, char** argv) {
  try {
    std::string cmd = argv[1];
    if (cmd == "profile") {
      Gn2Params p;
      p.align_type = (align_t)atoi(argv[2]);
      HMAPSequence q(argv[3]);
      SMAPSequence t(argv[4]);
      Hmap2Eval ev(p);
      DPMatrix<HMAPSequence, SMAPSequence, Hmap2Eval> dpm(q, t, ev, fwd, p.align_type);
      Optimal<HMAPSequence, SMAPSequence, Hmap2Eval> opt(p.align_type);
      AlignmentSet<HMAPSequence, SMAPSequence, Hmap2Eval> as(dpm, opt);
      dump("OPT", as);
      const int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
      printf("DIM %d %d\nH", Q, T);
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score));
      printf("\nS");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getSim(i, j)));
      printf("\n");
      return 0;
    }
    if (cmd == "gn2") {
      Gn2Params p;
      p.align_type = (align_t)atoi(argv[2]);
      HMAPSequence q(argv[3]);
      SMAPSequence t(argv[4]);
      unsigned long long st = strtoull(argv[5], 0, 10) * 2862933555777941757ull + 3037000493ull;
      auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xFFFFFF) / 16777216.0f; };
      const int T = (int)t.size(), n = (int)t.seq_length;
      // structural members the reference derives from a PDB file through Troll: synthetic here
      t.weighted_contact_number.resize(T);
      for (int i = 0; i < T; ++i) { t.weighted_contact_number[i] = 0.2f + rnd(); t[i]->hydropathy = rnd(); t[i]->rdata.isse = (int)(rnd() * 6) - 1; }
      for (size_t i = 0; i < q.size(); ++i) q[i]->hydropathy = rnd();
      t.distance.resize(n); t.brokenhb.resize(n);
      for (int i = 2; i < n + 2; ++i) {
        t.distance[i - 2].resize(i - 1); t.brokenhb[i - 2].resize(i - 1);
        for (int j = 0; j < i - 1; ++j) { t.distance[i - 2][j] = 3.f + 25.f * rnd(); t.brokenhb[i - 2][j] = (unsigned long)(rnd() * 4); }
      }
      Gn2Eval ev(p);
      DPMatrix<HMAPSequence, SMAPSequence, Gn2Eval> dpm(q, t, ev, fwd, p.align_type);
      Optimal<HMAPSequence, SMAPSequence, Gn2Eval> opt(p.align_type);
      AlignmentSet<HMAPSequence, SMAPSequence, Gn2Eval> as(dpm, opt);
      dump("OPT", as);
      const int Q = dpm.getQuerySize();
      printf("DIM %d %d\nH", Q, T);
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score));
      printf("\nPQ");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_query_idx);
      printf("\nPT");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_template_idx);
      printf("\nS");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getSim(i, j)));
      printf("\n");
      auto vec = [&](const char* name, const std::vector<float>& v) { printf("TAB %s", name); for (int j = 0; j < T; ++j) printf(" %08x", fbits(j < (int)v.size() ? v[j] : 0.f)); printf("\n"); };
      vec("v_gi", ev.v_gi); vec("v_ge", ev.v_ge); vec("v_cn", ev.v_cn);
      auto tri = [&](const char* name, const std::vector<std::vector<float> >& v) {   // T x T, [p2*T + p1], zero where undefined
        printf("TAB %s", name);
        for (int p2 = 0; p2 < T; ++p2) for (int p1 = 0; p1 < T; ++p1) printf(" %08x", fbits(p2 < (int)v.size() && p1 < (int)v[p2].size() ? v[p2][p1] : 0.f));
        printf("\n");
      };
      tri("dist", t.distance); tri("vv_gi", ev.vv_gi); tri("vv_ge", ev.vv_ge); tri("vv_cd", ev.vv_cd);
      return 0;
    }
    if (cmd == "setprofile") {  // setprofile <mode> <q1.hmap> <t1.hmap> [<q2.hmap> <t2.hmap> ...]: Hmap2Eval pairs in one launch
      Gn2Params p;
      p.align_type = (align_t)atoi(argv[2]);
      std::vector<HMAPSequence*> qo;
      std::vector<SMAPSequence*> to;
      std::vector<const HMAPSequence*> qv;
      std::vector<const SMAPSequence*> tv;
      for (int a = 3; a + 1 < argc; a += 2) { qo.push_back(new HMAPSequence(argv[a])); to.push_back(new SMAPSequence(argv[a + 1])); qv.push_back(qo.back()); tv.push_back(to.back()); }
      Hmap2Eval ev(p);
      DPMatrixSet<HMAPSequence, SMAPSequence, Hmap2Eval> set(qv, tv, ev, fwd, p.align_type);
      int bad = 0;
      for (size_t k = 0; k < set.size(); ++k) {
        DPMatrix<HMAPSequence, SMAPSequence, Hmap2Eval> dpm(*qv[k], *tv[k], ev, fwd, p.align_type);
        const int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
        for (int i = 0; i < Q; ++i)
          for (int j = 0; j < T; ++j) {
            const DPCell* a = dpm.getCell(i, j);
            const DPCell b = set.getCell(k, i, j);
            if (fbits(a->score) != fbits(b.score) || a->prev_query_idx != b.prev_query_idx || a->prev_template_idx != b.prev_template_idx) ++bad;
          }
        if (fbits(dpm.getSim(Q / 2, T / 2)) != fbits(set.getSim(k, Q / 2, T / 2))) ++bad;
        Optimal<HMAPSequence, SMAPSequence, Hmap2Eval> opt(p.align_type);
        AlignmentSet<HMAPSequence, SMAPSequence, Hmap2Eval> as(dpm, opt);
        DPMatrixSet<HMAPSequence, SMAPSequence, Hmap2Eval>::Alignment o = set.optimal(k);
        if (fbits(o.score) != fbits(as[0].score) || o.size() != as[0].size() || !std::equal(o.begin(), o.end(), as[0].begin())) ++bad;
      }
      printf("SET profile pairs %d mismatches %d\n%s\n", (int)set.size(), bad, bad ? "SET FAILED" : "SET OK");
      for (size_t k = 0; k < qo.size(); ++k) { delete qo[k]; delete to[k]; }
      return 0;
    }
    if (cmd == "set") {         // set <mode> <gi> <ge> <blosum> <q1> <t1> [<q2> <t2> ...]
      AliParams p;
      p.align_type = (align_t)atoi(argv[2]);
      p.gap_init_penalty = (float)atof(argv[3]);
      p.gap_extn_penalty = (float)atof(argv[4]);
      BlosumMatrix blosum(argv[5]);
      std::vector<AASequence*> own;
      std::vector<const AASequence*> qv, tv;
      for (int a = 6; a + 1 < argc; a += 2) {
        AASequence* q = new AASequence; AASequence* t = new AASequence;
        q->seq_name = "query"; t->seq_name = "templ";
        q->append("^"); q->append(argv[a]); q->append("$");
        t->append("^"); t->append(argv[a + 1]); t->append("$");
        own.push_back(q); own.push_back(t); qv.push_back(q); tv.push_back(t);
      }
      AASubstitutionEval<AASequence, AASequence> ev(p, blosum);
      HalfBlosumEval<AASequence, AASequence> hv(p, blosum);
      int bad = compare_set("aasub", qv, tv, ev, p.align_type);
      bad += compare_set("plugin", qv, tv, hv, p.align_type);
      PairShiftEval<AASequence, AASequence> pv(p, blosum);
      bad += compare_set("pairstate", qv, tv, pv, p.align_type);
      try {                       // an evaluator whose gap functions are tabulated per pair is refused, loudly
        SqrtGapEval<AASequence, AASequence> sq(p, blosum);
        DPMatrixSet<AASequence, AASequence, SqrtGapEval<AASequence, AASequence> > s2(qv, tv, sq, fwd, p.align_type);
        printf("SET tables: no refusal\n"); ++bad;
      } catch (std::string e) { printf("SET tables refused: %s\n", e.c_str()); }
      for (size_t k = 0; k < own.size(); ++k) delete own[k];
      printf(bad ? "SET FAILED\n" : "SET OK\n");
      return 0;
    }
    if (cmd == "plain") {       // plain <mode> <gi> <ge> <q> <t> <blosum>: the unmodified plugin above; prints H/PQ/PT, S and its gap tables
      AliParams p;
      p.align_type = (align_t)atoi(argv[2]);
      p.gap_init_penalty = (float)atof(argv[3]);
      p.gap_extn_penalty = (float)atof(argv[4]);
      AASequence q, t;
      q.seq_name = "query"; t.seq_name = "templ";
      q.append("^"); q.append(argv[5]); q.append("$");
      t.append("^"); t.append(argv[6]); t.append("$");
      BlosumMatrix blosum(argv[7]);
      typedef SqrtGapEval<AASequence, AASequence> PEval;
      PEval ev(p, blosum);
      DPMatrix<AASequence, AASequence, PEval> dpm(q, t, ev, fwd, p.align_type);
      Optimal<AASequence, AASequence, PEval> opt(p.align_type);
      AlignmentSet<AASequence, AASequence, PEval> as(dpm, opt);
      dump("OPT", as);
      const int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
      printf("DIM %d %d\nH", Q, T);
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score));
      printf("\nPQ");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_query_idx);
      printf("\nPT");
      for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_template_idx);
      printf("\n");
      return 0;
    }
    AliParams p;
    p.align_type = (align_t)atoi(argv[2]);
    p.gap_init_penalty = (float)atof(argv[3]);
    p.gap_extn_penalty = (float)atof(argv[4]);
    direction_t dir = strcmp(argv[5], "rev") == 0 ? rev : fwd;
    AASequence q, t;
    q.seq_name = "query"; t.seq_name = "templ";
    q.append("^"); q.append(argv[6]); q.append("$");
    t.append("^"); t.append(argv[7]); t.append("$");
    BlosumMatrix blosum(argv[8]);
    typedef AASubstitutionEval<AASequence, AASequence> AAEval;
    AAEval ev(p, blosum);
    DPMatrix<AASequence, AASequence, AAEval> dpm(q, t, ev, dir, p.align_type);
    const int Q = dpm.getQuerySize(), T = dpm.getTemplateSize();
    printf("DIM %d %d\nH", Q, T);
    for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score));
    printf("\nPQ");
    for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_query_idx);
    printf("\nPT");
    for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_template_idx);
    printf("\n");
    if (dir == fwd) {
      Optimal<AASequence, AASequence, AAEval> opt(p.align_type);
      AlignmentSet<AASequence, AASequence, AAEval> as(dpm, opt);
      dump("OPT", as);
      NOaliParams noa;
      noa.number_suboptimal = 20; noa.delta_ratio = 0.3f;
      UnconstrainedNearOptimal<AASequence, AASequence, AAEval> u(noa);
      u.enumerate(dpm, as);
      as.assignIdentity();
      dump("UCW", as);
    } else {
      Optimal_Rev<AASequence, AASequence, AAEval> opt(p.align_type);
      AlignmentSet<AASequence, AASequence, AAEval> as(dpm, opt);
      dump("OPT", as);
    }
    return 0;
  } catch (std::string e) {
    printf("THROW %s\n", e.c_str());
    return 0;
  }
}
