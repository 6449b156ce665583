// TEST INFRASTRUCTURE — not part of the product.
//
// Profile-path pinning harness.  Hmap2Eval / HMAPaliEval themselves cannot be compiled from the reference:
// hmap2_eval.h -> gn2_eval.h -> gn2lib_seq.h / struct.h need the Honig-lab "Troll" library, which is not in
// /root/reference nor in this image and must not be stubbed.  What CAN be compiled in place is
// everything around that glue: the REAL math primitives (hmath.h: dot_product, pearson_corr, norm_elements,
// shift_elements), the REAL SimilarityMatrix / DPMatrix / Optimal templates, driven through the reference's
// own plugin API by the evaluator below, which is written against evaluator.h the way a user plugin is and
// states the formulas of hmap2_eval.h:27-95 / hmap2_eval.cpp:17-25 over a self-contained profile element.
// So: hmath arithmetic, the DP with position-dependent min() gaps, normalisation and tracebacks are pinned by
// the reference's own code; only "Hmap2Eval uses exactly these formulas" is our reading of the source.
//
// stdin:  mode alpha beta zero_shift gi ge dir(1|2)
//         Q   then Q lines:  olc  aa[20]  sse[3]  conf
//         T   then T lines:  olc  aa[20]  sse[3]  conf
// stdout: DIM, PRIM (dot/pearson probes), TGI/TGE (pre_calculate), S (post-processed), H, PQ, PT, OPT ...
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <valarray>
#include <vector>

#include "alib.h"
#include "alignment.h"
#include "dpmatrix.h"
#include "evaluator.h"
#include "hmath.h"
#include "optimal.h"
#include "sequence.h"

struct ProfElem : public SequenceElem {
  std::valarray<float> aa_profile, sse_values;
  float sse_confid;
  float gap_values[2];
  ProfElem() : aa_profile(20), sse_values(3), sse_confid(0.f) { gap_values[0] = gap_values[1] = 0.f; }
  float gap_init() const { return gap_values[0]; }
  float gap_extn() const { return gap_values[1]; }
  void gap_init(float g) { gap_values[0] = g; }
  void gap_extn(float g) { gap_values[1] = g; }
  float p_coil() const { return sse_values[2]; }
};

class ProfSequence : public Sequence<ProfElem*> {};

struct ProfParams {
  align_t align_type;
  float alpha, beta, zero_shift, gap_init_penalty, gap_extn_penalty;
};

class ProfEval : public Evaluator<ProfSequence, ProfSequence, ProfEval> {
 public:
  ProfEval(ProfParams& p) : params(&p) {}
  inline float similarity(const ProfSequence& q, const ProfSequence& t, int q_pos, int t_pos) const {
    float ip = dot_product(q[q_pos]->aa_profile, t[t_pos]->aa_profile);
    float pc = pearson_corr(q[q_pos]->sse_values, t[t_pos]->sse_values);
    float sim = ip * exp(params->alpha * pc * q[q_pos]->sse_confid * t[t_pos]->sse_confid);
    return sim;
  }
  inline float deletion(const ProfSequence& q, const ProfSequence& t, int q_pos1, int q_pos2, int t_pos1, int t_pos2) const {
    int dist = t_pos2 - t_pos1;
    if (dist < 2) return 0;
    float gi, ge;
    gi = min(t[t_pos1]->gap_init(), t[t_pos2]->gap_init());
    ge = min(t[t_pos1]->gap_extn(), t[t_pos2]->gap_extn());
    switch (params->align_type) {
      case global: case global_local: return gi + ge * (dist - 2);
      case local: case semi_local: case local_global:
        if (t[t_pos1]->isHead() || t[t_pos2]->isTail()) return 0;
        else return gi + ge * (dist - 2);
      default: throw string("Illegal gap style");
    }
  }
  inline float insertion(const ProfSequence& q, const ProfSequence& t, int q_pos1, int q_pos2, int t_pos1, int t_pos2) const {
    int dist = q_pos2 - q_pos1;
    if (dist < 2) return 0;
    float gi, ge;
    gi = min(t[t_pos1]->gap_init(), t[t_pos2]->gap_init());
    ge = min(t[t_pos1]->gap_extn(), t[t_pos2]->gap_extn());
    switch (params->align_type) {
      case global: case local_global: return gi + ge * (dist - 2);
      case local: case semi_local: case global_local:
        if (q[q_pos1]->isHead() || q[q_pos2]->isTail()) return 0;
        else return gi + ge * (dist - 2);
      default: throw string("Illegal gap style");
    }
  }
  void pre_calculate(const ProfSequence& s1, const ProfSequence& s2) const {
    for (unsigned int i = 0; i < s2.size(); ++i) {
      float Pi = exp(params->beta * (1.f - 1.25f * s2[i]->p_coil()));
      s2[i]->gap_init(params->gap_init_penalty * Pi);
      s2[i]->gap_extn(params->gap_extn_penalty * Pi);
    }
  }
  inline void post_process(SimilarityMatrix& s) const {
    norm_elements(s, s, 1, s.rows() - 1, 1, s.cols() - 1);
    shift_elements(s, s, 1, s.rows() - 1, 1, s.cols() - 1, -params->zero_shift);
  }
 private:
  ProfParams* params;
};

static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

static void read_seq(ProfSequence& s, int n) {
  for (int i = 0; i < n; ++i) {
    ProfElem* e = new ProfElem();
    char olc[8];
    if (scanf("%7s", olc) != 1) exit(3);
    e->olc = olc[0]; e->index = i;
    for (int k = 0; k < 20; ++k) if (scanf("%f", &e->aa_profile[k]) != 1) exit(3);
    for (int k = 0; k < 3; ++k) if (scanf("%f", &e->sse_values[k]) != 1) exit(3);
    if (scanf("%f", &e->sse_confid) != 1) exit(3);
    s.push_back(e);
  }
}

int main() {
  try {
    ProfParams p;
    int mode, dir;
    if (scanf("%d %f %f %f %f %f %d", &mode, &p.alpha, &p.beta, &p.zero_shift, &p.gap_init_penalty, &p.gap_extn_penalty, &dir) != 7) return 3;
    p.align_type = (align_t)mode;
    int Q, T;
    ProfSequence q, t;
    if (scanf("%d", &Q) != 1) return 3;
    read_seq(q, Q);
    if (scanf("%d", &T) != 1) return 3;
    read_seq(t, T);
    printf("DIM %d %d\n", Q, T);
    // probes of the real hmath.h primitives on the inputs (before pre_calculate changes nothing they use)
    printf("PRIM");
    for (int i = 1; i < Q - 1 && i < 6; ++i)
      for (int j = 1; j < T - 1 && j < 6; ++j)
        printf(" %08x %08x", fbits(dot_product(q[i]->aa_profile, t[j]->aa_profile)), fbits(pearson_corr(q[i]->sse_values, t[j]->sse_values)));
    printf("\n");
    ProfEval ev(p);
    DPMatrix<ProfSequence, ProfSequence, ProfEval> dpm(q, t, ev, dir == 2 ? rev : fwd, p.align_type);
    printf("TGI"); for (int j = 0; j < T; ++j) printf(" %08x", fbits(t[j]->gap_init())); printf("\n");
    printf("TGE"); for (int j = 0; j < T; ++j) printf(" %08x", fbits(t[j]->gap_extn())); printf("\n");
    if (const char* path = getenv("REF_PROFILE_BIN")) {   // full-size goldens: raw int32 planes Q T, S bits, H bits, PQ, PT
      FILE* f = fopen(path, "wb");
      if (!f) return 4;
      int hdr[2] = {Q, T};
      fwrite(hdr, 4, 2, f);
      std::vector<int> row(T);
      for (int pl = 0; pl < 4; ++pl)
        for (int i = 0; i < Q; ++i) {
          for (int j = 0; j < T; ++j)
            row[j] = pl == 0 ? (int)fbits(dpm.getSim(i, j)) : pl == 1 ? (int)fbits(dpm.getCell(i, j)->score)
                   : pl == 2 ? dpm.getCell(i, j)->prev_query_idx : dpm.getCell(i, j)->prev_template_idx;
          fwrite(row.data(), 4, T, f);
        }
      fclose(f);
    } else {
    printf("S"); for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getSim(i, j))); printf("\n");
    printf("H"); for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %08x", fbits(dpm.getCell(i, j)->score)); printf("\n");
    printf("PQ"); for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_query_idx); printf("\n");
    printf("PT"); for (int i = 0; i < Q; ++i) for (int j = 0; j < T; ++j) printf(" %d", dpm.getCell(i, j)->prev_template_idx); printf("\n");
    }
    if (dir != 2) {
      Optimal<ProfSequence, ProfSequence, ProfEval> opt(p.align_type);
      AlignmentSet<ProfSequence, ProfSequence, ProfEval> as(dpm, opt);
      printf("OPT 1\nNOSTR\nALI %08x %08x %d %d", fbits(as[0].score), fbits(as[0].identity), as[0].uid, (int)as[0].size());
      for (std::list<AlignedPair<ProfSequence, ProfSequence> >::const_iterator it = as[0].begin(); it != as[0].end(); ++it)
        printf(" %d %d", it->query_idx(), it->template_idx());
      printf("\n");
    }
    return 0;
  } catch (std::string e) {
    printf("THROW %s\n", e.c_str());
    return 0;
  }
}
