// hmap_eval.h / hmap2_eval.h — profile-profile evaluators (reference hmap_eval.h:39-126 + hmap_eval.cpp:4-51,
// hmap2_eval.h:19-109 + hmap2_eval.cpp:17-25; the two are arithmetically identical).
//   similarity   = dot20(q.aa, t.aa) * exp(alpha * pearson3(q.sse, t.sse) * q.conf * t.conf)
//   post_process = z-normalise the interior of the matrix, then add -zero_shift
//   pre_calculate: template gap_init/extn = GAP_*_PENALTY * exp(beta * (1 - 1.25 p_coil))   (mutates the template, as the reference does)
//   deletion/insertion: min over the two template positions of gap_init / gap_extn, affine in the gap length, free end gaps per align_t
// On this engine the whole evaluator lowers to ALN_SIM_HMAP2 + ALN_GAP_AFFINE_TPOS_MIN (aln_lowering.h): similarity and
// normalisation run on the device, pre_calculate on the host (aln_hmap2_gap_arrays uses the host libm like the reference).
#ifndef ALN_HOST_HMAP_EVAL_H
#define ALN_HOST_HMAP_EVAL_H
#include <algorithm>
#include <cmath>
#include <string>
#include "alib.h"
#include "aln_lowering.h"
#include "evaluator.h"
#include "hmapalib_seq.h"
#include "noalib.h"

class HMAPaliParams : public AliParams, public NOaliParams {
 public:
  HMAPaliParams() : alpha(0.5f), beta(1.0f), gamma(0.1f), normalize_mtx(true), zero_shift(0.12f) {}
  void read(ParamStore* p) {
    std::string s;
    s = "CORE_MATCH_WEIGHT"; if (p->find(s)) p->getValue(s) >> alpha;
    s = "CORE_GAP_WEIGHT"; if (p->find(s)) p->getValue(s) >> beta;
    s = "MOTIF_MATCH_WEIGHT"; if (p->find(s)) p->getValue(s) >> gamma;
    s = "NORMALIZE_SIM_MTX"; if (p->find(s)) p->getValue(s) >> normalize_mtx;
    s = "ZERO_SHIFT"; if (p->find(s)) p->getValue(s) >> zero_shift;
    NOaliParams::read(p);
    AliParams::read(p);
  }
  float alpha, beta, gamma;
  bool normalize_mtx;
  float zero_shift;
};
class Gn2Params;   // gn2_eval.h; Hmap2Eval takes one and uses only its HMAPaliParams part (hmap2_eval.h:25)

namespace aln {
// the shared body of HMAPaliEval and Hmap2Eval
template <class TSeq, class Derived>
class HmapEvalBase : public Evaluator<HMAPSequence, TSeq, Derived> {
 public:
  explicit HmapEvalBase(HMAPaliParams& p) : params(&p) {}
  float similarity(const HMAPSequence& q, const TSeq& t, int qi, int ti) const {
    float ip = 0.f;
    for (int k = 0; k < 20; ++k) { float p = q[qi]->aa_profile[k] * t[ti]->aa_profile[k]; ip += p; }
    float a[3], b[3];
    norm3(q[qi]->sse_values, a); norm3(t[ti]->sse_values, b);
    float pc = 0.f;
    for (int k = 0; k < 3; ++k) { float p = a[k] * b[k]; pc += p; }
    pc = pc / 3.f;
    return ip * expf(params->alpha * pc * q[qi]->sse_confid * t[ti]->sse_confid);
  }
  float deletion(const HMAPSequence&, const TSeq& t, int, int, int t1, int t2) const {
    int dist = t2 - t1;
    if (dist < 2) return 0.f;
    float gi = std::min(t[t1]->gap_init(), t[t2]->gap_init()), ge = std::min(t[t1]->gap_extn(), t[t2]->gap_extn());
    if (free_del() && (t[t1]->isHead() || t[t2]->isTail())) return 0.f;
    return gi + ge * (dist - 2);
  }
  float insertion(const HMAPSequence& q, const TSeq& t, int q1, int q2, int t1, int t2) const {
    int dist = q2 - q1;
    if (dist < 2) return 0.f;
    float gi = std::min(t[t1]->gap_init(), t[t2]->gap_init()), ge = std::min(t[t1]->gap_extn(), t[t2]->gap_extn());
    if (free_ins() && (q[q1]->isHead() || q[q2]->isTail())) return 0.f;
    return gi + ge * (dist - 2);
  }
  void pre_calculate(const HMAPSequence&, const TSeq& t) const {
    for (unsigned int i = 0; i < t.size(); ++i) {
      float Pi = expf(params->beta * (1.f - 1.25f * t[i]->p_coil()));
      t[i]->gap_init(params->gap_init_penalty * Pi);
      t[i]->gap_extn(params->gap_extn_penalty * Pi);
    }
  }
  void post_process(SimilarityMatrix&) const {}   // done on the device for the lowered path
  const HMAPaliParams* hmapParams() const { return params; }
 private:
  static void norm3(const std::valarray<float>& v, float out[3]) {
    float sum = 0.f; sum += v[0]; sum += v[1]; sum += v[2];
    float sq = 0.f; { float s = v[0] * v[0]; sq += s; } { float s = v[1] * v[1]; sq += s; } { float s = v[2] * v[2]; sq += s; }
    float avg = sum / 3.f, var = sq / 3.f - avg * avg, sd = std::sqrt(var);
    for (int k = 0; k < 3; ++k) { float x = v[k]; x -= avg; x /= sd; out[k] = x; }
  }
  void check() const { if (params->align_type < 0 || params->align_type > 4) throw std::string("Illegal gap style"); }
  bool free_del() const { check(); return params->align_type == local || params->align_type == semi_local || params->align_type == local_global; }
  bool free_ins() const { check(); return params->align_type == local || params->align_type == semi_local || params->align_type == global_local; }
  HMAPaliParams* params;
};

template <class TSeq, class E>
void lower_hmap(const HMAPSequence& q, const TSeq& t, const E& e, Lowered& L) {
  const HMAPaliParams* p = e.hmapParams();
  if (p->align_type < 0 || p->align_type > 4) throw std::string("Illegal gap style");
  auto pack = [](const Sequence<HMAPElem*>& s, std::vector<float>& aa, std::vector<float>& sse, std::vector<float>& conf) {
    aa.resize(s.size() * 20); sse.resize(s.size() * 3); conf.resize(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
      for (int k = 0; k < 20; ++k) aa[i * 20 + k] = s[i]->aa_profile[k];
      for (int k = 0; k < 3; ++k) sse[i * 3 + k] = s[i]->sse_values[k];
      conf[i] = s[i]->sse_confid;
    }
  };
  pack(q, L.q_aa, L.q_sse, L.q_conf);
  pack(t, L.t_aa, L.t_sse, L.t_conf);
  L.sim.kind = ALN_SIM_HMAP2;
  L.sim.q_prof.aa = L.q_aa.data(); L.sim.q_prof.sse = L.q_sse.data(); L.sim.q_prof.conf = L.q_conf.data();
  L.sim.t_prof.aa = L.t_aa.data(); L.sim.t_prof.sse = L.t_sse.data(); L.sim.t_prof.conf = L.t_conf.data();
  L.sim.alpha = p->alpha;
  L.sim.zero_shift = p->zero_shift;
  L.sim.normalize = 1;
  L.gd.model = ALN_GAP_AFFINE_TPOS_MIN;
  L.gd.align_type = p->align_type;
  L.gd.t_gap_init.resize(t.size()); L.gd.t_gap_extn.resize(t.size());
  for (size_t j = 0; j < t.size(); ++j) { L.gd.t_gap_init[j] = t[j]->gap_init(); L.gd.t_gap_extn[j] = t[j]->gap_extn(); }   // pre_calculate ran already
  L.finish_gap();
}
}  // namespace aln

class HMAPaliEval : public aln::HmapEvalBase<HMAPSequence, HMAPaliEval> {
 public:
  explicit HMAPaliEval(HMAPaliParams& p) : aln::HmapEvalBase<HMAPSequence, HMAPaliEval>(p) {}
};
class Hmap2Eval : public aln::HmapEvalBase<SMAPSequence, Hmap2Eval> {
 public:
  explicit Hmap2Eval(HMAPaliParams& p) : aln::HmapEvalBase<SMAPSequence, Hmap2Eval>(p) {}   // a Gn2Params binds here (derived class)
};

namespace aln {
template <>
struct Lowering<HMAPSequence, HMAPSequence, HMAPaliEval> {
  static void lower(const HMAPSequence& q, const HMAPSequence& t, const HMAPaliEval& e, Lowered& L) { lower_hmap(q, t, e, L); }
};
template <>
struct Lowering<HMAPSequence, SMAPSequence, Hmap2Eval> {
  static void lower(const HMAPSequence& q, const SMAPSequence& t, const Hmap2Eval& e, Lowered& L) { lower_hmap(q, t, e, L); }
};
}  // namespace aln
#endif
