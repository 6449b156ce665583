// gn2_eval.h — Gn2Params and Gn2Eval, the structure-aware evaluator (reference gn2_eval.h:19-169, gn2_eval.cpp:16-158).
//   similarity = shift + w_aa*log_aa(norm. dot of the aa profiles) + w_ss*ss_lods + w_cn*log_cn + w_hp*log_hp   (gn2_eval.h:71-98)
//   deletion(t1,t2) = 8100 beyond 18 A, else vv_gi + vv_ge*(di-2) + vv_cd from (t2-2, t1)-indexed tables      (:100-130)
//   insertion       = v_gi[t1] + v_ge[t1]*(di-2) + v_cn[t1]                                                     (:132-165)
//   pre_calculate fills those tables from the template's p_coil, contact numbers, SSE membership, distances and
//   broken H-bond counts (gn2_eval.cpp:113-158).
// On this engine it lowers to a host-materialised SimilarityMatrix plane + ALN_GAP_DEL_TABLE_INS_TPOS: the deletion table
// is this class's own deletion() evaluated for every t1 < t2, so the device sees exactly the values the reference's loops
// would have computed.  The reference cannot be built here (SMAPSequence needs the Troll library), so this file is checked
// against the oracle's restatement only: parity UNPINNED.
#ifndef ALN_HOST_GN2_EVAL_H
#define ALN_HOST_GN2_EVAL_H
#include <cmath>
#include <string>
#include <vector>
#include "aln_lowering.h"
#include "hmap_eval.h"

class Gn2Params : public HMAPaliParams {
 public:
  Gn2Params()
      : ss_lods(36), gap_init_coil(1.2f), gap_extn_coil(0.08f), gap_init_ss(100.f), gap_extn_ss(1.f), aa_weight(1.00f),
        ss_weight(2.2f), cn_weight(3.4f), hp_weight(1.2f), hb_weight(0.13f), ic_weight(0.09f), dd_constr(8.f), gn2_shift(1.2f),
        ss_dependent_gp(true) {
    static const float lods[36] = {0.08f, 0.22f, 0.43f, -1.05f, -1.20f, -1.57f, -0.30f, -0.50f, -0.55f, 0.f, 0.f, 0.f,
                                   -0.56f, -0.79f, -1.70f, 0.32f, 0.44f, 0.60f, -0.13f, -0.22f, -0.49f, 0.f, 0.f, 0.f,
                                   -0.04f, -0.18f, -0.59f, 0.10f, 0.01f, -0.33f, 0.14f, 0.18f, 0.28f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 36; ++k) ss_lods[k] = lods[k];   // helix / strand rows carry a weight of 1 (gn2_eval.cpp:47-48)
  }
  void read(ParamStore* p) {
    std::string s;
    s = "GI_COIL"; if (p->find(s)) p->getValue(s) >> gap_init_coil;
    s = "GE_COIL"; if (p->find(s)) p->getValue(s) >> gap_extn_coil;
    s = "GI_SS"; if (p->find(s)) p->getValue(s) >> gap_init_ss;
    s = "GE_SS"; if (p->find(s)) p->getValue(s) >> gap_extn_ss;
    s = "AA_WEIGHT"; if (p->find(s)) p->getValue(s) >> aa_weight;
    s = "SS_WEIGHT"; if (p->find(s)) p->getValue(s) >> ss_weight;
    s = "CN_WEIGHT"; if (p->find(s)) p->getValue(s) >> cn_weight;
    s = "HP_WEIGHT"; if (p->find(s)) p->getValue(s) >> hp_weight;
    s = "HB_WEIGHT"; if (p->find(s)) p->getValue(s) >> hb_weight;
    s = "IC_WEIGHT"; if (p->find(s)) p->getValue(s) >> ic_weight;
    s = "GN2_SHIFT"; if (p->find(s)) p->getValue(s) >> gn2_shift;
    s = "DEL_DIST_CONSTR"; if (p->find(s)) p->getValue(s) >> dd_constr;
    s = "SS_DEPENDENT_GP"; if (p->find(s)) p->getValue(s) >> ss_dependent_gp;
    HMAPaliParams::read(p);
  }
  std::vector<float> ss_lods;
  float gap_init_coil, gap_extn_coil, gap_init_ss, gap_extn_ss;
  float aa_weight, ss_weight, cn_weight, hp_weight, hb_weight, ic_weight, dd_constr, gn2_shift;
  bool ss_dependent_gp;
};

class Gn2Eval : public Evaluator<HMAPSequence, SMAPSequence, Gn2Eval> {
 public:
  explicit Gn2Eval(Gn2Params& p) : params(&p) {}

  float similarity(const HMAPSequence& q, const SMAPSequence& t, int q_pos, int t_pos) const {
    float ip = norm_dot(q[q_pos]->aa_profile, t[t_pos]->aa_profile);
    unsigned int lods_idx = t[t_pos]->lods_type * 12 + q[q_pos]->lods_type;
    float log_aa = 0.543f / (2.85f - std::exp(ip)) - 0.738f;
    float log_ss = params->ss_lods[lods_idx];
    float log_cn = 2.f * t.weighted_contact_number[t_pos] - 0.9f;
    float log_hp = std::exp(std::exp(-std::abs(q[q_pos]->hydropathy - t[t_pos]->hydropathy)) *
                            (0.75f + 0.3f * std::abs(t[t_pos]->hydropathy - 0.22f))) - 1.8f;
    float sim = params->gn2_shift + params->aa_weight * log_aa + params->ss_weight * log_ss + params->cn_weight * log_cn +
                params->hp_weight * log_hp;
    return sim;
  }
  float deletion(const HMAPSequence&, const SMAPSequence& t, int, int, int t_pos1, int t_pos2) const {
    int di = t_pos2 - t_pos1;
    if (di < 2) return 0;
    int p1 = t_pos1, p2 = t_pos2 - 2;
    float GP = 8100.f;
    if (t.distance[p2][p1] < 18.f) GP = vv_gi[p2][p1] + vv_ge[p2][p1] * (di - 2) + vv_cd[p2][p1];
    switch (params->align_type) {
      case global: case global_local: return GP;
      case local: case semi_local: case local_global:
        if (t[t_pos1]->isHead() || t[t_pos2]->isTail()) return 0;
        return GP;
      default: throw std::string("Invalid align_type");
    }
  }
  float insertion(const HMAPSequence& q, const SMAPSequence&, int q_pos1, int q_pos2, int t_pos1, int) const {
    int di = q_pos2 - q_pos1;
    if (di < 2) return 0;
    float GP = v_gi[t_pos1] + v_ge[t_pos1] * (di - 2) + v_cn[t_pos1];
    switch (params->align_type) {
      case global: case local_global: return GP;
      case local: case semi_local: case global_local:
        if (q[q_pos1]->isHead() || q[q_pos2]->isTail()) return 0;
        return GP;
      default: throw std::string("Invalid align_type");
    }
  }
  void pre_calculate(const HMAPSequence&, const SMAPSequence& templ) const {
    const unsigned int n = templ.seq_length;
    v_gi.resize(n + 1); v_ge.resize(n + 1); v_cn.resize(n + 1);
    for (unsigned int i = 0; i <= n; ++i) {
      float v_coil = std::max(templ[i]->p_coil(), templ[i + 1]->p_coil());
      v_gi[i] = v_coil * params->gap_init_coil + (1 - v_coil) * params->gap_init_ss;
      v_ge[i] = v_coil * params->gap_extn_coil + (1 - v_coil) * params->gap_extn_ss;
      float cn = templ.weighted_contact_number[i] + templ.weighted_contact_number[i + 1];
      v_cn[i] = params->ic_weight * (1.693f - std::log(cn));
    }
    vv_gi.resize(n); vv_ge.resize(n); vv_cd.resize(n);
    for (unsigned int i = 2; i < n + 2; ++i) {
      vv_gi[i - 2].resize(i - 1); vv_ge[i - 2].resize(i - 1); vv_cd[i - 2].resize(i - 1);
      for (unsigned int j = 0; j < i - 1; ++j) {
        float v_allow = 1;
        if (templ[i]->rdata.isse == templ[j]->rdata.isse && templ[i]->rdata.isse > -1) v_allow = 0;
        vv_gi[i - 2][j] = v_allow * params->gap_init_coil + (1.f - v_allow) * params->gap_init_ss;
        vv_ge[i - 2][j] = v_allow * params->gap_extn_coil + (1.f - v_allow) * params->gap_extn_ss;
        vv_cd[i - 2][j] = std::exp(templ.distance[i - 2][j] - params->dd_constr);
        vv_cd[i - 2][j] += v_allow * params->hb_weight * templ.brokenhb[i - 2][j];
      }
    }
  }
  void post_process(SimilarityMatrix&) const {}
  const Gn2Params* gn2Params() const { return params; }
  // the tables pre_calculate() filled (for the lowering and for tests)
  mutable std::vector<float> v_gi, v_ge, v_cn;
  mutable std::vector<std::vector<float> > vv_gi, vv_ge, vv_cd;

 private:
  static float norm_dot(const std::valarray<float>& a, const std::valarray<float>& b) {   // hmath.h:27-40
    float res = 0.f, sa = 0.f, sb = 0.f;
    for (size_t k = 0; k < a.size(); ++k) { float p = a[k] * b[k]; res += p; }
    for (size_t k = 0; k < a.size(); ++k) { float p = a[k] * a[k]; sa += p; }
    for (size_t k = 0; k < b.size(); ++k) { float p = b[k] * b[k]; sb += p; }
    float norm = std::sqrt(sa) * std::sqrt(sb);
    return res / norm;
  }
  Gn2Params* params;
};

namespace aln {
// Gn2Eval: similarity plane on the host (needs exp/log of structural terms once per cell), gaps as tables.
template <>
struct Lowering<HMAPSequence, SMAPSequence, Gn2Eval> {
  static void lower(const HMAPSequence& q, const SMAPSequence& t, const Gn2Eval& e, Lowered& L) {
    const Gn2Params* p = e.gn2Params();
    if (p->align_type < 0 || p->align_type > 4) throw std::string("Invalid align_type");
    SimilarityMatrix sm(q, t, static_cast<const Evaluator<HMAPSequence, SMAPSequence, Gn2Eval>&>(e));
    L.plane.assign(sm.data(), sm.data() + sm.size());
    L.sim.kind = ALN_SIM_MATRIX;
    L.sim.planes = L.plane.data();
    L.plane_off0 = 0;
    L.sim.plane_off = &L.plane_off0;
    const size_t T = t.size();
    L.gd.model = ALN_GAP_DEL_TABLE_INS_TPOS;
    L.gd.align_type = p->align_type;
    L.gd.t_gap_init.assign(T, 0.f); L.gd.t_gap_extn.assign(T, 0.f); L.t_gap_cn.assign(T, 0.f);
    for (size_t j = 0; j < e.v_gi.size() && j < T; ++j) { L.gd.t_gap_init[j] = e.v_gi[j]; L.gd.t_gap_extn[j] = e.v_ge[j]; L.t_gap_cn[j] = e.v_cn[j]; }
    L.del_table.assign(T * T, 0.f);
    for (size_t t1 = 0; t1 < T; ++t1)
      for (size_t t2 = t1 + 1; t2 < T; ++t2) L.del_table[t1 * T + t2] = e.deletion(q, t, 0, 0, (int)t1, (int)t2);
    L.del_off0 = 0;
    L.finish_gap();
    L.gap.t_gap_cn = L.t_gap_cn.data();
    L.gap.del_table = L.del_table.data();
    L.gap.del_table_off = &L.del_off0;
  }
};
}  // namespace aln
#endif
