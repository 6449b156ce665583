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

  // Score of one cell = shift + Σ weight·term over four log-odds-like terms, summed left to right in fp32 (gn2_eval.h:71-98):
  //   residues   0.543 / (2.85 − e^cos) − 0.738, cos = normalised dot of the two 20-vectors
  //   sec.struct a 12×12 table entry picked by the two positions' lods types
  //   burial     2·(template contact number) − 0.9
  //   hydropathy e^( e^(−|hq − ht|) · (0.75 + 0.3·|ht − 0.22|) ) − 1.8
  float similarity(const HMAPSequence& q, const SMAPSequence& t, int qi, int ti) const {
    const HMAPSequence::value_type& qe = q[qi];
    const SMAPSequence::value_type& te = t[ti];
    const float term_res = 0.543f / (2.85f - std::exp(norm_dot(qe->aa_profile, te->aa_profile))) - 0.738f;
    const float term_sse = params->ss_lods[te->lods_type * 12 + qe->lods_type];
    const float term_bur = 2.f * t.weighted_contact_number[ti] - 0.9f;
    const float term_hyd = std::exp(std::exp(-std::abs(qe->hydropathy - te->hydropathy)) * (0.75f + 0.3f * std::abs(te->hydropathy - 0.22f))) - 1.8f;
    float total = params->gn2_shift;
    total += params->aa_weight * term_res;
    total += params->ss_weight * term_sse;
    total += params->cn_weight * term_bur;
    total += params->hp_weight * term_hyd;
    return total;
  }
  // Skipping template positions t1+1..t2-1 (gn2_eval.h:100-131): tabulated per (t2-2, t1) — affine part + distance/H-bond part —
  // unless the two flanks are 18 Å or more apart (then a prohibitive 8100); free where the align type frees template ends.
  float deletion(const HMAPSequence&, const SMAPSequence& t, int, int, int t1, int t2) const {
    const int hop = t2 - t1;
    if (hop < 2) return 0;
    check_type();
    if (template_ends_free() && (t[t1]->isHead() || t[t2]->isTail())) return 0;
    const int row = t2 - 2;
    if (!(t.distance[row][t1] < 18.f)) return 8100.f;
    return vv_gi[row][t1] + vv_ge[row][t1] * (hop - 2) + vv_cd[row][t1];
  }
  // Skipping query positions q1+1..q2-1 after template position t1 (gn2_eval.h:133-158): affine part + burial part of t1
  float insertion(const HMAPSequence& q, const SMAPSequence&, int q1, int q2, int t1, int) const {
    const int hop = q2 - q1;
    if (hop < 2) return 0;
    check_type();
    if (query_ends_free() && (q[q1]->isHead() || q[q2]->isTail())) return 0;
    return v_gi[t1] + v_ge[t1] * (hop - 2) + v_cn[t1];
  }
  // Tables per template (gn2_eval.cpp:113-158).  Between positions i and i+1: gap constants blended by the larger coil
  // probability of the two, and ic_weight·(1.693 − ln(sum of their contact numbers)).  Per (i-2, j), j < i-1: coil constants
  // unless both flanks lie in the same secondary-structure element, e^(distance − dd_constr), plus hb_weight·broken H-bonds
  // where a gap is allowed.
  void pre_calculate(const HMAPSequence&, const SMAPSequence& templ) const {
    const unsigned int n = templ.seq_length;
    v_gi.assign(n + 1, 0.f); v_ge.assign(n + 1, 0.f); v_cn.assign(n + 1, 0.f);
    for (unsigned int i = 0; i <= n; ++i) {
      const float coil = std::max(templ[i]->p_coil(), templ[i + 1]->p_coil());
      v_gi[i] = blend(coil, params->gap_init_coil, params->gap_init_ss);
      v_ge[i] = blend(coil, params->gap_extn_coil, params->gap_extn_ss);
      v_cn[i] = params->ic_weight * (1.693f - std::log(templ.weighted_contact_number[i] + templ.weighted_contact_number[i + 1]));
    }
    vv_gi.assign(n, std::vector<float>()); vv_ge.assign(n, std::vector<float>()); vv_cd.assign(n, std::vector<float>());
    for (unsigned int row = 0; row < n; ++row) {
      const unsigned int i = row + 2;
      std::vector<float> &gi = vv_gi[row], &ge = vv_ge[row], &cd = vv_cd[row];
      gi.resize(i - 1); ge.resize(i - 1); cd.resize(i - 1);
      for (unsigned int j = 0; j + 1 < i; ++j) {
        const int elem = templ[i]->rdata.isse;
        const float open = (elem > -1 && elem == templ[j]->rdata.isse) ? 0.f : 1.f;
        gi[j] = blend(open, params->gap_init_coil, params->gap_init_ss);
        ge[j] = blend(open, params->gap_extn_coil, params->gap_extn_ss);
        cd[j] = std::exp(templ.distance[row][j] - params->dd_constr);
        cd[j] += open * params->hb_weight * templ.brokenhb[row][j];
      }
    }
  }
  void post_process(SimilarityMatrix&) const {}
  const Gn2Params* gn2Params() const { return params; }
  // the tables pre_calculate() filled (for the lowering and for tests)
  mutable std::vector<float> v_gi, v_ge, v_cn;
  mutable std::vector<std::vector<float> > vv_gi, vv_ge, vv_cd;

 private:
  static float blend(float w, float at_one, float at_zero) { return w * at_one + (1.f - w) * at_zero; }
  void check_type() const { if (params->align_type < 0 || params->align_type > 4) throw std::string("Invalid align_type"); }
  bool template_ends_free() const { return params->align_type == local || params->align_type == semi_local || params->align_type == local_global; }
  bool query_ends_free() const { return params->align_type == local || params->align_type == semi_local || params->align_type == global_local; }
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
