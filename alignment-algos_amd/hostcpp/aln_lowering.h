// aln_lowering.h — how a host-side Evaluator becomes the plain arrays the C ABI takes (include/aln_hip.h).
//
// The reference calls the evaluator's similarity/deletion/insertion per DP cell (dpmatrix.h:447-486).  Here
// DPMatrix::build() runs pre_calculate() once, then asks aln::Lowering<S1,S2,Etype> for
//   * a similarity source:  residue codes + substitution table | a materialised SimilarityMatrix | HMAP profiles
//   * a gap model from the closed set of include/aln_hip.h (AFFINE_CONST, AFFINE_TPOS_MIN, DEL_TABLE_INS_TPOS)
// and hands both to aln_batch_dp().  The evaluator families of the reference (AASubstitutionEval, Hmap2Eval /
// HMAPaliEval, Gn2Eval) have specialisations below / in hmap_eval.h / gn2_eval.h.  Any other evaluator works through the generic
// path if it says which gap model its deletion()/insertion() implement:
//
//     void aln_describe_gaps(const S1& q, const S2& t, aln::GapDescription& g) const;
//
// (its similarity() + post_process() are evaluated once per cell on the host into a SimilarityMatrix plane).
#ifndef ALN_HOST_LOWERING_H
#define ALN_HOST_LOWERING_H
#include <string>
#include <vector>

#include "aasubalib.h"
#include "aln_hip.h"
#include "alib.h"
#include "evaluator.h"
#include "simmatrix.h"

namespace aln {

struct GapDescription {
  int model;                    // ALN_GAP_AFFINE_CONST or ALN_GAP_AFFINE_TPOS_MIN
  int align_type;               // the align_t the evaluator's free-end rules follow
  float gap_init, gap_extn;     // AFFINE_CONST
  std::vector<float> t_gap_init, t_gap_extn;   // AFFINE_TPOS_MIN, one per template position (sentinels included)
  GapDescription() : model(ALN_GAP_AFFINE_CONST), align_type(semi_local), gap_init(0.f), gap_extn(0.f) {}
};

// Everything aln_batch_dp needs, with the storage that backs the pointers.
struct Lowered {
  aln_sim sim;
  aln_gap gap;
  GapDescription gd;
  std::string alphabet;
  std::vector<float> table;
  std::vector<float> plane;
  int64_t plane_off0;
  std::vector<float> q_aa, q_sse, q_conf, t_aa, t_sse, t_conf;
  std::vector<float> t_gap_cn, del_table;     // ALN_GAP_DEL_TABLE_INS_TPOS (Gn2Eval)
  int64_t del_off0;
  Lowered() : plane_off0(0), del_off0(0) { sim = aln_sim(); gap = aln_gap(); }
  void finish_gap() {
    gap.model = gd.model; gap.align_type = gd.align_type; gap.gap_init = gd.gap_init; gap.gap_extn = gd.gap_extn;
    gap.t_gap_init = gd.t_gap_init.empty() ? 0 : gd.t_gap_init.data();
    gap.t_gap_extn = gd.t_gap_extn.empty() ? 0 : gd.t_gap_extn.data();
  }
};

// generic path: SimilarityMatrix on the host + the evaluator's own gap description
template <class S1, class S2, class Etype>
struct Lowering {
  static void lower(const S1& q, const S2& t, const Etype& e, Lowered& L) {
    SimilarityMatrix sm(q, t, static_cast<const Evaluator<S1, S2, Etype>&>(e));
    L.plane.assign(sm.data(), sm.data() + sm.size());
    L.sim.kind = ALN_SIM_MATRIX;
    L.sim.planes = L.plane.data();
    L.plane_off0 = 0;
    L.sim.plane_off = &L.plane_off0;
    e.aln_describe_gaps(q, t, L.gd);     // a user evaluator must provide this (see the header comment)
    L.finish_gap();
  }
};

// AASubstitutionEval: codes + table, constant affine gaps (aasubalib.h:17-77)
template <class S1, class S2>
struct Lowering<S1, S2, AASubstitutionEval<S1, S2> > {
  static void lower(const S1&, const S2&, const AASubstitutionEval<S1, S2>& e, Lowered& L) {
    const SubstitutionMatrix* m = e.subMatrix();
    L.alphabet = m->letters();
    const size_t n = L.alphabet.size();
    L.table.assign(m->table(), m->table() + n * n);
    L.sim.kind = ALN_SIM_SUBMATRIX;
    L.sim.sub.n = (int32_t)n;
    L.sim.sub.alphabet = L.alphabet.c_str();
    L.sim.sub.table = L.table.data();
    const AliParams* p = e.aliParams();
    if (p->align_type < 0 || p->align_type > 4) throw std::string("Illegal gap style");
    L.gd.model = ALN_GAP_AFFINE_CONST;
    L.gd.align_type = p->align_type;
    L.gd.gap_init = p->gap_init_penalty;
    L.gd.gap_extn = p->gap_extn_penalty;
    L.finish_gap();
  }
};

inline void check(int rc, aln_ctx* ctx = 0) {
  if (rc == ALN_OK) return;
  std::string msg = aln_error_string(rc);
  if (rc == ALN_E_HIP && ctx) msg += std::string(": ") + aln_last_error(ctx);
  throw msg;                   // the reference throws std::string everywhere (dpmatrix.h:361, optimal.h:74, ...)
}
// the process-wide context the host classes use (device 0 unless ALN_DEVICE is set)
inline aln_ctx* default_ctx() {
  static aln_ctx* ctx = 0;
  if (!ctx) {
    const char* d = getenv("ALN_DEVICE");
    check(aln_ctx_create(d ? atoi(d) : 0, 0, &ctx));
  }
  return ctx;
}

}  // namespace aln
#endif
