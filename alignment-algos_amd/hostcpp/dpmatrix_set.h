// dpmatrix_set.h — DPMatrixSet<S1,S2,Etype>: MANY DPMatrix builds in one launch (an extension; the reference builds one matrix
// per constructor call, dpmatrix.h:147-165, and its drivers loop over pairs).
//
// A DPMatrix of this library owns a one-pair aln_batch: a loop over pairs is a loop of launches and of device allocations, and the
// GPU sees one 2-wave workgroup at a time.  DPMatrixSet takes the pairs the loop would visit and gives the same per-pair surface —
//     set.getCell(p, i, j), set.getSim(p, i, j), set.optimal(p) (what `AlignmentSet as(dpm, Optimal); as[0]` holds),
//     set.enumerate(p, noa, flags) (ConstrainedNearOptimal / UnconstrainedNearOptimal on pair p's resident matrix),
//     set.scores() (Optimal's score of every pair)
// — from ONE resident batch: one aln_batch_dp, one aln_batch_optimal for all pairs.  Evaluator families:
//   * AASubstitutionEval (codes + table + constant affine gaps): lowered once, nothing per pair on the host;
//   * Hmap2Eval / HMAPaliEval (profiles + position-minimum gaps): the pairs' profile records and gap coefficients are pooled in
//     pair order, similarity + z-normalisation run on the device for all pairs — what a loop of nalign2-style builds becomes;
//   * any other evaluator: similarity()/post_process() evaluated per pair on the host into planes (as DPMatrix does) and the
//     gap model it names itself (aln_describe_gaps, constant affine).  Evaluators whose gap functions must be tabulated
//     (Gn2Eval's tables, plain plugins) are built pair by pair with DPMatrix (their lowering is per pair by nature).
// Ownership as in DPMatrix: sequences and evaluator are borrowed, the matrices are the set's.
#ifndef ALN_HOST_DPMATRIX_SET_H
#define ALN_HOST_DPMATRIX_SET_H
#include <string>
#include <vector>

#include "alignment.h"
#include "aln_lowering.h"
#include "dpmatrix.h"
#include "noalib.h"
#include "sflags.h"

template <class S1, class S2, class Etype>
class DPMatrixSet {
 public:
  DPMatrixSet(const std::vector<const S1*>& queries, const std::vector<const S2*>& templates, const Evaluator<S1, S2, Etype>& eval,
              direction_t dir = fwd, align_t type = global)
      : qs(queries), ts(templates), evaluator(&eval), direction(dir), islocal(type == local), batch_(0), have_opt(false) {
    if (qs.size() != ts.size()) throw std::string("DPMatrixSet: one template per query");
    const size_t n = qs.size();
    // sequence pools: every pair brings its own two sequences (the same object may appear in many pairs)
    std::vector<int64_t> qo(n + 1, 0), to(n + 1, 0);
    std::string qres, tres;
    for (size_t p = 0; p < n; ++p) {
      qres += *qs[p]->getString(); tres += *ts[p]->getString();
      qo[p + 1] = (int64_t)qres.size(); to[p + 1] = (int64_t)tres.size();
    }
    aln_seqs qp = {(int32_t)n, qo.data(), qres.c_str()}, tp = {(int32_t)n, to.data(), tres.c_str()};
    std::vector<int32_t> idx(n);
    for (size_t p = 0; p < n; ++p) idx[p] = (int32_t)p;
    aln::check(aln_batch_create(aln::default_ctx(), &qp, &tp, (int32_t)n, idx.data(), idx.data(), 0, &batch_), aln::default_ctx());
    try { build(); } catch (...) { aln_batch_destroy(batch_); batch_ = 0; throw; }
  }
  ~DPMatrixSet() { if (batch_) aln_batch_destroy(batch_); }

  size_t size() const { return qs.size(); }
  aln_batch* batch() const { return batch_; }
  int getQuerySize(size_t p) const { return (int)qs[p]->size(); }
  int getTemplateSize(size_t p) const { return (int)ts[p]->size(); }
  void reevaluate() { cells.clear(); have_opt = false; build(); }

  // DPMatrix::getCell of pair p (downloaded the first time a pair is asked for)
  DPCell getCell(size_t p, int qpos, int tpos) const {
    const PairCells& c = fetch(p);
    DPCell d;
    const size_t e = (size_t)qpos * getTemplateSize(p) + tpos;
    d.query_idx = qpos; d.template_idx = tpos;
    d.setTB(c.pq[e], c.pt[e], c.sc[e]);
    return d;
  }
  float getSim(size_t p, int i, int j) const {
    std::vector<float> s((size_t)getQuerySize(p) * getTemplateSize(p));
    aln::check(aln_batch_get_sim(batch_, (int32_t)p, s.data()), aln::default_ctx());
    return s[(size_t)i * getTemplateSize(p) + j];
  }
  // Optimal (optimal.h:48-124) for every pair, one launch; score of pair p / its alignment
  const std::vector<float>& scores() const { run_optimal(); return opt_score; }
  typedef AlignedPairList<S1, S2> Alignment;
  Alignment optimal(size_t p) const {
    run_optimal();
    if (opt_status[p] != 0) aln::check(opt_status[p]);
    Alignment l;
    const int32_t* src = opt_pairs.data() + (size_t)p * opt_stride * 2;
    for (int k = 0; k < opt_n[p]; ++k) l.append(src[2 * k], src[2 * k + 1]);
    l.score = opt_score[p];
    return l;
  }
  // ConstrainedNearOptimal / UnconstrainedNearOptimal on pair p's resident matrix, seeded with its Optimal alignment as the
  // drivers do (aa_ali.cpp:83-89): the sorted set
  std::vector<Alignment> enumerate(size_t p, const NOaliParams& params, const SuboptFlags* sflags, bool constrained = true) const {
    aln_noa noa = aln_noa();
    noa.kind = constrained ? ALN_ENUM_CW : ALN_ENUM_UCW;
    noa.number_suboptimal = params.number_suboptimal;
    noa.delta_ratio = params.delta_ratio;
    noa.n_existing = -1;
    if (constrained && !sflags) throw std::string("DPMatrixSet::enumerate: ConstrainedNearOptimal needs SuboptFlags");
    if (sflags && (int)sflags->size() != getTemplateSize(p)) throw std::string("SuboptFlags length differs from the template");
    int cap = params.number_suboptimal + 2;
    const int per = getQuerySize(p) + getTemplateSize(p);
    for (;;) {
      std::vector<aln_alignment> out(cap);
      std::vector<int32_t> pairs((size_t)cap * per * 2);
      int32_t n_out = 0;
      const int rc = aln_batch_enumerate(batch_, (int32_t)p, &noa, sflags ? sflags->data() : 0, out.data(), cap, pairs.data(), (int64_t)cap * per, &n_out);
      if (rc == ALN_E_OVERFLOW && n_out > cap) { cap = n_out; continue; }
      aln::check(rc, aln::default_ctx());
      std::vector<Alignment> set(n_out);
      for (int k = 0; k < n_out; ++k) {
        const int32_t* src = pairs.data() + 2 * out[k].pair_off;
        for (int i = 0; i < out[k].n_pairs; ++i) set[k].append(src[2 * i], src[2 * i + 1]);
        set[k].score = out[k].score; set[k].identity = out[k].identity; set[k].uid = out[k].uid;
      }
      return set;
    }
  }

 private:
  struct PairCells { std::vector<float> sc; std::vector<int32_t> pq, pt; };

  void build() {
    const size_t n = qs.size();
    aln::Lowered L;
    std::vector<int64_t> plane_off;
    std::vector<float> planes;
    if (n == 0) return;
    // pre_calculate belongs to ONE pair: the reference calls it right before building that pair (dpmatrix.h:298), so an
    // evaluator may keep per-pair state in itself.  Every lowering below is therefore preceded by its own pair's call.
    evaluator->pre_calculate(*qs[0], *ts[0]);
    aln::Lowering<S1, S2, Etype>::lower(*qs[0], *ts[0], evaluator->Derived(), L);
    std::vector<float> q_aa, q_sse, q_conf, t_aa, t_sse, t_conf, t_gi, t_ge;    // profile pools (Hmap2Eval / HMAPaliEval)
    if (L.sim.kind == ALN_SIM_HMAP2 && L.gap.model == ALN_GAP_AFFINE_TPOS_MIN) {
      // profile evaluators: per-position records and per-template-position gap coefficients, pooled in pair order like the residues
      auto app = [](std::vector<float>& dst, const std::vector<float>& src) { dst.insert(dst.end(), src.begin(), src.end()); };
      for (size_t p = 0; p < n; ++p) {
        aln::Lowered Lp;
        if (p) {
          evaluator->pre_calculate(*qs[p], *ts[p]);
          aln::Lowering<S1, S2, Etype>::lower(*qs[p], *ts[p], evaluator->Derived(), Lp);
        }
        const aln::Lowered& X = p ? Lp : L;
        if (X.sim.kind != ALN_SIM_HMAP2 || X.sim.alpha != L.sim.alpha || X.sim.zero_shift != L.sim.zero_shift || X.gap.align_type != L.gap.align_type)
          throw std::string("DPMatrixSet: the evaluator's parameters differ from pair to pair; build DPMatrix objects");
        app(q_aa, X.q_aa); app(q_sse, X.q_sse); app(q_conf, X.q_conf);
        app(t_aa, X.t_aa); app(t_sse, X.t_sse); app(t_conf, X.t_conf);
        app(t_gi, X.gd.t_gap_init); app(t_ge, X.gd.t_gap_extn);
      }
      L.sim.q_prof.aa = q_aa.data(); L.sim.q_prof.sse = q_sse.data(); L.sim.q_prof.conf = q_conf.data();
      L.sim.t_prof.aa = t_aa.data(); L.sim.t_prof.sse = t_sse.data(); L.sim.t_prof.conf = t_conf.data();
      L.gap.t_gap_init = t_gi.data(); L.gap.t_gap_extn = t_ge.data();
      L.gap.dp_local = islocal ? 2 : 1;
      aln::check(aln_batch_dp(batch_, &L.sim, &L.gap, (int)direction, ALN_DP_AUTO, 0), aln::default_ctx());
      return;
    }
    if (L.gap.model != ALN_GAP_AFFINE_CONST)
      throw std::string("DPMatrixSet: this evaluator's gap functions are lowered per pair; build DPMatrix objects");
    if (L.sim.kind == ALN_SIM_MATRIX) {                     // a plane per pair, as DPMatrix::build makes them one at a time
      plane_off.assign(n, 0);
      planes = L.plane;
      for (size_t p = 1; p < n; ++p) {
        aln::Lowered Lp;
        evaluator->pre_calculate(*qs[p], *ts[p]);
        aln::Lowering<S1, S2, Etype>::lower(*qs[p], *ts[p], evaluator->Derived(), Lp);
        if (Lp.gap.model != ALN_GAP_AFFINE_CONST || Lp.gap.gap_init != L.gap.gap_init || Lp.gap.gap_extn != L.gap.gap_extn ||
            Lp.gap.align_type != L.gap.align_type)
          throw std::string("DPMatrixSet: the evaluator's gap model differs from pair to pair; build DPMatrix objects");
        plane_off[p] = (int64_t)planes.size();
        planes.insert(planes.end(), Lp.plane.begin(), Lp.plane.end());
      }
      L.sim.planes = planes.data();
      L.sim.plane_off = plane_off.data();
    } else if (L.sim.kind != ALN_SIM_SUBMATRIX) {
      throw std::string("DPMatrixSet: this evaluator's similarity source is lowered per pair; build DPMatrix objects");
    } else {
      for (size_t p = 1; p < n; ++p) evaluator->pre_calculate(*qs[p], *ts[p]);   // one table for all pairs: the hook still runs once per pair
    }
    L.gap.dp_local = islocal ? 2 : 1;                       // the constructor's `type` decides the clipping (dpmatrix.h:155)
    aln::check(aln_batch_dp(batch_, &L.sim, &L.gap, (int)direction, ALN_DP_AUTO, 0), aln::default_ctx());
  }
  const PairCells& fetch(size_t p) const {
    if (cells.size() != qs.size()) cells.assign(qs.size(), PairCells());
    PairCells& c = cells[p];
    if (c.sc.empty()) {
      const size_t e = (size_t)getQuerySize(p) * getTemplateSize(p);
      c.sc.resize(e); c.pq.resize(e); c.pt.resize(e);
      aln::check(aln_batch_get_cells(batch_, (int32_t)p, c.sc.data(), c.pq.data(), c.pt.data()), aln::default_ctx());
    }
    return c;
  }
  void run_optimal() const {
    if (have_opt) return;
    const size_t n = qs.size();
    int mx = 2;
    for (size_t p = 0; p < n; ++p) mx = std::max(mx, std::min(getQuerySize(p), getTemplateSize(p)) + 2);
    opt_stride = mx;
    opt_score.assign(n, 0.f); opt_n.assign(n, 0); opt_status.assign(n, 0);
    opt_pairs.assign(n * (size_t)opt_stride * 2, 0);
    aln::check(aln_batch_optimal(batch_, opt_score.data(), opt_n.data(), opt_pairs.data(), opt_stride, opt_status.data()), aln::default_ctx());
    have_opt = true;
  }

  std::vector<const S1*> qs;
  std::vector<const S2*> ts;
  const Evaluator<S1, S2, Etype>* evaluator;
  direction_t direction;
  bool islocal;
  aln_batch* batch_;
  mutable std::vector<PairCells> cells;
  mutable bool have_opt;
  mutable int opt_stride;
  mutable std::vector<float> opt_score;
  mutable std::vector<int32_t> opt_n, opt_status, opt_pairs;

  DPMatrixSet(const DPMatrixSet&);
  DPMatrixSet& operator=(const DPMatrixSet&);
};
#endif
