// dpmatrix.h — DPMatrix<S1,S2,Etype>: the reference's dynamic-programming matrix (dpmatrix.h:28-113) with the
// same constructors and accessors, backed by a device-resident aln_batch of one pair.
//
//   DPMatrix(q, t, eval, direction = fwd, type = global)             build() on construction   (reference :147-165)
//   DPMatrix(q, t, eval, q1_end, t1_end, q2_beg, t2_beg, dir, type)  build_subdpm()            (:169-189, B8 order)
//   getCell(i,j) -> const DPCell*   score + (prev_query_idx, prev_template_idx); valid until reevaluate()/dtor
//   getSim(i,j), getQuerySize(), getTemplateSize(), getQuerySequence(), getTemplateSequence(), getEvaluator(),
//   getDirection(), setEvaluator(eval, dir), reevaluate(), operator<<
// Ownership is the reference's: sequences and evaluator are borrowed (:220-227), matrices are owned.
// The build runs pre_calculate() on the host, lowers the evaluator (aln_lowering.h) and calls aln_batch_dp();
// cells are downloaded lazily the first time getCell()/getSim() is used.
#ifndef ALN_HOST_DPMATRIX_H
#define ALN_HOST_DPMATRIX_H
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "alib.h"
#include "aln_hip.h"
#include "aln_lowering.h"
#include "evaluator.h"
#include "simmatrix.h"

enum direction_t { fwd = 1, rev = 2 };

struct DPCell {
  int prev_query_idx;
  int prev_template_idx;
  int query_idx;
  int template_idx;
  float score;
  static const int null = -1;
  DPCell() : prev_query_idx(null), prev_template_idx(null), query_idx(null), template_idx(null), score(0.0f) {}
  void setTB(int pq, int pt, float s) { prev_query_idx = pq; prev_template_idx = pt; score = s; }
};

template <class S1, class S2, class Etype>
class DPMatrix {
 public:
  DPMatrix(const S1& qs, const S2& ts, const Evaluator<S1, S2, Etype>& eval, direction_t dir = fwd, align_t type = global)
      : query_seq(&qs), templ_seq(&ts), evaluator(&eval), direction(dir), islocal(type == local), sub(false), batch_(0),
        have_cells(false), have_sim(false) {
    q0 = t0 = 0; q1 = (int)qs.size() - 1; t1 = (int)ts.size() - 1;
    create_batch();
    build();
  }
  // sub-rectangle build; argument order of the reference's DEFINITION and callers (dpmatrix.h:172-173)
  DPMatrix(const S1& qs, const S2& ts, const Evaluator<S1, S2, Etype>& eval, int q1_end, int t1_end, int q2_beg, int t2_beg,
           direction_t dir = fwd, align_t type = global)
      : query_seq(&qs), templ_seq(&ts), evaluator(&eval), direction(dir), islocal(type == local), sub(true), batch_(0),
        have_cells(false), have_sim(false) {
    q0 = q1_end; t0 = t1_end; q1 = q2_beg; t1 = t2_beg;
    create_batch();
    build();
  }
  ~DPMatrix() { if (batch_) aln_batch_destroy(batch_); }

  void setEvaluator(const Evaluator<S1, S2, Etype>& eval, direction_t dir) { direction = dir; evaluator = &eval; reevaluate(); }
  void reevaluate() { build(); }                       // resetMtxVal + build (dpmatrix.h:213-218)
  const DPCell* getCell(int qpos, int tpos) const { fetch_cells(); return &cells[(size_t)qpos * getTemplateSize() + tpos]; }
  direction_t getDirection() const { return direction; }
  int getQuerySize() const { return (int)query_seq->size(); }
  int getTemplateSize() const { return (int)templ_seq->size(); }
  const Evaluator<S1, S2, Etype>* getEvaluator() const { return evaluator; }
  const S1* getQuerySequence() const { return query_seq; }
  const S2* getTemplateSequence() const { return templ_seq; }
  float getSim(int i, int j) const { fetch_sim(); return simv[(size_t)i * getTemplateSize() + j]; }
  // engine handles for the enumerators
  aln_batch* batch() const { return batch_; }
  bool isLocal() const { return islocal; }
  bool isSub() const { return sub; }

 protected:
  void build_forw_dpm_linear_gaps() { throw std::string("Under construction!~Please use nonlinear gap algorithm"); }   // dpmatrix.h:1032-1042
  void build_rev_dpm_linear_gaps() { throw std::string("Under construction!~Please use nonlinear gap algorithm"); }

  void create_batch() {
    const std::string& q = *query_seq->getString();
    const std::string& t = *templ_seq->getString();
    int64_t qo[2] = {0, (int64_t)q.size()}, to[2] = {0, (int64_t)t.size()};
    aln_seqs qs = {1, qo, q.c_str()}, ts = {1, to, t.c_str()};
    int32_t zero = 0;
    aln::check(aln_batch_create(aln::default_ctx(), &qs, &ts, 1, &zero, &zero, 0, &batch_), aln::default_ctx());
  }
  void build() {
    have_cells = have_sim = false;
    evaluator->pre_calculate(*query_seq, *templ_seq);                         // dpmatrix.h:298
    aln::Lowered L;
    aln::Lowering<S1, S2, Etype>::lower(*query_seq, *templ_seq, evaluator->Derived(), L);
    // the constructor's own `type` decides the clipping (dpmatrix.h:155), the evaluator's align_type the end gaps
    L.gap.dp_local = islocal ? 2 : 1;
    int rc;
    if (sub) {
      int32_t bounds[4] = {q0, t0, q1, t1};
      rc = aln_batch_dp_sub(batch_, &L.sim, &L.gap, (int)direction, bounds);
    } else {
      const char* bug = getenv("ALN_REFERENCE_BUG_B4");                       // dpmatrix.h:868, off by default
      rc = aln_batch_dp(batch_, &L.sim, &L.gap, (int)direction, ALN_DP_AUTO, bug && bug[0] == '1');
    }
    aln::check(rc, aln::default_ctx());
  }
  void fetch_cells() const {
    if (have_cells) return;
    const int Q = getQuerySize(), T = getTemplateSize();
    std::vector<float> sc((size_t)Q * T);
    std::vector<int32_t> pq((size_t)Q * T), pt((size_t)Q * T);
    aln::check(aln_batch_get_cells(batch_, 0, sc.data(), pq.data(), pt.data()), aln::default_ctx());
    cells.resize((size_t)Q * T);
    for (int i = 0; i < Q; ++i)
      for (int j = 0; j < T; ++j) {
        DPCell& c = cells[(size_t)i * T + j];
        c.query_idx = i; c.template_idx = j;                                   // initMtxVal, dpmatrix.h:261-273
        c.setTB(pq[(size_t)i * T + j], pt[(size_t)i * T + j], sc[(size_t)i * T + j]);
      }
    have_cells = true;
  }
  void fetch_sim() const {
    if (have_sim) return;
    simv.resize((size_t)getQuerySize() * getTemplateSize());
    aln::check(aln_batch_get_sim(batch_, 0, simv.data()), aln::default_ctx());
    have_sim = true;
  }

  const S1* query_seq;
  const S2* templ_seq;
  const Evaluator<S1, S2, Etype>* evaluator;
  direction_t direction;
  bool islocal;
  bool sub;
  int q0, q1, t0, t1;
  aln_batch* batch_;
  mutable std::vector<DPCell> cells;
  mutable std::vector<float> simv;
  mutable bool have_cells, have_sim;

 private:
  DPMatrix(const DPMatrix&);
  DPMatrix& operator=(const DPMatrix&);
};

// prints the whole score matrix, tab separated (reference dpmatrix.h:116-129)
template <class S1, class S2, class Etype>
std::ostream& operator<<(std::ostream& o, const DPMatrix<S1, S2, Etype>& dpm) {
  const int ql = dpm.getQuerySize(), tl = dpm.getTemplateSize();
  for (int i = 0; i < ql; ++i) {
    for (int j = 0; j < tl; ++j) o << dpm.getCell(i, j)->score << "\t";
    o << std::endl;
  }
  return o;
}
#endif
