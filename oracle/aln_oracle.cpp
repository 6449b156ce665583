// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See aln_oracle.h.
//
// Literal CPU restatement of the reference's O(n^3) dynamic programme, its tracebacks,
// its Waterman-style near-optimal enumerators and its gapped-string writer.  C-style code
// compiled as C++ only so that sortSet can call the very same libstdc++ std::sort /
// std::partial_sort the reference calls (alignment.h:926-930): the order of equal-score
// alignments after an unstable sort is part of the observable behaviour.
#include "aln_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline float clip0(float s, int local) {
  // std::max(0.f, s) as written in dpmatrix.h:580 — returns 0.f unless 0.f < s
  return local ? ((0.f < s) ? s : 0.f) : s;
}

struct Mat {
  int Q, T;
  float* D; int* PQ; int* PT;
  inline void setTB(int i, int j, int pq, int pt, float s) {  // dpmatrix.cpp:27-32
    PQ[i * T + j] = pq; PT[i * T + j] = pt; D[i * T + j] = s;
  }
  inline float d(int i, int j) const { return D[i * T + j]; }
};

}  // namespace

extern "C" {

// aasubalib.h:27-51 / hmap2_eval.h:41-67
float orc_deletion(const orc_gap* g, int Q, int T, int q1, int q2, int t1, int t2, int* err) {
  (void)Q;
  if (g->model == ORC_GAP_CALLBACK) return g->del_cb(q1, q2, t1, t2);
  if (g->model == ORC_GAP_GN2) {                   // gn2_eval.h:100-130
    int di = t2 - t1;
    if (di < 2) return 0;
    int p1 = t1, p2 = t2 - 2;
    float GP = 8100.f;
    if (g->dist[(size_t)p2 * T + p1] < 18.f)
      GP = g->vvgi[(size_t)p2 * T + p1] + g->vvge[(size_t)p2 * T + p1] * (di - 2) + g->vvcd[(size_t)p2 * T + p1];
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_GLOBAL_LOCAL:
        return GP;
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_LOCAL_GLOBAL:
        if (t1 == 0 || t2 == T - 1) return 0;
        return GP;
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  }
  if (g->model == ORC_GAP_AFFINE_CONST) {
    int len = t2 - t1 - 1;
    if (len < 1) return 0.f;
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_GLOBAL_LOCAL:
        return g->gi + g->ge * (len - 1);
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_LOCAL_GLOBAL:
        if (t1 == 0 || t2 == T - 1) return 0;
        return g->gi + g->ge * (len - 1);
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  } else {
    int dist = t2 - t1;
    if (dist < 2) return 0;
    float gi = std::min(g->tgi[t1], g->tgi[t2]);
    float ge = std::min(g->tge[t1], g->tge[t2]);
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_GLOBAL_LOCAL:
        return gi + ge * (dist - 2);
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_LOCAL_GLOBAL:
        if (t1 == 0 || t2 == T - 1) return 0;
        return gi + ge * (dist - 2);
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  }
}

// aasubalib.h:53-77 / hmap2_eval.h:69-95 (coefficients come from the TEMPLATE positions t1,t2)
float orc_insertion(const orc_gap* g, int Q, int T, int q1, int q2, int t1, int t2, int* err) {
  (void)T;
  if (g->model == ORC_GAP_CALLBACK) return g->ins_cb(q1, q2, t1, t2);
  if (g->model == ORC_GAP_GN2) {                   // gn2_eval.h:132-165
    int di = q2 - q1;
    if (di < 2) return 0;
    float GP = g->tgi[t1] + g->tge[t1] * (di - 2) + g->tcn[t1];
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_LOCAL_GLOBAL:
        return GP;
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_GLOBAL_LOCAL:
        if (q1 == 0 || q2 == Q - 1) return 0;
        return GP;
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  }
  if (g->model == ORC_GAP_AFFINE_CONST) {
    int len = q2 - q1 - 1;
    if (len < 1) return 0.f;
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_LOCAL_GLOBAL:
        return g->gi + g->ge * (len - 1);
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_GLOBAL_LOCAL:
        if (q1 == 0 || q2 == Q - 1) return 0;
        return g->gi + g->ge * (len - 1);
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  } else {
    int dist = q2 - q1;
    if (dist < 2) return 0;
    float gi = std::min(g->tgi[t1], g->tgi[t2]);
    float ge = std::min(g->tge[t1], g->tge[t2]);
    switch (g->align_type) {
      case ORC_GLOBAL: case ORC_LOCAL_GLOBAL:
        return gi + ge * (dist - 2);
      case ORC_LOCAL: case ORC_SEMI_LOCAL: case ORC_GLOBAL_LOCAL:
        if (q1 == 0 || q2 == Q - 1) return 0;
        return gi + ge * (dist - 2);
      default: if (err) *err = ORC_E_GAPSTYLE; return 0.f;
    }
  }
}

// simmatrix.h:51-72 with aasubalib.h:17-25 as the evaluator
int orc_sim_submatrix(int Q, int T, const char* qres, const char* tres,
                      const char* alphabet, int n, const float* table, float* S) {
  int idx[256];
  for (int i = 0; i < 256; ++i) idx[i] = -1;
  for (int i = 0; i < n; ++i) idx[(unsigned char)alphabet[i]] = i;
  for (int i = 0; i < Q; ++i) { S[i * T + 0] = 0.f; S[i * T + T - 1] = 0.f; }
  for (int j = 0; j < T; ++j) { S[0 * T + j] = 0.f; S[(Q - 1) * T + j] = 0.f; }
  for (int i = 1; i < Q - 1; ++i)
    for (int j = 1; j < T - 1; ++j) {
      char a = qres[i], b = tres[j];
      if (a == '^' || a == '$' || b == '^' || b == '$') { S[i * T + j] = 0.f; continue; }
      int ia = idx[(unsigned char)a], ib = idx[(unsigned char)b];
      if (ia < 0 || ib < 0) return ORC_E_RESIDUE;
      S[i * T + j] = table[ia * n + ib];
    }
  return ORC_OK;
}

// hmath.h:18-26 — element-wise products then valarray::sum() (sequential, starting from T())
float orc_dot(const float* a, const float* b, int n) {
  float r = 0.f;
  for (int i = 0; i < n; ++i) { float p = a[i] * b[i]; r += p; }
  return r;
}

namespace {
// hmath.h:43-60
void norm_vec(float* res, const float* v, int n) {
  float sum = 0.f;
  for (int i = 0; i < n; ++i) sum += v[i];
  float sumsq = 0.f;
  for (int i = 0; i < n; ++i) { float s = v[i] * v[i]; sumsq += s; }
  float avg = sum / float(n);
  float var = sumsq / float(n) - avg * avg;
  float sd = std::sqrt(var);
  for (int i = 0; i < n; ++i) { float x = v[i]; x -= avg; x /= sd; res[i] = x; }
}
}  // namespace

// hmath.h:94-103
float orc_pearson(const float* a, const float* b, int n) {
  std::vector<float> n1(n), n2(n);
  norm_vec(n1.data(), a, n);
  norm_vec(n2.data(), b, n);
  return orc_dot(n1.data(), n2.data(), n) / float(n);
}

// hmap2_eval.h:27-39
int orc_sim_hmap2(int Q, int T, const float* q_aa, const float* q_sse, const float* q_conf,
                  const float* t_aa, const float* t_sse, const float* t_conf, float alpha, float* S) {
  for (int i = 0; i < Q; ++i) { S[i * T + 0] = 0.f; S[i * T + T - 1] = 0.f; }
  for (int j = 0; j < T; ++j) { S[0 * T + j] = 0.f; S[(Q - 1) * T + j] = 0.f; }
  for (int i = 1; i < Q - 1; ++i)
    for (int j = 1; j < T - 1; ++j) {
      float ip = orc_dot(q_aa + 20 * i, t_aa + 20 * j, 20);
      float pc = orc_pearson(q_sse + 3 * i, t_sse + 3 * j, 3);
      float sim = ip * expf(alpha * pc * q_conf[i] * t_conf[j]);
      S[i * T + j] = sim;
    }
  return ORC_OK;
}

// hmap2_eval.h:98-101 -> hmath.h:62-92 over the interior [1,Q-1) x [1,T-1)
int orc_norm_shift(int Q, int T, float* S, float zero_shift) {
  int i0 = 1, i1 = Q - 1, j0 = 1, j1 = T - 1;
  if (i0 >= i1 || j0 >= j1) { i0 = 0; j0 = 0; i1 = Q; j1 = T; }
  int n = (i1 - i0) * (j1 - j0);
  std::vector<float> v1(n), v2(n);
  int c = 0;
  for (int i = i0; i < i1; ++i) for (int j = j0; j < j1; ++j) v1[c++] = S[i * T + j];
  norm_vec(v2.data(), v1.data(), n);
  c = 0;
  for (int i = i0; i < i1; ++i) for (int j = j0; j < j1; ++j) S[i * T + j] = v2[c++];
  float shift = -zero_shift;
  // shift_elements is called with the original (1..rows-1) bounds, hmap2_eval.h:100
  i0 = 1; i1 = Q - 1; j0 = 1; j1 = T - 1;
  if (i0 >= i1 || j0 >= j1) { i0 = 0; j0 = 0; i1 = Q; j1 = T; }
  for (int i = i0; i < i1; ++i) for (int j = j0; j < j1; ++j) S[i * T + j] = S[i * T + j] + shift;
  return ORC_OK;
}

// hmap2_eval.cpp:17-25
void orc_hmap2_precalc(int T, const float* t_pcoil, float gi, float ge, float beta, float* tgi, float* tge) {
  for (int i = 0; i < T; ++i) {
    float Pi = expf(beta * (1.f - 1.25f * t_pcoil[i]));
    tgi[i] = gi * Pi;
    tge[i] = ge * Pi;
  }
}

// dpmatrix.cpp:17-25
void orc_dp_init(int Q, int T, float* D, int* PQ, int* PT) {
  for (int k = 0; k < Q * T; ++k) { D[k] = 0.f; PQ[k] = -1; PT[k] = -1; }
}

namespace {

// dpmatrix.h:356-536 (local=0) and :538-689 (local=1)
int build_forward(Mat& m, const float* S, const orc_gap* g, int local, int q0, int q1, int t0, int t1) {
  const int Q = m.Q, T = m.T;
  int err = 0;
  if (q1 <= q0 || t1 <= t0) return ORC_E_BOUNDS;
  float s_initial = m.d(q0, t0);
  int q0_p1 = q0 + 1, t0_p1 = t0 + 1, q0_p2 = q0 + 2, t0_p2 = t0 + 2, q1_m1 = q1 - 1, t1_m1 = t1 - 1;
  float s;
  if (q1 == q0_p1) {              // :375-381 / :558-564 (no clip in either)
    s = s_initial;
    s -= orc_deletion(g, Q, T, q0, q1, t0, t1, &err);
    s += S[q1 * T + t1];
    m.setTB(q1, t1, q0, t0, s);
    return err;
  }
  if (t1 == t0_p1) {              // :384-390 / :567-573
    s = s_initial;
    s -= orc_insertion(g, Q, T, q0, q1, t0, t1, &err);
    s += S[q1 * T + t1];
    m.setTB(q1, t1, q0, t0, s);
    return err;
  }
  s = s_initial + S[q0_p1 * T + t0_p1];           // :409-410 / :579-581
  s = clip0(s, local);
  m.setTB(q0_p1, t0_p1, q0, t0, s);
  for (int j = t0_p2; j < t1; ++j) {               // :413-418 / :584-590
    s = s_initial;
    s -= orc_deletion(g, Q, T, q0, q0_p1, t0, j, &err);
    s += S[q0_p1 * T + j];
    s = clip0(s, local);
    m.setTB(q0_p1, j, q0, t0, s);
  }
  for (int i = q0_p2; i < q1; ++i) {               // :421-426 / :593-599
    s = s_initial;
    s -= orc_insertion(g, Q, T, q0, i, t0, t0_p1, &err);
    s += S[i * T + t0_p1];
    s = clip0(s, local);
    m.setTB(i, t0_p1, q0, t0, s);
  }
  int opt_i, opt_j; float opt_s;
  for (int i = q0_p2; i < q1; ++i) {               // :447-486 / :607-649
    for (int j = t0_p2; j < t1; ++j) {
      int i_m1 = i - 1, j_m1 = j - 1;
      opt_i = i_m1; opt_j = j_m1;
      opt_s = m.d(opt_i, opt_j) + S[i * T + j];
      opt_s = clip0(opt_s, local);
      for (int k = t0_p1; k < j_m1; ++k) {
        s = m.d(i_m1, k);
        s -= orc_deletion(g, Q, T, i_m1, i, k, j, &err);
        s += S[i * T + j];
        s = clip0(s, local);
        if (s > opt_s) { opt_i = i_m1; opt_j = k; opt_s = s; }
      }
      for (int k = q0_p1; k < i_m1; ++k) {
        s = m.d(k, j_m1);
        s -= orc_insertion(g, Q, T, k, i, j_m1, j, &err);
        s += S[i * T + j];
        s = clip0(s, local);
        if (s > opt_s) { opt_i = k; opt_j = j_m1; opt_s = s; }
      }
      m.setTB(i, j, opt_i, opt_j, opt_s);
    }
  }
  opt_i = q1_m1; opt_j = t1_m1;                    // :505-534 / :655-687
  opt_s = m.d(opt_i, opt_j) + S[q1 * T + t1];
  opt_s = clip0(opt_s, local);
  for (int k = t0_p1; k < t1; ++k) {
    s = m.d(q1_m1, k);
    s -= orc_deletion(g, Q, T, q1_m1, q1, k, t1, &err);
    s += S[q1 * T + t1];
    s = clip0(s, local);
    if (s > opt_s) { opt_i = q1_m1; opt_j = k; opt_s = s; }
  }
  for (int k = q0_p1; k < q1; ++k) {
    s = m.d(k, t1_m1);
    s -= orc_insertion(g, Q, T, k, q1, t1_m1, t1, &err);
    s += S[q1 * T + t1];
    s = clip0(s, local);
    if (s > opt_s) { opt_i = k; opt_j = t1_m1; opt_s = s; }
  }
  m.setTB(q1, t1, opt_i, opt_j, opt_s);
  return err;
}

// dpmatrix.h:691-877 (local=0; bug_b4 reproduces :868) and :879-1030 (local=1)
int build_reverse(Mat& m, const float* S, const orc_gap* g, int local, int q0, int q1, int t0, int t1, int bug_b4) {
  const int Q = m.Q, T = m.T;
  int err = 0;
  if (q1 <= q0 || t1 <= t0) return ORC_E_BOUNDS;
  float s_initial = m.d(q1, t1);
  int q0_p1 = q0 + 1, t0_p1 = t0 + 1, q1_m2 = q1 - 2, t1_m2 = t1 - 2, q1_m1 = q1 - 1, t1_m1 = t1 - 1;
  float s;
  if (q1 == q0_p1) {
    s = s_initial;
    s -= orc_deletion(g, Q, T, q0, q1, t0, t1, &err);
    s += S[q0 * T + t0];
    m.setTB(q0, t0, q1, t1, s);
    return err;
  }
  if (t1 == t0_p1) {
    s = s_initial;
    s -= orc_insertion(g, Q, T, q0, q1, t0, t1, &err);
    s += S[q0 * T + t0];
    m.setTB(q0, t0, q1, t1, s);
    return err;
  }
  s = s_initial + S[q1_m1 * T + t1_m1];
  s = clip0(s, local);
  m.setTB(q1_m1, t1_m1, q1, t1, s);
  for (int j = t1_m2; j > t0; --j) {
    s = s_initial;
    s -= orc_deletion(g, Q, T, q1_m1, q1, j, t1, &err);
    s += S[q1_m1 * T + j];
    s = clip0(s, local);
    m.setTB(q1_m1, j, q1, t1, s);
  }
  for (int i = q1_m2; i > q0; --i) {
    s = s_initial;
    s -= orc_insertion(g, Q, T, i, q1, t1_m1, t1, &err);
    s += S[i * T + t1_m1];
    s = clip0(s, local);
    m.setTB(i, t1_m1, q1, t1, s);
  }
  int opt_i, opt_j; float opt_s;
  for (int i = q1_m2; i > q0; --i) {
    for (int j = t1_m2; j > t0; --j) {
      int i_p1 = i + 1, j_p1 = j + 1;
      opt_i = i_p1; opt_j = j_p1;
      opt_s = m.d(opt_i, opt_j) + S[i * T + j];
      opt_s = clip0(opt_s, local);
      for (int k = t1_m1; k > j_p1; --k) {
        s = m.d(i_p1, k);
        s -= orc_deletion(g, Q, T, i, i_p1, j, k, &err);
        s += S[i * T + j];
        s = clip0(s, local);
        if (s > opt_s) { opt_i = i_p1; opt_j = k; opt_s = s; }
      }
      for (int k = q1_m1; k > i_p1; --k) {
        s = m.d(k, j_p1);
        s -= orc_insertion(g, Q, T, i, k, j, j_p1, &err);
        s += S[i * T + j];
        s = clip0(s, local);
        if (s > opt_s) { opt_i = k; opt_j = j_p1; opt_s = s; }
      }
      m.setTB(i, j, opt_i, opt_j, opt_s);
    }
  }
  opt_i = q0_p1; opt_j = t0_p1;
  opt_s = m.d(opt_i, opt_j) + S[q0 * T + t0];
  opt_s = clip0(opt_s, local);
  for (int k = t1_m1; k > t0; --k) {
    s = m.d(q0_p1, k);
    s -= orc_deletion(g, Q, T, q0, q0_p1, t0, k, &err);
    s += S[q0 * T + t0];
    s = clip0(s, local);
    if (s > opt_s) { opt_i = q0_p1; opt_j = k; opt_s = s; }
  }
  for (int k = q1_m1; k > q0; --k) {
    s = m.d(k, t0_p1);
    s -= orc_insertion(g, Q, T, q0, k, t0, t0_p1, &err);
    s += S[q0 * T + t0];
    s = clip0(s, local);
    if (s > opt_s) {
      opt_i = k;
      opt_j = (!local && bug_b4) ? t1_m1 : t0_p1;   // dpmatrix.h:868 stores t1_m1 (B4); :1022 is correct
      opt_s = s;
    }
  }
  m.setTB(q0, t0, opt_i, opt_j, opt_s);
  return err;
}

}  // namespace

int orc_dp_build(int Q, int T, const float* S, const orc_gap* gap, int direction, int islocal,
                 int q0, int q1, int t0, int t1, int bug_b4, float* D, int* PQ, int* PT) {
  Mat m = {Q, T, D, PQ, PT};
  // build(): dpmatrix.h:306-307 ; build_subdpm(): :333-334
  D[q0 * T + t0] = 0.f;
  D[q1 * T + t1] = 0.f;
  if (direction == ORC_FWD) return build_forward(m, S, gap, islocal, q0, q1, t0, t1);
  return build_reverse(m, S, gap, islocal, q0, q1, t0, t1, bug_b4);
}

// optimal.h:48-124
int orc_optimal(int Q, int T, const float* D, const int* PQ, const int* PT, int islocal,
                int* pairs, int* npairs, float* score) {
  std::vector<int> rq, rt;   // reversed list (prepend == push_back)
  int q_last = Q - 1, t_last = T - 1;
  if (!islocal) {
    *score = D[q_last * T + t_last];
    rq.push_back(q_last); rt.push_back(t_last);
    while (q_last > 0) {
      int c = q_last * T + t_last;
      q_last = PQ[c]; t_last = PT[c];
      rq.push_back(q_last); rt.push_back(t_last);
      if (q_last < 0 || t_last < 0) break;   // reference would index out of bounds; stop instead
    }
    int n = (int)rq.size();
    for (int k = 0; k < n; ++k) { pairs[2 * k] = rq[n - 1 - k]; pairs[2 * k + 1] = rt[n - 1 - k]; }
    *npairs = n;
    if (q_last != 0 || t_last != 0) return ORC_E_STARTPAIR;
    return ORC_OK;
  }
  rq.push_back(q_last); rt.push_back(t_last);
  // find_max, optimal.h:108-124: seeded with (Q-2,T-2); strict '<' replaces
  int mq = Q - 2, mt = T - 2; float ms = D[mq * T + mt];
  for (int i = 0; i < Q - 1; ++i)
    for (int j = 0; j < T - 1; ++j)
      if (ms < D[i * T + j]) { mq = i; mt = j; ms = D[i * T + j]; }
  q_last = mq; t_last = mt;
  *score = ms;
  rq.push_back(q_last); rt.push_back(t_last);
  while (q_last > 0) {
    int c = q_last * T + t_last;
    q_last = PQ[c]; t_last = PT[c];
    if (q_last < 0 || t_last < 0) break;     // untouched cell (-1,-1): reference reads out of bounds
    if (D[q_last * T + t_last] <= 0.f) break;
    rq.push_back(q_last); rt.push_back(t_last);
  }
  if (q_last != 0 && t_last != 0) { rq.push_back(0); rt.push_back(0); }
  int n = (int)rq.size();
  for (int k = 0; k < n; ++k) { pairs[2 * k] = rq[n - 1 - k]; pairs[2 * k + 1] = rt[n - 1 - k]; }
  *npairs = n;
  return ORC_OK;
}

// optimal_rev.h:44-131
int orc_optimal_rev(int Q, int T, const float* D, const int* PQ, const int* PT, int islocal,
                    int* pairs, int* npairs, float* score) {
  int q_last = Q - 1, t_last = T - 1, q_first = 0, t_first = 0, n = 0;
  if (!islocal) {
    *score = D[0];
    pairs[0] = 0; pairs[1] = 0; n = 1;
    while (q_first < q_last) {
      int c = q_first * T + t_first;
      q_first = PQ[c]; t_first = PT[c];
      pairs[2 * n] = q_first; pairs[2 * n + 1] = t_first; ++n;
      if (q_first < 0 || t_first < 0) break;
    }
    *npairs = n;
    if (q_first != q_last || t_first != t_last) return ORC_E_STARTPAIR;
    return ORC_OK;
  }
  pairs[0] = 0; pairs[1] = 0; n = 1;
  int mq = 0, mt = 0; float ms = D[0];                 // :117-131
  for (int i = Q - 1; i > 0; --i)
    for (int j = T - 1; j > 0; --j)
      if (ms < D[i * T + j]) { mq = i; mt = j; ms = D[i * T + j]; }
  q_first = mq; t_first = mt; *score = ms;
  pairs[2 * n] = q_first; pairs[2 * n + 1] = t_first; ++n;
  while (q_first < q_last) {
    int c = q_first * T + t_first;
    q_first = PQ[c]; t_first = PT[c];
    if (q_first < 0 || t_first < 0) break;
    if (D[q_first * T + t_first] <= 0.f) break;
    pairs[2 * n] = q_first; pairs[2 * n + 1] = t_first; ++n;
  }
  if (q_first != q_last && t_first != t_last) { pairs[2 * n] = q_last; pairs[2 * n + 1] = t_last; ++n; }
  *npairs = n;
  return ORC_OK;
}

// optimal_subali.h:60-84
int orc_optimal_subali(int Q, int T, const float* D, const int* PQ, const int* PT,
                       int q1_end, int t1_end, int q2_beg, int t2_beg,
                       int* pairs, int* npairs, float* score) {
  (void)Q;
  std::vector<int> rq, rt;
  int q_last = q2_beg, t_last = t2_beg;
  *score = D[q_last * T + t_last];
  rq.push_back(q_last); rt.push_back(t_last);
  while (q_last > q1_end) {
    int c = q_last * T + t_last;
    q_last = PQ[c]; t_last = PT[c];
    rq.push_back(q_last); rt.push_back(t_last);
    if (q_last < 0 || t_last < 0) break;
  }
  int n = (int)rq.size();
  for (int k = 0; k < n; ++k) { pairs[2 * k] = rq[n - 1 - k]; pairs[2 * k + 1] = rt[n - 1 - k]; }
  *npairs = n;
  if (q_last != q1_end || t_last != t1_end) return ORC_E_STARTPAIR;
  return ORC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Alignment sets.  An alignment keeps its pairs REVERSED (back() is the list's front) because the
// enumerators only ever prepend (alignment.h:851-853).
struct OrcAli {
  std::vector<int> rq, rt;
  float score, identity; int uid;
  OrcAli() : score(0.f), identity(0.f), uid(-1) {}                        // alignment.h:55-56
  bool operator<(const OrcAli& a) const { return score > a.score; }      // alignment.h:104-105
  void prepend(int q, int t) { rq.push_back(q); rt.push_back(t); }
};
struct orc_set { std::vector<OrcAli> v; };

extern "C" {

orc_set* orc_set_new(void) { return new orc_set(); }
void orc_set_free(orc_set* s) { delete s; }
int orc_set_size(const orc_set* s) { return (int)s->v.size(); }
void orc_set_push(orc_set* s, const int* pairs, int npairs, float score, int uid) {
  OrcAli a; a.score = score; a.uid = uid;
  for (int k = npairs - 1; k >= 0; --k) a.prepend(pairs[2 * k], pairs[2 * k + 1]);
  s->v.push_back(a);
}
int orc_set_npairs(const orc_set* s, int k) { return (int)s->v[k].rq.size(); }
void orc_set_get(const orc_set* s, int k, int* pairs, float* score, float* identity, int* uid) {
  const OrcAli& a = s->v[k];
  int n = (int)a.rq.size();
  for (int i = 0; i < n; ++i) { pairs[2 * i] = a.rq[n - 1 - i]; pairs[2 * i + 1] = a.rt[n - 1 - i]; }
  if (score) *score = a.score;
  if (identity) *identity = a.identity;
  if (uid) *uid = a.uid;
}
// alignment.h:922-932
void orc_set_sort(orc_set* s, int max) {
  std::vector<OrcAli>& v = s->v;
  if (max >= (int)v.size()) std::sort(v.begin(), v.end());
  else if (max > 0) {
    std::partial_sort(v.begin(), v.begin() + max, v.end());
    v.erase(v.begin() + max, v.end());
  }
}
// alignment.h:856-865
void orc_set_identity(orc_set* s, const char* qstr, const char* tstr) {
  int total = (int)std::min(strlen(qstr), strlen(tstr)) - 2;
  for (size_t a = 0; a < s->v.size(); ++a) {
    int same = -2;
    OrcAli& al = s->v[a];
    for (size_t i = 0; i < al.rq.size(); ++i)
      if (qstr[al.rq[i]] == tstr[al.rt[i]]) ++same;
    al.identity = float(same) / float(total) * 100.f;
  }
}

}  // extern "C"

namespace {

struct Enum {
  int kind, Q, T;
  const float* D; const int* PQ; const int* PT; const float* S;
  const orc_gap* gap; const unsigned char* flags;
  unsigned user_limit; float thr;
  std::vector<OrcAli>* as;
  int err;
  float d(int i, int j) const { return D[i * T + j]; }
  float del(int a, int b, int c, int e) { return orc_deletion(gap, Q, T, a, b, c, e, &err); }
  float ins(int a, int b, int c, int e) { return orc_insertion(gap, Q, T, a, b, c, e, &err); }

  void base_case(int q0, int t0, int k0) {                 // cw.h:100-108 / ucw.h:93-101
    (*as)[k0].prepend(q0, t0);
    (*as)[k0].prepend(0, 0);
    (*as)[k0].score += d(q0, t0);
  }

  // cw.h:215-284
  void opt_path_cw(int q0, int t0, int k0, bool force_opt) {
    if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); return; }
    int pq = -1, pt = -1;
    bool flag = !flags[t0];
    while (t0 > 1 && q0 > 1) {
      if (!force_opt && (flags[t0] != 0) == flag) break;
      (*as)[k0].prepend(q0, t0);
      (*as)[k0].score += S[q0 * T + t0];
      pq = PQ[q0 * T + t0];
      pt = PT[q0 * T + t0];
      float g;
      if (q0 - pq == 1) g = del(pq, q0, pt, t0);
      else g = ins(pq, q0, pt, t0);
      (*as)[k0].score -= g;
      t0 = pt; q0 = pq;
    }
    branch_cw(pq, pt, k0, force_opt);
  }

  // cw.h:95-212
  void branch_cw(int q0, int t0, int k0, bool force_opt) {
    if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); return; }
    if (force_opt) { opt_path_cw(q0, t0, k0, force_opt); return; }
    int k = k0;
    float f, r, g;
    OrcAli curr((*as)[k0]);
    if (as->size() > user_limit) { opt_path_cw(q0, t0, k0, true); return; }
    r = curr.score + S[q0 * T + t0];
    f = d(q0 - 1, t0 - 1);
    if (f + r > thr) {
      if ((int)as->size() == k) as->push_back(curr);
      (*as)[k].prepend(q0, t0);
      (*as)[k].score = r;
      opt_path_cw(q0 - 1, t0 - 1, k, force_opt);
      k = (int)as->size();
    }
    for (int i = t0 - 2; i > 0; --i) {
      f = d(q0 - 1, i);
      g = del(q0 - 1, q0, i, t0);
      if (f + r - g > thr) {
        if ((int)as->size() == k) as->push_back(curr);
        (*as)[k].prepend(q0, t0);
        (*as)[k].score = r - g;
        opt_path_cw(q0 - 1, i, k, force_opt);
        k = (int)as->size();
      }
    }
    for (int j = q0 - 2; j > 0; --j) {
      f = d(j, t0 - 1);
      g = ins(j, q0, t0 - 1, t0);
      if (f + r - g > thr) {
        if ((int)as->size() == k) as->push_back(curr);
        (*as)[k].prepend(q0, t0);
        (*as)[k].score = r - g;
        opt_path_cw(j, t0 - 1, k, force_opt);
        k = (int)as->size();
      }
    }
    if (k == k0) opt_path_cw(q0, t0, k0, true);
  }

  // ucw.h:194-236
  void opt_path_ucw(int q0, int t0, int k0) {
    int pq = -1, pt = -1;
    while (t0 > 1 && q0 > 1) {
      (*as)[k0].prepend(q0, t0);
      (*as)[k0].score += S[q0 * T + t0];
      pq = PQ[q0 * T + t0];
      pt = PT[q0 * T + t0];
      float g;
      if (q0 - pq == 1) g = del(pq, q0, pt, t0);
      else g = ins(pq, q0, pt, t0);
      (*as)[k0].score -= g;
      t0 = pt; q0 = pq;
    }
    base_case(q0, t0, k0);
  }

  // ucw.h:87-192
  void branch_ucw(int q0, int t0, int k0) {
    if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); return; }
    int k = k0;
    float f, r, g;
    OrcAli curr((*as)[k0]);
    if (as->size() > user_limit) { opt_path_ucw(q0, t0, k0); return; }
    r = curr.score + S[q0 * T + t0];
    f = d(q0 - 1, t0 - 1);
    if (f + r > thr) {
      if ((int)as->size() == k) as->push_back(curr);
      (*as)[k].prepend(q0, t0);
      (*as)[k].score = r;
      branch_ucw(q0 - 1, t0 - 1, k);
      k = (int)as->size();
    }
    for (int i = t0 - 2; i > 0; --i) {
      f = d(q0 - 1, i);
      g = del(q0 - 1, q0, i, t0);
      if (f + r - g > thr) {
        if ((int)as->size() == k) as->push_back(curr);
        (*as)[k].prepend(q0, t0);
        (*as)[k].score = r - g;
        branch_ucw(q0 - 1, i, k);
        k = (int)as->size();
      }
    }
    for (int j = q0 - 2; j > 0; --j) {
      f = d(j, t0 - 1);
      g = ins(j, q0, t0 - 1, t0);
      if (f + r - g > thr) {
        if ((int)as->size() == k) as->push_back(curr);
        (*as)[k].prepend(q0, t0);
        (*as)[k].score = r - g;
        branch_ucw(j, t0 - 1, k);
        k = (int)as->size();
      }
    }
    if (k == k0) opt_path_ucw(q0, t0, k0);
  }

  // ---- KSConstrainedNearOptimal (kscw.h:109-351): every branch node keeps only its k_limit best operations -------------
  struct op_data {                                         // kscw.h:38-46
    unsigned int limit;
    int q0, t0, k0;
    float score, thresh, new_r;
    op_data(unsigned int l, int q, int t, int k, float th, float s = 0.f, float n = 0.f)
        : limit(l), q0(q), t0(t), k0(k), score(s), thresh(th), new_r(n) {}
    bool operator<(const op_data& a) const { return score > a.score; }
  };

  // kscw.h:291-351
  void opt_path_ks(op_data& op, bool force_opt) {
    unsigned int k_limit = op.limit;
    int q0 = op.q0, t0 = op.t0, k0 = op.k0;
    float threshold = op.thresh;
    if (k_limit <= 1) force_opt = true;
    if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); return; }
    int pq = -1, pt = -1;
    bool flag = !flags[t0];
    while (t0 > 1 && q0 > 1) {
      if (!force_opt && (flags[t0] != 0) == flag) break;
      (*as)[k0].prepend(q0, t0);
      (*as)[k0].score += S[q0 * T + t0];
      pq = PQ[q0 * T + t0];
      pt = PT[q0 * T + t0];
      float g;
      if (q0 - pq == 1) g = del(pq, q0, pt, t0);
      else g = ins(pq, q0, pt, t0);
      (*as)[k0].score -= g;
      t0 = pt; q0 = pq;
    }
    op_data nop(k_limit, pq, pt, k0, threshold);
    branch_ks(nop);
  }

  // kscw.h:139-288
  void branch_ks(op_data& op) {
    unsigned int k_limit = op.limit;
    int q0 = op.q0, t0 = op.t0, k0 = op.k0;
    float threshold = op.thresh;
    if (q0 == 1 || t0 == 1) { base_case(q0, t0, k0); return; }
    if (q0 < 1 || t0 < 1) { err = ORC_E_ARG; return; }     // the reference would index row/column -1 here
    float f, r, g, sum;
    OrcAli curr((*as)[k0]);
    std::vector<op_data> k_sort;
    k_sort.reserve(q0 + t0);
    if (as->size() > user_limit) { opt_path_ks(op, true); return; }
    r = curr.score + S[q0 * T + t0];
    f = d(q0 - 1, t0 - 1);
    sum = f + r;
    if (sum > threshold) k_sort.push_back(op_data(k_limit / 2, q0 - 1, t0 - 1, k0, threshold, sum, r));
    for (int i = t0 - 2; i > 0; --i) {
      f = d(q0 - 1, i);
      g = del(q0 - 1, q0, i, t0);
      sum = f + r - g;
      if (sum > threshold) k_sort.push_back(op_data(k_limit / 2, q0 - 1, i, k0, threshold, sum, r - g));
    }
    for (int j = q0 - 2; j > 0; --j) {
      f = d(j, t0 - 1);
      g = ins(j, q0, t0 - 1, t0);
      sum = f + r - g;
      if (sum > threshold) k_sort.push_back(op_data(k_limit / 2, j, t0 - 1, k0, threshold, sum, r - g));
    }
    if (k_sort.size() == 0) {
      op_data new_op(1, q0, t0, k0, threshold);
      opt_path_ks(new_op, true);
      return;
    }
    if (k_sort.size() > k_limit) {
      std::partial_sort(k_sort.begin(), k_sort.begin() + k_limit, k_sort.end());
      k_sort.erase(k_sort.begin() + k_limit, k_sort.end());
    } else {
      std::sort(k_sort.begin(), k_sort.end());
    }
    std::vector<op_data>::iterator it = k_sort.begin();
    it->limit *= 2;                                        // only the best operation keeps the node's own limit
    for (int k = k0; it != k_sort.end(); ++it) {
      it->k0 = k;
      if ((int)as->size() == k) { as->push_back(curr); (*as)[k].uid = k; }
      (*as)[k].prepend(q0, t0);
      (*as)[k].score = it->new_r;
      opt_path_ks(*it, false);
      k = (int)as->size();
    }
  }

  // ---- CRConstrainedNearOptimal (crcw.h:134-594): "controlled redundancy" — a branch node sorts its operations, follows each
  // one along the stored pointers to the end of the template's current flag region, drops the operations whose sub-path shares
  // more than max_overlap of an accepted, better one's, and recurses on the survivors (at most the node's limit). ---------
  struct cr_op {                                           // crcw.h:47-56
    unsigned int limit, index;
    int q0, t0, k0;
    float score, new_r;
    cr_op(unsigned int l, int q, int t, int k, float s = 0.f, float n = 0.f) : limit(l), index(0), q0(q), t0(t), k0(k), score(s), new_r(n) {}
    bool operator<(const cr_op& a) const { return score > a.score; }
  };
  unsigned cr_sort_limit; float cr_max_overlap;
  std::vector<int> cr_regions;                             // crcw.h:174-179: regions[i] = flips of the flags among positions 0 .. i+1
  std::vector<std::vector<int> > cr_ali;                   // alignments[sort_limit][t_last]
  long cr_oob;                                             // times the reference would have read regions[-1] (crcw.h:387)
  // The reference indexes regions[t-1] with t == 0 when a sub-path reaches the matrix origin (crcw.h:387): a read of the heap word
  // in front of the array (UB).  With glibc that word is the chunk size, a number far above any region count, and the value is
  // only compared for equality with other sub-paths' end states — so sub-paths ending at t == 0 form a class of their own.
  // kOriginState stands for that class; cr_oob counts how often it was needed.
  enum { kOriginState = -1 };
  int cr_region_at(int t) { if (t < 1) { ++cr_oob; return kOriginState; } return cr_regions[t - 1]; }

  // crcw.h:552-592
  void force_opt_path_cr(cr_op& op) {
    int pq = -1, pt = -1;
    int q0 = op.q0, t0 = op.t0, k0 = op.k0;
    while (t0 > 0 && q0 > 0) {
      (*as)[k0].prepend(q0, t0);
      (*as)[k0].score += S[q0 * T + t0];
      pq = PQ[q0 * T + t0];
      pt = PT[q0 * T + t0];
      float g;
      if (q0 - pq == 1) g = del(pq, q0, pt, t0);
      else g = ins(pq, q0, pt, t0);
      (*as)[k0].score -= g;
      t0 = pt; q0 = pq;
    }
    (*as)[k0].prepend(0, 0);
  }

  // crcw.h:345-550
  void filter_and_extend_cr(int q0, int t0, std::vector<cr_op>& v_op) {
    const int end_alignment = 2;
    const size_t n = v_op.size();
    std::vector<char> filter(n);
    std::vector<int> p_rq(n), p_rt(n), l_sp(n), state(n);
    std::vector<float> rs(n);
    for (size_t i = 0; i < n; ++i)                         // reinit_mem(t0, n)
      for (int j = 0; j < t0; ++j) cr_ali[i][j] = -1;
    for (size_t i = 0; i < n; ++i) {
      float g;
      int pq, pt;
      v_op[i].index = (unsigned)i;
      int q = v_op[i].q0;
      int t = v_op[i].t0;
      l_sp[i] = 1;
      state[i] = cr_region_at(t);
      rs[i] = v_op[i].new_r;
      while (q > 0 && t > 0 && cr_regions[t - 1] == state[i]) {
        cr_ali[i][t - 1] = q;
        ++l_sp[i];
        pq = PQ[q * T + t];
        pt = PT[q * T + t];
        if (q - pq == 1) g = del(pq, q, pt, t);
        else g = ins(pq, q, pt, t);
        rs[i] += S[q * T + t];
        rs[i] -= g;
        q = pq; t = pt;
      }
      p_rq[i] = q; p_rt[i] = t;
      state[i] = cr_region_at(t);                          // crcw.h:387 (t may be 0 here)
    }
    for (size_t i = 1; i < n; ++i) filter[i] = false;
    filter[0] = true;
    unsigned accepted = 1;
    const unsigned lim = v_op.back().limit;
    for (size_t i = 1; i < n && accepted < lim; ++i) {
      filter[i] = true;
      for (size_t j = 0; j < i; ++j) {
        if (filter[i] && filter[j] && state[i] == state[j]) {
          float overlap = 0.f;
          float overlap_max = cr_max_overlap * (float)l_sp[j];
          if (p_rq[i] == p_rq[j] && p_rt[i] == p_rt[j]) ++overlap;
          for (int k = t0 - 1; k >= p_rt[i]; --k) {
            if (cr_ali[i][k] > -1 && cr_ali[j][k] > -1 && cr_ali[i][k] == cr_ali[j][k]) {
              ++overlap;
              if (overlap > overlap_max) { filter[i] = false; break; }
            }
          }
        }
      }
      if (filter[i]) ++accepted;
    }
    std::vector<cr_op> tmp;
    accepted = 0;
    for (size_t i = 0; i < n && accepted < lim; ++i)
      if (filter[i]) { tmp.push_back(v_op[i]); ++accepted; }
    tmp.swap(v_op);
    for (size_t i = 1; i < v_op.size(); ++i) v_op[i].limit = std::max(2u, lim / 2);
    int k = v_op[0].k0;
    OrcAli curr((*as)[k]);
    for (size_t i = 0; i < v_op.size(); ++i) {
      int oi = (int)v_op[i].index;
      if (k == (int)as->size()) { as->push_back(curr); (*as)[k].uid = k; }
      (*as)[k].prepend(q0, t0);
      for (int j = t0 - 1; j > p_rt[oi]; --j) {
        int aq = cr_ali[oi][j - 1];
        if (aq > -1) (*as)[k].prepend(aq, j);
      }
      (*as)[k].score = rs[oi];
      v_op[i].q0 = p_rq[oi];
      v_op[i].t0 = p_rt[oi];
      v_op[i].k0 = k;
      if (p_rq[oi] <= end_alignment || p_rt[oi] <= end_alignment) {
        force_opt_path_cr(v_op[i]);
        v_op[i].k0 = -1;
      }
      k = (int)as->size();
    }
  }

  // crcw.h:205-338
  void branch_cr(cr_op& op) {
    unsigned int k_limit = op.limit;
    int q0 = op.q0, t0 = op.t0, k0 = op.k0;
    if (k_limit < 2) { force_opt_path_cr(op); return; }
    if (as->size() > user_limit) { force_opt_path_cr(op); return; }
    if (q0 < 1 || t0 < 1) { err = ORC_E_ARG; return; }     // the reference would index row/column -1 here
    std::vector<cr_op> all_op;
    all_op.reserve(q0 + t0);
    float f, r, g, sum;
    OrcAli curr((*as)[k0]);
    r = curr.score + S[q0 * T + t0];
    f = d(q0 - 1, t0 - 1);
    sum = f + r;
    if (sum > thr) all_op.push_back(cr_op(k_limit, q0 - 1, t0 - 1, k0, sum, r));
    for (int i = t0 - 2; i > 0; --i) {
      f = d(q0 - 1, i);
      g = del(q0 - 1, q0, i, t0);
      sum = f + r - g;
      if (sum > thr) all_op.push_back(cr_op(k_limit, q0 - 1, i, k0, sum, r - g));
    }
    for (int j = q0 - 2; j > 0; --j) {
      f = d(j, t0 - 1);
      g = ins(j, q0, t0 - 1, t0);
      sum = f + r - g;
      if (sum > thr) all_op.push_back(cr_op(k_limit, j, t0 - 1, k0, sum, r - g));
    }
    if (all_op.size() == 0) { force_opt_path_cr(op); return; }
    if (all_op.size() > cr_sort_limit) {
      std::partial_sort(all_op.begin(), all_op.begin() + cr_sort_limit, all_op.end());
      all_op.erase(all_op.begin() + cr_sort_limit, all_op.end());
    } else {
      std::sort(all_op.begin(), all_op.end());
    }
    filter_and_extend_cr(q0, t0, all_op);
    for (std::vector<cr_op>::iterator it = all_op.begin(); it != all_op.end(); ++it)
      if (it->k0 > -1) branch_cr(*it);
  }
};

}  // namespace

extern "C" {

// CRConstrainedNearOptimal::enumerate (crcw.h:134-166).  Parity UNPINNED: crcw.h does not compile on LP64 (:242, min(size_t,
// unsigned)) and its debug operator<< ties it to Troll-dependent types; this restates the source as written, with the one
// out-of-bounds read (regions[-1], :387) given the value it has in practice (see cr_region_at).  *oob_reads (may be NULL)
// receives how often that read would have happened.
int orc_enumerate_cr(int Q, int T, const float* D, const int* PQ, const int* PT, const float* S, const orc_gap* gap,
                     const unsigned char* flags, int number_suboptimal, float delta_ratio, unsigned k_limit, unsigned sort_limit,
                     unsigned user_limit, float max_overlap, orc_set* as, long* oob_reads) {
  if (sort_limit < 1 || Q < 2 || T < 2) return ORC_E_ARG;
  Enum e;
  e.kind = 3; e.Q = Q; e.T = T; e.D = D; e.PQ = PQ; e.PT = PT; e.S = S; e.gap = gap;
  e.flags = flags; e.as = &as->v; e.err = 0;
  e.user_limit = user_limit;
  e.cr_sort_limit = sort_limit; e.cr_max_overlap = max_overlap; e.cr_oob = 0;
  int q_last = Q - 1, t_last = T - 1;
  // init_mem (crcw.h:168-185)
  e.cr_ali.assign(sort_limit, std::vector<int>(t_last, -1));
  e.cr_regions.assign(t_last, 0);
  { int state = 0; for (int i = 0; i < T - 1; ++i) { if ((flags[i + 1] != 0) != (flags[i] != 0)) ++state; e.cr_regions[i] = state; } }
  OrcAli seed;
  seed.uid = 1;                          // crcw.h:147
  as->v.push_back(seed);
  int init = (int)as->v.size() - 1;
  float top = D[q_last * T + t_last];
  float threshold = (1.f - delta_ratio) * top;
  threshold = std::min(threshold, top - 0.1f);
  e.thr = threshold;
  Enum::cr_op op(k_limit, q_last, t_last, init);
  e.branch_cr(op);
  orc_set_sort(as, number_suboptimal);
  if (oob_reads) *oob_reads = e.cr_oob;
  return e.err;
}

// cw.h:68-92 / ucw.h:64-85
int orc_enumerate(int kind, int Q, int T, const float* D, const int* PQ, const int* PT,
                  const float* S, const orc_gap* gap, const unsigned char* flags,
                  int number_suboptimal, float delta_ratio, unsigned user_limit, orc_set* as) {
  Enum e;
  e.kind = kind; e.Q = Q; e.T = T; e.D = D; e.PQ = PQ; e.PT = PT; e.S = S; e.gap = gap;
  e.flags = flags; e.as = &as->v; e.err = 0;
  e.user_limit = user_limit ? user_limit : (kind == 0 ? 1000000u : 100000u);   // cw.h:76 / ucw.h:72
  int q_last = Q - 1, t_last = T - 1;
  OrcAli seed;
  if (kind == 0) seed.uid = 0;          // cw.h:83 sets uid 0; ucw.h:78 leaves -1
  as->v.push_back(seed);
  int k_last = (int)as->v.size() - 1;
  float top = D[q_last * T + t_last];
  float threshold = (1.f - delta_ratio) * top;
  threshold = std::min(threshold, top - 0.1f);
  e.thr = threshold;
  if (kind != 0 && kind != 1) return ORC_E_ARG;            // KSConstrainedNearOptimal has its own entry point
  if (kind == 0) e.branch_cw(q_last, t_last, k_last, false);
  else e.branch_ucw(q_last, t_last, k_last);
  orc_set_sort(as, number_suboptimal);
  return e.err;
}

// KSConstrainedNearOptimal::enumerate (kscw.h:109-136).  Parity UNPINNED: kscw.h does not compile on LP64 (:188) and its debug
// operator<< ties it to Troll-dependent types; this restates the source as written.
int orc_enumerate_ks(int Q, int T, const float* D, const int* PQ, const int* PT, const float* S, const orc_gap* gap,
                     const unsigned char* flags, int number_suboptimal, float delta_ratio, unsigned k_limit, unsigned user_limit,
                     orc_set* as) {
  Enum e;
  e.kind = 2; e.Q = Q; e.T = T; e.D = D; e.PQ = PQ; e.PT = PT; e.S = S; e.gap = gap;
  e.flags = flags; e.as = &as->v; e.err = 0;
  e.user_limit = user_limit;             // NOaliParams::user_limit (default 100000, noalib.cpp:20); not hard-wired here
  int q_last = Q - 1, t_last = T - 1;
  OrcAli seed;
  seed.uid = 1;                          // kscw.h:121
  as->v.push_back(seed);
  int k_last = (int)as->v.size() - 1;
  float top = D[q_last * T + t_last];
  float threshold = (1.f - delta_ratio) * top;
  threshold = std::min(threshold, top - 0.1f);
  Enum::op_data op(k_limit, q_last, t_last, k_last, threshold);
  e.branch_ks(op);
  orc_set_sort(as, number_suboptimal);
  return e.err;
}

// length (without NUL) of every line orc_gapped_strings will write: T + sum of anchors (gstrings.h:84-115)
int orc_gapped_len(const orc_set* as, int T) {
  std::vector<int> anchors(T - 1, 0);
  int gap_total = 0;
  for (size_t a = 0; a < as->v.size(); ++a) {
    const OrcAli& al = as->v[a];
    int n = (int)al.rq.size();
    if (n == 0) continue;
    int pq = al.rq[n - 1], pt = al.rt[n - 1];
    for (int i = n - 2; i >= 0; --i) {
      int cq = al.rq[i], ct = al.rt[i];
      if (cq != pq + 1 && pt >= 0 && pt < T - 1) anchors[pt] = std::max(anchors[pt], cq - pq - 1);
      pq = cq; pt = ct;
    }
  }
  for (int i = 0; i < T - 1; ++i) gap_total += anchors[i];
  return T + gap_total;
}

// gstrings.h:84-164 + gstrings.cpp:17-29
int orc_gapped_strings(const orc_set* as, int Q, int T, const char* qstr, const char* tstr,
                       char* tline, char* qlines, int stride) {
  std::vector<int> anchors(T - 1, 0);
  int gap_total = 0;
  for (size_t a = 0; a < as->v.size(); ++a) {          // buildAnchors :84-115 (mask empty => all)
    const OrcAli& al = as->v[a];
    int n = (int)al.rq.size();
    // list order = reversed storage
    int pq = al.rq[n - 1], pt = al.rt[n - 1];
    for (int i = n - 2; i >= 0; --i) {
      int cq = al.rq[i], ct = al.rt[i];
      if (cq != pq + 1) {
        int gap = cq - pq - 1;
        anchors[pt] = std::max(anchors[pt], gap);
      }
      pq = cq; pt = ct;
    }
  }
  for (int i = 0; i < T - 1; ++i) gap_total += anchors[i];
  {                                                    // gstrings.cpp:17-29
    std::string r;
    for (int i = 0; i < T - 1; ++i) { r.append(1, tstr[i]); r.append(anchors[i], '-'); }
    r.append(1, tstr[T - 1]);
    strcpy(tline, r.c_str());
  }
  std::string seq(qstr);
  for (size_t a = 0; a < as->v.size(); ++a) {          // build(seq, ali, result) :118-164
    const OrcAli& al = as->v[a];
    int n = (int)al.rq.size();
    int it = n - 1;                                     // iterator into list order
    std::string result;
    for (int j = 0; j < T - 1; ++j) {
      int a_gap = anchors[j] + 1;
      if (it >= 0 && al.rt[it] == j) {
        int av = al.rt[it], x = al.rq[it];
        --it;
        if (it < 0) return ORC_E_ARG;                   // reference would dereference end()
        int b = al.rt[it], y = al.rq[it];
        std::string subseq = seq.substr(x, y - x);
        if (!(b - av == 1 || y - x == 1)) {
          for (size_t c = 1; c < subseq.size(); ++c) subseq[c] = (char)tolower(subseq[c]);
        }
        result.append(subseq);
        a_gap -= y - x;
      }
      if (a_gap > 0) result.append(a_gap, '-');
      else if (a_gap < 0) return ORC_E_ARG;             // std::string::append(size_t(-k)) would throw
    }
    int z = T + gap_total - (int)result.size();
    if (z > 1) result.append(z - 1, '-');
    result.append(1, seq[(int)seq.size() - 1]);
    if ((int)result.size() >= stride) return ORC_E_ARG;
    strcpy(qlines + a * stride, result.c_str());
  }
  (void)Q;
  return ORC_OK;
}

}  // extern "C"
