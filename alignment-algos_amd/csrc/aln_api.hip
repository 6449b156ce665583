// aln_api.hip — the extern "C" boundary of libalnhip.so: contexts, resident batches, dispatch, getters.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <thread>

#include "aln_device.h"

using namespace aln;

namespace {

const char* kErrStr(int s) {
  switch (s) {
    case ALN_OK: return "ok";
    case ALN_E_BOUNDS: return "Illegal bounds building DPM";          // dpmatrix.h:361
    case ALN_E_GAPSTYLE: return "Illegal gap style";                   // aasubalib.h:46
    case ALN_E_STARTPAIR: return "Illegal alignment start pair";       // optimal.h:74
    case ALN_E_RESIDUE: return "Residue not in substitution matrix alphabet";
    case ALN_E_ARG: return "Bad argument";
    case ALN_E_HIP: return "HIP runtime error";
    case ALN_E_NOMEM: return "Out of memory";
    case ALN_E_TOO_LONG: return "Sequence too long for this build";
    case ALN_E_NOT_INTEGRAL: return "Scores or gap penalties are not small integers";
    case ALN_E_STATE: return "Call order error (no DP built)";
    case ALN_E_OVERFLOW: return "Output buffer too small";
    default: return "unknown";
  }
}

template <class T>
int dalloc(aln_ctx* ctx, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  ALN_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return ALN_OK;
}

bool valid_align_type(int a) { return a >= 0 && a <= 4; }

}  // namespace

extern "C" {

const char* aln_error_string(int status) { return kErrStr(status); }
const char* aln_last_error(const aln_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
// Does THIS shared object carry a gfx950 code object?  hipcc embeds one offload bundle per --offload-arch whose entry id is
// "hipv4-amdgcn-amd-amdhsa--gfx950": look for that id in the file the function itself was loaded from (works without a GPU).
int aln_has_gfx950(void) {
  static int cached = -1;
  if (cached >= 0) return cached;
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void*>(&aln_has_gfx950), &info) || !info.dli_fname) return cached = 0;
  FILE* f = fopen(info.dli_fname, "rb");
  if (!f) return cached = 0;
  static const char kId[] = "hipv4-amdgcn-amd-amdhsa--gfx950";
  const size_t idn = sizeof kId - 1;
  std::vector<char> buf(1 << 20);
  size_t keep = 0, got;
  int found = 0;
  while (!found && (got = fread(buf.data() + keep, 1, buf.size() - keep, f)) > 0) {
    const size_t n = keep + got;
    for (size_t k = 0; k + idn <= n; ++k)
      if (buf[k] == 'h' && memcmp(buf.data() + k, kId, idn) == 0) { found = 1; break; }
    keep = n < idn ? n : idn - 1;
    memmove(buf.data(), buf.data() + n - keep, keep);
  }
  fclose(f);
  return cached = found;
}

int aln_ctx_create(int device_id, void* stream, aln_ctx** out) {
  if (!out) return ALN_E_ARG;
  *out = nullptr;
  aln_ctx* c = new aln_ctx();
  c->device = device_id;
  c->stream = nullptr;
  c->own_stream = false;
  if (hipSetDevice(device_id) != hipSuccess) { delete c; return ALN_E_HIP; }
  if (stream) c->stream = reinterpret_cast<hipStream_t>(stream);
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ALN_E_HIP; }
    c->own_stream = true;
  }
  aln::hints_from_env(&c->hints);
  *out = c;
  return ALN_OK;
}

void aln_ctx_destroy(aln_ctx* ctx) {
  if (!ctx) return;
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  delete ctx;
}

int aln_ctx_synchronize(aln_ctx* ctx) {
  if (!ctx) return ALN_E_ARG;
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ALN_OK;
}

int aln_batch_create(aln_ctx* ctx, const aln_seqs* queries, const aln_seqs* templates, int32_t n_pairs,
                     const int32_t* q_idx, const int32_t* t_idx, int32_t score_only, aln_batch** out) {
  if (!ctx || !queries || !templates || !out || n_pairs < 0 || (n_pairs > 0 && (!q_idx || !t_idx))) return ALN_E_ARG;
  *out = nullptr;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  aln_batch* b = new aln_batch();
  b->ctx = ctx;
  b->n_pairs = n_pairs;
  b->score_only = score_only != 0;
  b->d_pairs = nullptr; b->d_qcodes = nullptr; b->d_tcodes = nullptr; b->d_H = nullptr; b->d_P = nullptr; b->d_S = nullptr;
  b->d_res = nullptr; b->d_table32 = nullptr; b->d_tablef = nullptr; b->d_tgi = nullptr; b->d_tge = nullptr;
  b->d_path = nullptr; b->d_bounds = nullptr; b->ev0 = nullptr; b->ev1 = nullptr;
  b->have_dp = false; b->have_sub = false; b->islocal = false; b->alpha_n = 0; b->ptr_mode = 0; b->h_mode = 0;
  b->q_offsets.assign(queries->offsets, queries->offsets + queries->n_seqs + 1);
  b->t_offsets.assign(templates->offsets, templates->offsets + templates->n_seqs + 1);
  b->q_total = b->q_offsets.back();
  b->t_total = b->t_offsets.back();
  b->q_res.assign(queries->residues, queries->residues + b->q_total);
  b->t_res.assign(templates->residues, templates->residues + b->t_total);
  b->h_pairs.resize(n_pairs);
  b->maxQ = 0; b->maxT = 0; b->cells = 0;
  int row_align = ctx->hints.plane_row_align;
  if (row_align != 16 && row_align != 32 && row_align != 64) row_align = 8;
  int64_t off = 0;
  for (int p = 0; p < n_pairs; ++p) {
    int qi = q_idx[p], ti = t_idx[p];
    if (qi < 0 || qi >= queries->n_seqs || ti < 0 || ti >= templates->n_seqs) { delete b; return ALN_E_ARG; }
    PairDesc& d = b->h_pairs[p];
    int64_t Q = b->q_offsets[qi + 1] - b->q_offsets[qi], T = b->t_offsets[ti + 1] - b->t_offsets[ti];
    if (Q < 2 || T < 2) { delete b; return ALN_E_ARG; }          // every sequence carries '^' and '$'
    if (Q > kMaxLen || T > kMaxLen) { delete b; return ALN_E_TOO_LONG; }
    d.Q = (int)Q; d.T = (int)T; d.ld = row_stride((int)T, row_align);
    b->maxld = std::max(b->maxld, d.ld);
    d.q_seq = qi; d.t_seq = ti;
    d.q_off = b->q_offsets[qi]; d.t_off = b->t_offsets[ti];
    d.plane_off = off;
    d.q0 = 0; d.q1 = d.Q - 1; d.t0 = 0; d.t1 = d.T - 1;
    if (!b->score_only) off += (int64_t)d.Q * d.ld;
    b->maxQ = std::max(b->maxQ, d.Q); b->maxT = std::max(b->maxT, d.T);
    b->cells += (int64_t)(d.Q - 2) * (d.T - 2);
  }
  b->plane_elems = off;
  b->path_stride = std::min(b->maxQ, b->maxT) + 3;
  int rc;
#define TRY(x) if ((rc = (x)) != ALN_OK) { aln_batch_destroy(b); return rc; }
  TRY(dalloc(ctx, &b->d_pairs, (size_t)n_pairs));
  TRY(dalloc(ctx, &b->d_qcodes, (size_t)b->q_total));
  TRY(dalloc(ctx, &b->d_tcodes, (size_t)b->t_total));
  TRY(dalloc(ctx, &b->d_res, (size_t)n_pairs));
  TRY(dalloc(ctx, &b->d_table32, 32 * 32));
  TRY(dalloc(ctx, &b->d_tablef, 32 * 32));
  if (!b->score_only) {
    TRY(dalloc(ctx, &b->d_H, (size_t)off));
    TRY(dalloc(ctx, &b->d_P, (size_t)off));
    TRY(dalloc(ctx, &b->d_path, (size_t)n_pairs * b->path_stride * 2));
  }
#undef TRY
  if (hipMemcpyAsync(b->d_pairs, b->h_pairs.data(), sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemsetAsync(b->d_res, 0, sizeof(PairResult) * (n_pairs ? n_pairs : 1), ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->last_error = "batch upload failed";
    aln_batch_destroy(b);
    return ALN_E_HIP;
  }
  *out = b;
  return ALN_OK;
}

void aln_batch_destroy(aln_batch* b) {
  if (!b) return;
  hipFree(b->d_pairs); hipFree(b->d_qcodes); hipFree(b->d_tcodes); hipFree(b->d_H); hipFree(b->d_P); hipFree(b->d_S); hipFree(b->d_sabs);
  hipFree(b->d_res); hipFree(b->d_table32); hipFree(b->d_tablef); hipFree(b->d_tgi); hipFree(b->d_tge);
  hipFree(b->d_path); hipFree(b->d_bounds); hipFree(b->d_xscratch); hipFree(b->d_tagq); hipFree(b->d_tagstate); hipFree(b->d_deltabR); hipFree(b->d_pair_deloff);
  aln::free_string_buffers(b);
  if (b->stage_ev) hipEventDestroy(b->stage_ev);
  if (b->h_stage_pin) hipHostFree(b->h_stage_pin);
  for (auto& sc : b->enum_scratch) hipFree(sc.p);
  hipFree(b->d_tcn); hipFree(b->d_deltab); hipFree(b->d_deltab_off); hipFree(b->d_instab);
  for (int k = 0; k < 2; ++k) { if (b->h_slot[k]) hipHostFree(b->h_slot[k]); if (b->slot_ev[k]) hipEventDestroy(b->slot_ev[k]); }
  for (int k = 0; k < aln_batch::kEvRing; ++k) { if (b->ring0[k]) hipEventDestroy(b->ring0[k]); if (b->ring1[k]) hipEventDestroy(b->ring1[k]); }
  delete b;
}

int32_t aln_batch_n_pairs(const aln_batch* b) { return b ? b->n_pairs : 0; }
int64_t aln_batch_cells(const aln_batch* b) { return b ? b->cells : 0; }
static int64_t matrix_cells(const aln_batch* b) {
  int64_t n = 0;
  for (const PairDesc& d : b->h_pairs) n += (int64_t)d.Q * d.T;
  return n;
}
// score element + pointer element of the layout the last build chose (aln_device.h load_score / load_ptr_word)
int32_t aln_batch_plane_bytes_per_cell(const aln_batch* b) { return b ? (b->h_mode ? 2 : 4) + (b->ptr_mode ? 2 : 4) : 0; }
int64_t aln_batch_dp_algorithmic_bytes(const aln_batch* b) { return b ? matrix_cells(b) * aln_batch_plane_bytes_per_cell(b) : 0; }
int64_t aln_batch_dp_contract_bytes(const aln_batch* b) { return b ? matrix_cells(b) * 8 : 0; }   // fp32 score + packed pointer
int64_t aln_batch_device_bytes(const aln_batch* b) {
  if (!b) return 0;
  int64_t n = (int64_t)sizeof(PairDesc) * b->n_pairs + b->q_total + b->t_total + (int64_t)sizeof(PairResult) * b->n_pairs;
  if (!b->score_only) n += b->plane_elems * 8 + (int64_t)b->n_pairs * b->path_stride * 8;
  if (b->d_S) n += b->plane_elems * 4;
  return n;
}
const char* aln_batch_dp_kernel_name(const aln_batch* b) { return b ? b->kernel_name.c_str() : ""; }

}  // extern "C"

namespace {

// residues -> codes under `alphabet`; '^' and '$' get the two sentinel codes whose table rows are zero
int encode(const std::string& res, const int* idx, std::vector<uint8_t>& codes) {
  codes.resize(res.size());
  for (size_t k = 0; k < res.size(); ++k) {
    unsigned char ch = (unsigned char)res[k];
    int c = (ch == '^') ? kCodeHead : (ch == '$') ? kCodeTail : idx[ch];
    if (c < 0) return ALN_E_RESIDUE;
    codes[k] = (uint8_t)c;
  }
  return ALN_OK;
}

int upload_submatrix(aln_batch* b, const aln_submatrix* sub) {
  aln_ctx* ctx = b->ctx;
  if (!sub->alphabet || !sub->table || sub->n < 1 || sub->n > 30) return ALN_E_ARG;
  int idx[256];
  for (int i = 0; i < 256; ++i) idx[i] = -1;
  for (int i = 0; i < sub->n; ++i) idx[(unsigned char)sub->alphabet[i]] = i;
  // staging in PINNED memory kept with the batch: a hipMemcpyAsync from pageable vectors makes the runtime pin them on the fly,
  // which cost ~25 ms per build of 1024 pairs (the whole DP kernel takes 3)
  const size_t nq = b->q_res.size(), nt = b->t_res.size();
  const size_t need = nq + nt + sizeof(float) * 1024 + sizeof(int32_t) * 1024 + 64;
  if (b->h_stage_bytes < need) {
    if (b->h_stage_pin) hipHostFree(b->h_stage_pin);
    b->h_stage_pin = nullptr; b->h_stage_bytes = 0;
    ALN_HIP_CHECK(ctx, hipHostMalloc((void**)&b->h_stage_pin, need, hipHostMallocDefault));
    b->h_stage_bytes = need;
  }
  // an earlier upload may still read the staging buffer: wait for ITS copies (an event behind them), not for the kernels that
  // followed them on the stream — a caller that pipelines builds would stall here for a whole DP kernel
  if (b->stage_ev) ALN_HIP_CHECK(ctx, hipEventSynchronize(b->stage_ev));
  else ALN_HIP_CHECK(ctx, hipEventCreateWithFlags(&b->stage_ev, hipEventDisableTiming));
  uint8_t* qc = b->h_stage_pin;
  uint8_t* tc = qc + nq;
  float* tf = reinterpret_cast<float*>(b->h_stage_pin + ((nq + nt + 15) & ~(size_t)15));
  int32_t* ti = reinterpret_cast<int32_t*>(tf + 1024);
  // residue -> alphabet index, branch-free inside (an unknown residue is found by OR-ing the lookups), a few host threads for the
  // megabytes of a big batch: encoding is on the end-to-end path of every new batch
  uint8_t lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = idx[i] < 0 ? (uint8_t)0x80 : (uint8_t)idx[i];
  lut[(unsigned char)'^'] = (uint8_t)kCodeHead; lut[(unsigned char)'$'] = (uint8_t)kCodeTail;
  auto enc_range = [&lut](const char* src, uint8_t* dst, size_t n) -> int {
    unsigned bad = 0;
    for (size_t k = 0; k < n; ++k) { const uint8_t c = lut[(unsigned char)src[k]]; dst[k] = c; bad |= c; }
    return (bad & 0x80u) ? ALN_E_RESIDUE : ALN_OK;
  };
  auto enc = [&](const std::string& res, uint8_t* codes) -> int {
    const size_t n = res.size();
    const int n_thr = (int)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), 4, n >> 19}));
    if (n_thr == 1) return enc_range(res.data(), codes, n);
    std::vector<int> rcs(n_thr, ALN_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < n_thr; ++k) {
      const size_t lo = n * k / n_thr, hi = n * (k + 1) / n_thr;
      th.emplace_back([&, k, lo, hi]() { rcs[k] = enc_range(res.data() + lo, codes + lo, hi - lo); });
    }
    for (auto& x : th) x.join();
    for (int r : rcs) if (r) return r;
    return ALN_OK;
  };
  int rc = enc(b->q_res, qc);
  if (rc) return rc;
  rc = enc(b->t_res, tc);
  if (rc) return rc;
  for (int i = 0; i < 32 * 32; ++i) { tf[i] = 0.f; ti[i] = 0; }
  for (int i = 0; i < sub->n; ++i)
    for (int j = 0; j < sub->n; ++j) {
      tf[i * 32 + j] = sub->table[i * sub->n + j];
      ti[i * 32 + j] = (int32_t)sub->table[i * sub->n + j];
    }
  b->h_table.assign(sub->table, sub->table + sub->n * sub->n);
  b->alpha_n = sub->n;
  b->alphabet.assign(sub->alphabet, sub->alphabet + sub->n);
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_qcodes, qc, nq, hipMemcpyHostToDevice, ctx->stream));
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tcodes, tc, nt, hipMemcpyHostToDevice, ctx->stream));
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tablef, tf, sizeof(float) * 1024, hipMemcpyHostToDevice, ctx->stream));
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_table32, ti, sizeof(int32_t) * 1024, hipMemcpyHostToDevice, ctx->stream));
  ALN_HIP_CHECK(ctx, hipEventRecord(b->stage_ev, ctx->stream));
  return ALN_OK;             // (the staging buffer lives with the batch: no wait here, the launch follows on the same stream)
}

int upload_simplanes(aln_batch* b, const aln_sim* sim, bool* integral) {
  aln_ctx* ctx = b->ctx;
  if (!sim->planes || !sim->plane_off) return ALN_E_ARG;
  if (!b->d_S) { int rc = dalloc(ctx, &b->d_S, (size_t)b->plane_elems); if (rc) return rc; }
  bool integ = true;
  for (int p = 0; p < b->n_pairs; ++p) {
    const PairDesc& d = b->h_pairs[p];
    const float* src = sim->planes + sim->plane_off[p];
    for (int64_t k = 0; k < (int64_t)d.Q * d.T && integ; ++k) {
      float v = src[k];
      if (!(v == (float)(int)v) || fabsf(v) > 4096.f) integ = false;
    }
    b->sabs_valid = false;                                 // the caller's planes replace what hmap2_apply_kernel described
    ALN_HIP_CHECK(ctx, hipMemcpy2DAsync(b->d_S + d.plane_off, (size_t)d.ld * 4, src, (size_t)d.T * 4, (size_t)d.T * 4, d.Q,
                                        hipMemcpyHostToDevice, ctx->stream));
  }
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *integral = integ;
  return ALN_OK;
}

// everything upload_tgaps will dereference, checked before a single resident table is touched
int check_gap_arrays(const aln_gap* gap) {
  if (gap->model == ALN_GAP_AFFINE_CONST) return ALN_OK;
  if (gap->model != ALN_GAP_TABLES && (!gap->t_gap_init || !gap->t_gap_extn)) return ALN_E_ARG;
  if (gap->model == ALN_GAP_DEL_TABLE_INS_TPOS && !gap->t_gap_cn) return ALN_E_ARG;
  if ((gap->model == ALN_GAP_DEL_TABLE_INS_TPOS || gap->model == ALN_GAP_TABLES) && (!gap->del_table || !gap->del_table_off)) return ALN_E_ARG;
  if (gap->model == ALN_GAP_TABLES && (!gap->ins_tables || !gap->ins_table_off)) return ALN_E_ARG;
  return ALN_OK;
}

int upload_tgaps(aln_batch* b, const aln_gap* gap) {
  aln_ctx* ctx = b->ctx;
  { const int rc = check_gap_arrays(gap); if (rc) return rc; }
  if (!b->d_tgi) { int rc = dalloc(ctx, &b->d_tgi, (size_t)std::max<int64_t>(b->t_total, 1)); if (rc) return rc; }
  if (!b->d_tge) { int rc = dalloc(ctx, &b->d_tge, (size_t)std::max<int64_t>(b->t_total, 1)); if (rc) return rc; }
  if (gap->model != ALN_GAP_TABLES) {             // per-position coefficient arrays (the table model has none)
    if (!gap->t_gap_init || !gap->t_gap_extn) return ALN_E_ARG;
    bool nonneg = true;                           // gaps that grow with the distance: what the exact kernel's chunk skipping needs
    for (int64_t k = 0; k < b->t_total; ++k) nonneg = nonneg && (gap->t_gap_extn[k] >= 0.f);
    b->gap_ext_nonneg = nonneg;
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tgi, gap->t_gap_init, b->t_total * 4, hipMemcpyHostToDevice, ctx->stream));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tge, gap->t_gap_extn, b->t_total * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  if (gap->model == ALN_GAP_DEL_TABLE_INS_TPOS || gap->model == ALN_GAP_TABLES) {
    // Gn2Eval's v_cn and the per-template deletion tables (T x T floats each)
    const bool gn2 = gap->model == ALN_GAP_DEL_TABLE_INS_TPOS;
    if ((gn2 && !gap->t_gap_cn) || !gap->del_table || !gap->del_table_off) return ALN_E_ARG;
    const int ns = (int)b->t_offsets.size() - 1;
    std::vector<int64_t> off(ns);
    int64_t total = 0;
    for (int s = 0; s < ns; ++s) {
      const int64_t T = b->t_offsets[s + 1] - b->t_offsets[s];
      off[s] = total;
      total += T * T;
    }
    if (gn2 && !b->d_tcn) { int rc = dalloc(ctx, &b->d_tcn, (size_t)b->t_total); if (rc) return rc; }
    b->deltabR_valid = false;
    if (b->d_deltabR) { hipFree(b->d_deltabR); b->d_deltabR = nullptr; }
    if (b->d_deltab) { hipFree(b->d_deltab); b->d_deltab = nullptr; }
    if (b->d_deltab_off) { hipFree(b->d_deltab_off); b->d_deltab_off = nullptr; }
    { int rc = dalloc(ctx, &b->d_deltab, (size_t)std::max<int64_t>(total, 1)); if (rc) return rc; }
    ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_deltab_off, (size_t)ns * 8));
    if (gn2) ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tcn, gap->t_gap_cn, b->t_total * 4, hipMemcpyHostToDevice, ctx->stream));
    for (int s = 0; s < ns; ++s) {      // the caller's tables may sit anywhere: one copy per template into the packed pool
      const int64_t T = b->t_offsets[s + 1] - b->t_offsets[s];
      ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_deltab + off[s], gap->del_table + gap->del_table_off[s], (size_t)(T * T) * 4,
                                        hipMemcpyHostToDevice, ctx->stream));
    }
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_deltab_off, off.data(), (size_t)ns * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!gn2) {
      // the evaluator's insertion(), three T x Q planes per pair
      if (!gap->ins_tables || !gap->ins_table_off) return ALN_E_ARG;
      int64_t itotal = 0;
      for (int p = 0; p < b->n_pairs; ++p) { b->h_pairs[p].ins_off = itotal; itotal += (int64_t)3 * b->h_pairs[p].Q * b->h_pairs[p].T; }
      if (b->d_instab) { hipFree(b->d_instab); b->d_instab = nullptr; }
      { int rc = dalloc(ctx, &b->d_instab, (size_t)std::max<int64_t>(itotal, 1)); if (rc) return rc; }
      for (int p = 0; p < b->n_pairs; ++p) {
        const size_t n = (size_t)3 * b->h_pairs[p].Q * b->h_pairs[p].T;
        ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_instab + b->h_pairs[p].ins_off, gap->ins_tables + gap->ins_table_off[p], n * 4,
                                          hipMemcpyHostToDevice, ctx->stream));
      }
      if (b->n_pairs) ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_pairs, b->h_pairs.data(), sizeof(PairDesc) * b->n_pairs, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ALN_OK;
}

int run_dp(aln_batch* b, bool simplane_integral) {
  aln_ctx* ctx = b->ctx;
  if (b->n_pairs == 0) { b->have_dp = true; b->kernel_name = "(empty batch)"; return ALN_OK; }   // nothing to launch
  const bool sub = b->sim_kind == ALN_SIM_SUBMATRIX;
  bool fast = false, tagged = false;
  if (b->algo != ALN_DP_EXACT && !b->have_sub && b->direction == ALN_FWD) {
    fast = fast_path_legal(b, sub ? b->h_table.data() : nullptr, b->alpha_n, &b->gap, simplane_integral);
    tagged = fast && sub && tag_path_legal(b, b->h_table.data(), b->alpha_n, &b->gap) && ctx->hints.tag_kernel;
  }
  if (b->algo == ALN_DP_FAST && !fast) return (b->maxld > 8192 && b->gap.model == ALN_GAP_AFFINE_CONST) ? ALN_E_TOO_LONG : ALN_E_NOT_INTEGRAL;
  // pointer dialect of the P plane (aln_device.h decode_ptr): 0 = packed 32-bit, 1 / 2 = 16-bit tagged words with 11 / 12 tag bits
  b->ptr_mode = tagged ? (tag_path_bits(b, b->h_table.data(), b->alpha_n, &b->gap) == 12 ? 2 : 1) : 0;
  b->h_mode = (tagged && b->islocal && ctx->hints.h16 && tag_h16_legal(b)) ? 1 : 0;   // local scores of the tagged path are integers in [0, 65535]
  {
    const int slot = (int)(b->n_builds % aln_batch::kEvRing);
    if (!b->ring0[slot]) { ALN_HIP_CHECK(ctx, hipEventCreate(&b->ring0[slot])); ALN_HIP_CHECK(ctx, hipEventCreate(&b->ring1[slot])); }
    b->ev0 = b->ring0[slot]; b->ev1 = b->ring1[slot];
    ++b->n_builds;
  }
  ALN_HIP_CHECK(ctx, hipEventRecord(b->ev0, ctx->stream));
  b->tag_segmented = false;
  const bool solo = tagged && ctx->hints.tag_solo && dp_affine_solo_legal(b);
  int rc = solo ? launch_dp_affine_solo(b) : tagged ? launch_dp_affine_tag(b) : fast ? launch_dp_affine_int(b, !sub) : launch_dp_exact(b);
  if (rc) return rc;
  ALN_HIP_CHECK(ctx, hipEventRecord(b->ev1, ctx->stream));
  rc = launch_dp_corner(b);
  if (rc) return rc;
  b->have_dp = true;
  return ALN_OK;
}

}  // namespace

extern "C" {

// everything of aln_batch_dp except the launch: parameters, uploads, similarity planes
static int prepare_dp(aln_batch* b, const aln_sim* sim, const aln_gap* gap, int32_t direction, int32_t algo, int32_t bug_b4, bool* integral_out) {
  if (!b || !sim || !gap) return ALN_E_ARG;
  if (b->score_only) return ALN_E_ARG;
  aln_ctx* ctx = b->ctx;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!valid_align_type(gap->align_type)) return ALN_E_GAPSTYLE;
  if (direction != ALN_FWD && direction != ALN_REV) return ALN_E_ARG;
  if (gap->model != ALN_GAP_AFFINE_CONST && gap->model != ALN_GAP_AFFINE_TPOS_MIN && gap->model != ALN_GAP_DEL_TABLE_INS_TPOS && gap->model != ALN_GAP_TABLES) return ALN_E_ARG;
  b->gap = *gap;
  b->sim_kind = sim->kind;
  b->direction = direction;
  b->algo = algo;
  b->bug_b4 = bug_b4;
  b->islocal = gap->dp_local == 0 ? (gap->align_type == ALN_LOCAL) : (gap->dp_local == 2);   // dpmatrix.h:155
  if (b->have_sub) b->pairs_dirty = true;                           // the device still holds the last sub-rectangles
  b->have_sub = false;
  for (PairDesc& d : b->h_pairs) { d.q0 = 0; d.q1 = d.Q - 1; d.t0 = 0; d.t1 = d.T - 1; }
  b->gapdev.model = gap->model;
  b->gapdev.align_type = gap->align_type;
  b->gapdev.gi = gap->gap_init; b->gapdev.ge = gap->gap_extn;
  b->gapdev.free_del = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_LOCAL_GLOBAL);
  b->gapdev.free_ins = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_GLOBAL_LOCAL);
  int rc;
  bool integral = false;
  b->gap_ext_nonneg = gap->gap_extn >= 0.f;
  if (gap->model != ALN_GAP_AFFINE_CONST) { rc = upload_tgaps(b, gap); if (rc) return rc; }
  if (sim->kind == ALN_SIM_SUBMATRIX) { rc = upload_submatrix(b, &sim->sub); if (rc) return rc; }
  else if (sim->kind == ALN_SIM_MATRIX) { rc = upload_simplanes(b, sim, &integral); if (rc) return rc; }
  else if (sim->kind == ALN_SIM_HMAP2) {
    // Hmap2Eval: SimilarityMatrix + post_process computed on the device into the resident plane, then a plain plane build
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_pairs, b->h_pairs.data(), sizeof(PairDesc) * b->n_pairs, hipMemcpyHostToDevice, ctx->stream));
    b->pairs_dirty = false;
    rc = launch_sim_hmap2(b, sim);
    if (rc) return rc;
    b->sim_kind = ALN_SIM_MATRIX;
  }
  else return ALN_E_ARG;
  b->gap.t_gap_init = nullptr; b->gap.t_gap_extn = nullptr;        // host pointers are not retained
  // the descriptors on the device are the full-rectangle ones the batch was created with unless a sub-rectangle build or a table
  // model rewrote them: an H2D copy from pageable memory stalls a pipelined caller, so it is made only then
  if (b->n_pairs && b->pairs_dirty) {
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_pairs, b->h_pairs.data(), sizeof(PairDesc) * b->n_pairs, hipMemcpyHostToDevice, ctx->stream));
    b->pairs_dirty = false;
  }
  *integral_out = integral;
  return ALN_OK;
}

int aln_batch_dp(aln_batch* b, const aln_sim* sim, const aln_gap* gap, int32_t direction, int32_t algo, int32_t bug_b4) {
  bool integral = false;
  int rc = prepare_dp(b, sim, gap, direction, algo, bug_b4, &integral);
  if (rc) return rc;
  b->simplane_integral = integral;
  return run_dp(b, integral);
}

// New gap parameters for the resident batch: the similarity source (codes + table, resident planes) stays where it is, only
// the gap description is replaced — constants, per-position arrays, deletion / insertion tables are uploaded again.  The next
// aln_batch_reevaluate rebuilds with them.  This is the engine-side half of the reference's refinement rounds
// (gn2.cpp:146-185: enumerate -> templ.updateCore() -> dpm.reevaluate(), where pre_calculate derives new gap tables,
// gn2_eval.cpp:113-158) for callers whose similarity does not change between rounds.
int aln_batch_set_gap(aln_batch* b, const aln_gap* gap) {
  if (!b || !gap) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub) return ALN_E_STATE;
  aln_ctx* ctx = b->ctx;
  ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!valid_align_type(gap->align_type)) return ALN_E_GAPSTYLE;
  if (gap->model != ALN_GAP_AFFINE_CONST && gap->model != ALN_GAP_AFFINE_TPOS_MIN && gap->model != ALN_GAP_DEL_TABLE_INS_TPOS && gap->model != ALN_GAP_TABLES) return ALN_E_ARG;
  { const int rc = check_gap_arrays(gap); if (rc) return rc; }     // a bad description leaves the batch as it was
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));           // nothing may still read the tables that are replaced
  b->gap_ext_nonneg = gap->gap_extn >= 0.f;
  if (gap->model != ALN_GAP_AFFINE_CONST) {
    const int rc = upload_tgaps(b, gap);
    if (rc) {                       // a device error half way: resident tables may be gone — the batch needs a full aln_batch_dp again
      b->have_dp = false;
      return rc;
    }
  }
  // the upload succeeded: commit the description
  b->gap = *gap;
  b->islocal = gap->dp_local == 0 ? (gap->align_type == ALN_LOCAL) : (gap->dp_local == 2);
  b->gapdev.model = gap->model;
  b->gapdev.align_type = gap->align_type;
  b->gapdev.gi = gap->gap_init; b->gapdev.ge = gap->gap_extn;
  b->gapdev.free_del = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_LOCAL_GLOBAL);
  b->gapdev.free_ins = (gap->align_type == ALN_LOCAL || gap->align_type == ALN_SEMI_LOCAL || gap->align_type == ALN_GLOBAL_LOCAL);
  b->gap.t_gap_init = nullptr; b->gap.t_gap_extn = nullptr;        // host pointers are not retained
  return ALN_OK;
}

int aln_batch_reevaluate(aln_batch* b) {
  if (!b) return ALN_E_ARG;
  if (!b->have_dp) return ALN_E_STATE;
  return run_dp(b, b->simplane_integral);      // the resident similarity planes are the ones the first build proved integral (or not)
}

int aln_batch_dp_sub(aln_batch* b, const aln_sim* sim, const aln_gap* gap, int32_t direction, const int32_t* bounds) {
  if (!b || !sim || !gap || !bounds) return ALN_E_ARG;
  // validate first: the reference throws "Illegal bounds building DPM" (dpmatrix.h:360) before touching anything
  for (int p = 0; p < b->n_pairs; ++p) {
    const PairDesc& d = b->h_pairs[p];
    int q0 = bounds[4 * p], t0 = bounds[4 * p + 1], q1 = bounds[4 * p + 2], t1 = bounds[4 * p + 3];
    if (q0 < 0 || t0 < 0 || q1 >= d.Q || t1 >= d.T) return ALN_E_ARG;
    if (q1 <= q0 || t1 <= t0) return ALN_E_BOUNDS;
  }
  // same parameter handling as a full build (without launching one), then narrow every pair to its rectangle and use the
  // exact kernel: a batch of small rectangles is one launch (the reference's SSSS loop fill builds them one by one, ssss.h:621-631)
  bool integral = false;
  int rc = prepare_dp(b, sim, gap, direction, ALN_DP_EXACT, 0, &integral);
  if (rc) return rc;
  b->simplane_integral = integral;
  for (int p = 0; p < b->n_pairs; ++p) {
    PairDesc& d = b->h_pairs[p];
    d.q0 = bounds[4 * p]; d.t0 = bounds[4 * p + 1]; d.q1 = bounds[4 * p + 2]; d.t1 = bounds[4 * p + 3];
  }
  b->have_sub = true;
  b->pairs_dirty = true;
  aln_ctx* ctx = b->ctx;
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_pairs, b->h_pairs.data(), sizeof(PairDesc) * b->n_pairs, hipMemcpyHostToDevice, ctx->stream));
  return run_dp(b, false);
}

// hint exact_debug: far chunks of the last tiled exact build — [0] tested, [1] skipped (far-left deletions), [2] tested, [3] skipped
// (far insertions); the nearest chunk of every scan is always scanned and never tested
int aln_batch_last_exact_stats(const aln_batch* b, uint64_t* out4) {
  if (!b || !out4) return ALN_E_ARG;
  for (int k = 0; k < 4; ++k) out4[k] = b->exact_stats[k];
  return ALN_OK;
}

int aln_batch_last_dp_ms(aln_batch* b, float* ms) {
  if (!b || !ms) return ALN_E_ARG;
  if (!b->have_dp) return ALN_E_STATE;
  if (b->n_pairs == 0) { *ms = 0.f; return ALN_OK; }
  ALN_HIP_CHECK(b->ctx, hipEventSynchronize(b->ev1));
  ALN_HIP_CHECK(b->ctx, hipEventElapsedTime(ms, b->ev0, b->ev1));
  return ALN_OK;
}

int aln_batch_dp_ms_history(aln_batch* b, float* ms, int32_t max_n) {
  if (!b || !ms || max_n < 0) return -1;
  long avail = b->n_builds < aln_batch::kEvRing ? b->n_builds : aln_batch::kEvRing;
  int n = (int)(avail < max_n ? avail : max_n);
  if (b->n_pairs == 0) { for (int k = 0; k < n; ++k) ms[k] = 0.f; return n; }
  for (int k = 0; k < n; ++k) {                       // ms[0] = the latest build, ms[1] the one before ...
    const int slot = (int)((b->n_builds - 1 - k) % aln_batch::kEvRing);
    if (hipEventSynchronize(b->ring1[slot]) != hipSuccess || hipEventElapsedTime(&ms[k], b->ring0[slot], b->ring1[slot]) != hipSuccess) return -1;
  }
  return n;
}

int aln_batch_get_cells(aln_batch* b, int32_t pair, float* score, int32_t* prev_q, int32_t* prev_t) {
  if (!b || pair < 0 || pair >= b->n_pairs) return ALN_E_ARG;
  if (!b->have_dp) return ALN_E_STATE;
  aln_ctx* ctx = b->ctx;
  const PairDesc& d = b->h_pairs[pair];
  const size_t n = (size_t)d.Q * d.ld;
  std::vector<float> h(score ? n : 0);
  std::vector<uint32_t> p((prev_q || prev_t) ? n : 0);
  if (score) {
    const size_t hsz = b->h_mode == 0 ? 4 : 2;
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), reinterpret_cast<const char*>(b->d_H) + (size_t)d.plane_off * hsz, n * hsz,
                                      hipMemcpyDeviceToHost, ctx->stream));
  }
  if (prev_q || prev_t) {
    // mode 1 planes hold 16-bit words at the same element offsets (aln_device.h load_ptr_word)
    const size_t esz = b->ptr_mode == 0 ? 4 : 2;
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(p.data(), reinterpret_cast<const char*>(b->d_P) + (size_t)d.plane_off * esz, n * esz,
                                      hipMemcpyDeviceToHost, ctx->stream));
  }
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < d.Q; ++i)
    for (int j = 0; j < d.T; ++j) {
      const size_t o = (size_t)i * d.T + j;
      if (score) score[o] = load_score(h.data(), 0, d.ld, i, j, b->h_mode);
      if (prev_q || prev_t) {
        int pq, pt;
        decode_ptr(load_ptr_word(p.data(), 0, d.ld, i, j, b->ptr_mode), b->ptr_mode, i, j, pq, pt);
        if (prev_q) prev_q[o] = pq;
        if (prev_t) prev_t[o] = pt;
      }
    }
  return ALN_OK;
}

int aln_batch_get_sim(aln_batch* b, int32_t pair, float* sim) {
  if (!b || !sim || pair < 0 || pair >= b->n_pairs) return ALN_E_ARG;
  if (!b->have_dp) return ALN_E_STATE;
  aln_ctx* ctx = b->ctx;
  const PairDesc& d = b->h_pairs[pair];
  if (b->sim_kind == ALN_SIM_SUBMATRIX) {
    // SimilarityMatrix of AASubstitutionEval (simmatrix.h:51-72, aasubalib.h:17-25) from the retained table
    int idx[256];
    for (int i = 0; i < 256; ++i) idx[i] = -1;
    for (int i = 0; i < b->alpha_n; ++i) idx[(unsigned char)b->alphabet[i]] = i;
    const char* q = b->q_res.data() + d.q_off;
    const char* t = b->t_res.data() + d.t_off;
    for (int i = 0; i < d.Q; ++i)
      for (int j = 0; j < d.T; ++j) {
        float v = 0.f;
        if (i > 0 && j > 0 && i < d.Q - 1 && j < d.T - 1) {
          int a = idx[(unsigned char)q[i]], c = idx[(unsigned char)t[j]];
          if (a >= 0 && c >= 0) v = b->h_table[a * b->alpha_n + c];
        }
        sim[(size_t)i * d.T + j] = v;
      }
    return ALN_OK;
  }
  if (!b->d_S) return ALN_E_STATE;
  ALN_HIP_CHECK(ctx, hipMemcpy2DAsync(sim, (size_t)d.T * 4, b->d_S + d.plane_off, (size_t)d.ld * 4, (size_t)d.T * 4, d.Q,
                                      hipMemcpyDeviceToHost, ctx->stream));
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ALN_OK;
}

int aln_batch_get_corner_scores(aln_batch* b, float* scores) {
  if (!b || !scores) return ALN_E_ARG;
  if (!b->have_dp) return ALN_E_STATE;
  if (b->n_pairs == 0) return ALN_OK;
  std::vector<PairResult> r(b->n_pairs);
  ALN_HIP_CHECK(b->ctx, hipMemcpyAsync(r.data(), b->d_res, sizeof(PairResult) * b->n_pairs, hipMemcpyDeviceToHost, b->ctx->stream));
  ALN_HIP_CHECK(b->ctx, hipStreamSynchronize(b->ctx->stream));
  for (int p = 0; p < b->n_pairs; ++p) scores[p] = r[p].corner;
  return ALN_OK;
}

static int fetch_paths(aln_batch* b, float* scores, int32_t* n, int32_t* pairs, int32_t pair_stride, int32_t* status,
                       bool corner_score, bool flip) {
  aln_ctx* ctx = b->ctx;
  std::vector<PairResult> r(b->n_pairs);
  ALN_HIP_CHECK(ctx, hipMemcpyAsync(r.data(), b->d_res, sizeof(PairResult) * b->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<int32_t> path;
  if (pairs) {
    path.resize((size_t)b->n_pairs * b->path_stride * 2);
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(path.data(), b->d_path, path.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  int rc = ALN_OK;
  for (int p = 0; p < b->n_pairs; ++p) {
    if (scores) scores[p] = corner_score ? r[p].corner : r[p].best;
    if (n) n[p] = r[p].n_path;
    if (status) status[p] = r[p].status;
    if (pairs) {
      int len = r[p].n_path;
      if (len > pair_stride) { len = pair_stride; rc = ALN_E_OVERFLOW; }
      const int32_t* src = path.data() + (size_t)p * b->path_stride * 2;
      int32_t* dst = pairs + (size_t)p * pair_stride * 2;
      for (int k = 0; k < len; ++k) {          // forward builds: device order is end -> start
        const int sk = flip ? (r[p].n_path - 1 - k) : k;
        dst[2 * k] = src[2 * sk];
        dst[2 * k + 1] = src[2 * sk + 1];
      }
    }
  }
  return rc;
}

int aln_batch_optimal(aln_batch* b, float* scores, int32_t* n, int32_t* pairs, int32_t pair_stride, int32_t* status) {
  if (!b) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub) return ALN_E_STATE;
  if (b->n_pairs == 0) return ALN_OK;
  int rc = launch_traceback(b, false);
  if (rc) return rc;
  return fetch_paths(b, scores, n, pairs, pair_stride, status, !b->islocal, b->direction == ALN_FWD);
}

int aln_batch_optimal_enqueue(aln_batch* b) {
  if (!b) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub) return ALN_E_STATE;
  if (b->slot_count == 2) return ALN_E_STATE;
  aln_ctx* ctx = b->ctx;
  const int s = (b->slot_head + b->slot_count) & 1;
  if (b->n_pairs > 0) {
    if (!b->h_slot[s]) {
      ALN_HIP_CHECK(ctx, hipHostMalloc((void**)&b->h_slot[s], sizeof(PairResult) * b->n_pairs));
      ALN_HIP_CHECK(ctx, hipEventCreateWithFlags(&b->slot_ev[s], hipEventDisableTiming));
    }
    int rc = launch_traceback(b, false);
    if (rc) return rc;
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->h_slot[s], b->d_res, sizeof(PairResult) * b->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    ALN_HIP_CHECK(ctx, hipEventRecord(b->slot_ev[s], ctx->stream));
  }
  b->slot_local[s] = b->islocal;
  ++b->slot_count;
  return ALN_OK;
}

int aln_batch_optimal_collect(aln_batch* b, float* scores, int32_t* n, int32_t* status) {
  if (!b) return ALN_E_ARG;
  if (b->slot_count == 0) return ALN_E_STATE;
  const int s = b->slot_head;
  if (b->n_pairs > 0) {
    ALN_HIP_CHECK(b->ctx, hipEventSynchronize(b->slot_ev[s]));
    const PairResult* r = b->h_slot[s];
    for (int p = 0; p < b->n_pairs; ++p) {
      if (scores) scores[p] = b->slot_local[s] ? r[p].best : r[p].corner;
      if (n) n[p] = r[p].n_path;
      if (status) status[p] = r[p].status;
    }
  }
  b->slot_head ^= 1;
  --b->slot_count;
  return ALN_OK;
}

int aln_batch_optimal_subali(aln_batch* b, float* scores, int32_t* n, int32_t* pairs, int32_t pair_stride, int32_t* status) {
  if (!b) return ALN_E_ARG;
  if (!b->have_dp || !b->have_sub) return ALN_E_STATE;
  if (b->n_pairs == 0) return ALN_OK;
  int rc = launch_traceback(b, true);
  if (rc) return rc;
  return fetch_paths(b, scores, n, pairs, pair_stride, status, true, true);
}

// aln_batch_optimal_strings (+ _enqueue / _collect) live in gapped_strings.hip
// aln_batch_enumerate lives in enumerate.hip

}  // extern "C"
