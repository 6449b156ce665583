// gapped_strings.hip — the display strings and identity of every pair's Optimal alignment, built on the device right after the
// traceback (gfx950), and the pipelined readout aln_batch_optimal_strings_enqueue / _collect.
//
// Reference: AlignedPairList::calcIdentity (alignment.h:856-865), SequenceGaps::buildAnchors / build (gstrings.h:84-164,
// gstrings.cpp:17-29) for a set that holds ONE alignment — what `AlignmentSet as(dpm, optimal); as.assignIdentity();
// cout << FastaOut(len) << as` prints per pair (aa_ali.cpp:83-92).  host_strings.cpp is the general renderer (sets of many
// alignments share gap columns); here the set is the list itself, which makes the layout closed-form:
//
//   a list (q_0,t_0) .. (q_{n-1},t_{n-1}) rises strictly in both indices and ends at the tail pair (Q-1,T-1).  The template line
//   gives every template position one column plus, after t_k, (q_{k+1}-q_k-1) gap columns (the anchors of a one-alignment set).
//   So pair k starts at column  c_k = t_k + (q_k - q_0) - k,  occupies  dq + dt - 1  columns (dq, dt = steps to pair k+1):
//       template:  t[t_k]  '-' x (dq-1)      t[t_k+1 .. t_{k+1}-1]
//       query:     q[q_k .. q_{k+1}-1]       '-' x (dt-1)            (residues after the first lower-cased when dq != 1 and dt != 1)
//   template positions before t_0 show their residue over '-', the last column shows t[T-1] over q[Q-1], and the line length is
//   T + (Q-1-q_0) - (n-1).  Every column is written by exactly one lane: one wave per pair, a lane per list entry; segments
//   wider than 4 columns (the end jumps of a local alignment) are written by the whole wave, 64 columns at a time.
//   identity = (#k with q[q_k] == t[t_k]) - 2 over min(Q,T) - 2: the count is taken here, the fp32 division on the host.
//
// A list that does not rise strictly (an all but empty sequence) is laid out by one lane with the general procedure.
// A list SequenceGaps cannot print (it does not end at the tail pair, or repeats a pair: Optimal_Rev's local lists can do both)
// gives empty lines and length 0, as the host path does.  Only the strings (~(Q+T-n) bytes per line) travel to the host, on the
// context's copy stream, into one of two pinned slots: step k's copy and host work overlap step k+1's kernels.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "aln_device.h"

namespace aln {

struct StrOut {            // per pair, 32 bytes
  float score;
  int32_t status;          // the traceback's status (0 or ALN_E_STARTPAIR / ALN_E_HIP ...)
  int32_t length;          // line length, 0 = no lines
  int32_t same;            // identical aligned residues - 2 (calcIdentity's numerator)
  int32_t err;             // ALN_OK, ALN_E_OVERFLOW (line does not fit the stride) or ALN_E_ARG (a list that does not rise strictly)
  int32_t pad[3];
};

struct StrParams {
  int path_stride;
  int flip;                // forward builds: the traceback wrote the list end -> start
  int corner_score;        // non-local: the corner's score is the alignment's
  int stride;              // chars per line slot
};

__device__ __forceinline__ char lower_ascii(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c | 0x20) : c; }

// A list that does not rise strictly in both indices (an empty template or query: the local list of "^PAWHE$" against "^$" is
// (0,0) (5,0) (6,1)) has no closed-form layout.  One lane then walks host_strings.cpp's general procedure literally
// (insertion_widths + aln_gapped_strings for a set of one): O(T x n) steps, for sequences that are all but empty.
__device__ int render_serial(const int2* __restrict__ pl, int cnt, bool flip, int Q, int T, const char* __restrict__ qs,
                             const char* __restrict__ ts, char* __restrict__ tl, char* __restrict__ ql, int stride, int* len_out) {
  auto entry = [&](int k) { return pl[flip ? cnt - 1 - k : k]; };
  if (Q < 2 || T < 2) return ALN_E_ARG;
  auto width = [&](int j) {                               // longest insertion after template position j
    int w = 0;
    for (int k = 1; k < cnt; ++k) {
      const int2 a = entry(k - 1), b = entry(k);
      const int dq = b.x - a.x;
      if (dq != 1 && a.y == j) w = max(w, dq - 1);
    }
    return w;
  };
  int total = T;
  for (int j = 0; j < T - 1; ++j) total += width(j);
  if (total >= stride) return ALN_E_OVERFLOW;
  int pos = 0;
  for (int j = 0; j < T - 1; ++j) {
    tl[pos++] = ts[j];
    for (int w = width(j); w > 0; --w) tl[pos++] = '-';
  }
  tl[pos++] = ts[T - 1];
  tl[pos] = 0;
  int k = 0, n = 0;                                       // next pair to place, characters of the query line so far
  for (int j = 0; j < T - 1; ++j) {
    int room = width(j) + 1;
    if (k < cnt && entry(k).y == j) {
      if (k + 1 >= cnt) return ALN_E_ARG;
      const int x = entry(k).x, y = entry(k + 1).x;
      const int dt = entry(k + 1).y - j, dq = y - x;
      if (x < 0 || y > Q || dq < 0) return ALN_E_ARG;
      const bool zig = !(dt == 1 || dq == 1);
      for (int c = 0; c < dq; ++c) {
        if (n >= stride - 1) return ALN_E_OVERFLOW;
        ql[n++] = (zig && c > 0) ? lower_ascii(qs[x + c]) : qs[x + c];
      }
      room -= dq;
      ++k;
    }
    if (room < 0) return ALN_E_ARG;
    for (; room > 0; --room) { if (n >= stride - 1) return ALN_E_OVERFLOW; ql[n++] = '-'; }
  }
  for (int rest = total - n; rest > 1; --rest) { if (n >= stride - 1) return ALN_E_OVERFLOW; ql[n++] = '-'; }
  if (n >= stride - 1) return ALN_E_OVERFLOW;
  ql[n++] = qs[Q - 1];
  ql[n] = 0;
  *len_out = total;
  return ALN_OK;
}

__global__ __launch_bounds__(64) void gapped_strings_kernel(const PairDesc* __restrict__ pairs, const PairResult* __restrict__ res,
                                                            const int32_t* __restrict__ path, const char* __restrict__ qchars,
                                                            const char* __restrict__ tchars, char* __restrict__ lines,
                                                            StrOut* __restrict__ out, StrParams prm) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const PairDesc pd = pairs[p];
  const PairResult r = res[p];
  const int Q = pd.Q, T = pd.T, cnt = r.n_path;
  const int2* __restrict__ pl = reinterpret_cast<const int2*>(path + (size_t)p * prm.path_stride * 2);
  char* __restrict__ tl = lines + (size_t)p * 2 * prm.stride;
  char* __restrict__ ql = tl + prm.stride;
  const char* __restrict__ qs = qchars + pd.q_off;
  const char* __restrict__ ts = tchars + pd.t_off;
  auto entry = [&](int k) { return pl[prm.flip ? cnt - 1 - k : k]; };
  StrOut o = {};
  o.score = prm.corner_score ? r.corner : r.best;
  o.status = r.status;
  auto leave = [&]() { if (lane == 0) { tl[0] = 0; ql[0] = 0; out[p] = o; } };
  if (r.status != 0) { leave(); return; }

  // ---- pass 1: identity count, printable?, strictly rising? ------------------------------------------------------------------
  int same = 0;
  bool unprintable = cnt <= 0, crooked = false, outside = false;
  for (int k0 = 0; k0 < cnt; k0 += 64) {
    const int k = k0 + lane;
    if (k < cnt) {
      const int2 a = entry(k);
      const bool inside = a.x >= 0 && a.x < Q && a.y >= 0 && a.y < T;
      if (inside && qs[a.x] == ts[a.y]) ++same;
      if (!inside) outside = true;
      if (k > 0) {
        const int2 b = entry(k - 1);
        if (a.x == b.x && a.y == b.y) unprintable = true;
        else if (a.x - b.x < 1 || a.y - b.y < 1) crooked = true;
      }
      if (k == cnt - 1 && !(a.x == Q - 1 && a.y == T - 1)) unprintable = true;
    }
  }
  for (int off = 32; off; off >>= 1) same += __shfl_xor(same, off);
  o.same = same - 2;                                       // the head and tail pairs always match themselves (alignment.h:859)
  if (__ballot(unprintable) != 0ull) { leave(); return; }
  if (__ballot(outside) != 0ull) { o.err = ALN_E_ARG; leave(); return; }
  if (__ballot(crooked) != 0ull) {                         // the general procedure, one lane
    if (lane == 0) {
      tl[0] = 0; ql[0] = 0;
      int len1 = 0;
      o.err = render_serial(pl, cnt, prm.flip != 0, Q, T, qs, ts, tl, ql, prm.stride, &len1);
      if (o.err == ALN_OK) o.length = len1; else { tl[0] = 0; ql[0] = 0; }
      out[p] = o;
    }
    return;
  }
  const int2 first = entry(0);
  const int len = T + (Q - 1 - first.x) - (cnt - 1);
  if (len >= prm.stride) { o.err = ALN_E_OVERFLOW; leave(); return; }

  // ---- pass 2: the columns ---------------------------------------------------------------------------------------------------
  for (int j = lane; j < first.y; j += 64) { tl[j] = ts[j]; ql[j] = '-'; }        // template residues in front of the list
  auto put = [&](int c, int q0, int t0, int dq, bool zig, int x) {                  // column x of the segment that starts at column c
    if (x < dq) {
      tl[c + x] = x == 0 ? ts[t0] : '-';
      const char ch = qs[q0 + x];
      ql[c + x] = (zig && x > 0) ? lower_ascii(ch) : ch;
    } else {
      tl[c + x] = ts[t0 + (x - dq + 1)];
      ql[c + x] = '-';
    }
  };
  for (int k0 = 0; k0 < cnt - 1; k0 += 64) {
    const int k = k0 + lane;
    const bool act = k < cnt - 1;
    int c = 0, q0 = 0, t0 = 0, dq = 1, dt = 1;
    if (act) {
      const int2 a = entry(k), b = entry(k + 1);
      q0 = a.x; t0 = a.y; dq = b.x - a.x; dt = b.y - a.y;
      c = t0 + (q0 - first.x) - k;
    }
    const int width = dq + dt - 1;
    const bool zig = dq != 1 && dt != 1;
    const bool narrow = width <= 4;
    if (act && narrow)
      for (int x = 0; x < width; ++x) put(c, q0, t0, dq, zig, x);
    unsigned long long wide = __ballot(act && !narrow);
    while (wide) {                                         // the whole wave writes one wide segment
      const int src = __builtin_ctzll(wide);
      wide &= wide - 1;
      const int sc = __shfl(c, src), sq0 = __shfl(q0, src), st0 = __shfl(t0, src), sdq = __shfl(dq, src), sdt = __shfl(dt, src);
      const int sw = sdq + sdt - 1;
      const bool szig = sdq != 1 && sdt != 1;
      for (int x = lane; x < sw; x += 64) put(sc, sq0, st0, sdq, szig, x);
    }
  }
  if (lane == 0) {
    tl[len - 1] = ts[T - 1]; ql[len - 1] = qs[Q - 1];
    tl[len] = 0; ql[len] = 0;
    o.length = len;
    out[p] = o;
  }
}

}  // namespace aln

using namespace aln;

namespace {

int ensure_string_buffers(aln_batch* b, int32_t stride) {
  aln_ctx* ctx = b->ctx;
  const size_t n = (size_t)b->n_pairs;
  if (!ctx->copy_stream) ALN_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!b->d_qchars) {                                     // the residue characters as the caller gave them (codes lose nothing, but the
    const size_t nq = std::max<size_t>(b->q_res.size(), 1), nt = std::max<size_t>(b->t_res.size(), 1);   // planes / profiles paths have none)
    ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_qchars, nq));
    ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_tchars, nt));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_qchars, b->q_res.data(), b->q_res.size(), hipMemcpyHostToDevice, ctx->stream));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->d_tchars, b->t_res.data(), b->t_res.size(), hipMemcpyHostToDevice, ctx->stream));
    ALN_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // (pageable source: done before the caller may touch its strings again)
  }
  if (b->str_stride != stride) {
    if (b->str_count != 0) return ALN_E_STATE;             // slots of another stride are still waiting
    for (int s = 0; s < 2; ++s) {
      if (b->d_str_lines[s]) { hipFree(b->d_str_lines[s]); b->d_str_lines[s] = nullptr; }
      if (b->h_str_lines[s]) { hipHostFree(b->h_str_lines[s]); b->h_str_lines[s] = nullptr; }
    }
    b->str_stride = stride;
  }
  for (int s = 0; s < 2; ++s) {
    if (!b->d_str_lines[s]) ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_str_lines[s], n * 2 * (size_t)stride));
    if (!b->h_str_lines[s]) ALN_HIP_CHECK(ctx, hipHostMalloc((void**)&b->h_str_lines[s], n * 2 * (size_t)stride));
    if (!b->d_str_out[s]) ALN_HIP_CHECK(ctx, hipMalloc((void**)&b->d_str_out[s], n * sizeof(StrOut)));
    if (!b->h_str_out[s]) ALN_HIP_CHECK(ctx, hipHostMalloc((void**)&b->h_str_out[s], n * sizeof(StrOut)));
    if (!b->str_ev[s]) ALN_HIP_CHECK(ctx, hipEventCreateWithFlags(&b->str_ev[s], hipEventDisableTiming));
    if (!b->str_kernel_ev[s]) ALN_HIP_CHECK(ctx, hipEventCreateWithFlags(&b->str_kernel_ev[s], hipEventDisableTiming));
  }
  return ALN_OK;
}

}  // namespace

namespace aln {
void free_string_buffers(aln_batch* b) {
  for (int s = 0; s < 2; ++s) {
    hipFree(b->d_str_lines[s]); hipFree(b->d_str_out[s]);
    if (b->h_str_lines[s]) hipHostFree(b->h_str_lines[s]);
    if (b->h_str_out[s]) hipHostFree(b->h_str_out[s]);
    if (b->str_ev[s]) hipEventDestroy(b->str_ev[s]);
    if (b->str_kernel_ev[s]) hipEventDestroy(b->str_kernel_ev[s]);
  }
  hipFree(b->d_qchars); hipFree(b->d_tchars);
}
}  // namespace aln

extern "C" {

int aln_batch_optimal_strings_enqueue(aln_batch* b, int32_t stride) {
  if (!b || stride < 1) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub) return ALN_E_STATE;
  if (b->str_count == 2) return ALN_E_STATE;
  aln_ctx* ctx = b->ctx;
  if (b->n_pairs > 0) {
    ALN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_string_buffers(b, stride);
    if (rc) return rc;
    const int s = (b->str_head + b->str_count) & 1;
    rc = launch_traceback(b, false);
    if (rc) return rc;
    StrParams prm;
    prm.path_stride = b->path_stride;
    prm.flip = b->direction == ALN_FWD ? 1 : 0;
    prm.corner_score = b->islocal ? 0 : 1;
    prm.stride = stride;
    hipLaunchKernelGGL(gapped_strings_kernel, dim3(b->n_pairs), dim3(64), 0, ctx->stream, b->d_pairs, b->d_res, b->d_path,
                       b->d_qchars, b->d_tchars, b->d_str_lines[s], reinterpret_cast<StrOut*>(b->d_str_out[s]), prm);
    ALN_HIP_CHECK(ctx, hipGetLastError());
    // the copies ride the context's copy stream: the launch stream is free for the next build at once.  The device slot is
    // rewritten two enqueues later at the earliest, and only after its collect (str_count <= 2).
    ALN_HIP_CHECK(ctx, hipEventRecord(b->str_kernel_ev[s], ctx->stream));
    ALN_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->copy_stream, b->str_kernel_ev[s], 0));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->h_str_out[s], b->d_str_out[s], (size_t)b->n_pairs * sizeof(StrOut), hipMemcpyDeviceToHost, ctx->copy_stream));
    ALN_HIP_CHECK(ctx, hipMemcpyAsync(b->h_str_lines[s], b->d_str_lines[s], (size_t)b->n_pairs * 2 * (size_t)stride, hipMemcpyDeviceToHost, ctx->copy_stream));
    ALN_HIP_CHECK(ctx, hipEventRecord(b->str_ev[s], ctx->copy_stream));
  }
  ++b->str_count;
  return ALN_OK;
}

int aln_batch_optimal_strings_collect(aln_batch* b, float* scores, float* identity, int32_t* status, char* tlines, char* qlines,
                                      int32_t stride, int32_t* lengths) {
  if (!b || !tlines || !qlines || stride < 1) return ALN_E_ARG;
  if (b->str_count == 0) return ALN_E_STATE;
  if (b->n_pairs > 0 && stride != b->str_stride) return ALN_E_ARG;
  const int s = b->str_head;
  int worst = ALN_OK;
  if (b->n_pairs > 0) {
    ALN_HIP_CHECK(b->ctx, hipEventSynchronize(b->str_ev[s]));
    const StrOut* o = reinterpret_cast<const StrOut*>(b->h_str_out[s]);
    const char* src = b->h_str_lines[s];
    const int n = b->n_pairs;
    auto work = [&](int p0, int p1) {                      // lines of different pairs are independent: a few host threads share the copies
      for (int p = p0; p < p1; ++p) {
        const PairDesc& d = b->h_pairs[p];
        char* tl = tlines + (size_t)p * stride;
        char* ql = qlines + (size_t)p * stride;
        if (scores) scores[p] = o[p].score;
        if (status) status[p] = o[p].status;
        if (lengths) lengths[p] = o[p].length;
        if (identity) identity[p] = o[p].status == 0 ? float(o[p].same) / float(std::min(d.Q, d.T) - 2) * 100.f : 0.f;   // alignment.h:864
        if (o[p].length > 0) {
          memcpy(tl, src + (size_t)p * 2 * stride, (size_t)o[p].length + 1);
          memcpy(ql, src + ((size_t)p * 2 + 1) * stride, (size_t)o[p].length + 1);
        } else { tl[0] = 0; ql[0] = 0; }
      }
    };
    const int n_thr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 4, n / 256}));
    if (n_thr == 1) work(0, n);
    else {
      std::vector<std::thread> th;
      for (int k = 0; k < n_thr; ++k) th.emplace_back(work, (int)((long)n * k / n_thr), (int)((long)n * (k + 1) / n_thr));
      for (auto& x : th) x.join();
    }
    for (int p = 0; p < n; ++p) {
      const int e = o[p].status != 0 ? o[p].status : o[p].err;
      if (e == ALN_E_OVERFLOW) worst = e;
      else if (e != ALN_OK && worst == ALN_OK) worst = e;
    }
  }
  b->str_head ^= 1;
  --b->str_count;
  return worst;
}

// Optimal + assignIdentity + SequenceGaps for every pair of the batch: the strings a driver prints for
// `AlignmentSet alignments(dpm, optimal); alignments.assignIdentity(); cout << FastaOut(len) << alignments`
// (aa_ali.cpp:83-92, fastaio.h:51-76, gstrings.h:84-164): enqueue + collect in one call.
int aln_batch_optimal_strings(aln_batch* b, float* scores, float* identity, int32_t* status, char* tlines, char* qlines,
                              int32_t stride, int32_t* lengths) {
  if (!b || !tlines || !qlines || stride < 1) return ALN_E_ARG;
  if (!b->have_dp || b->have_sub) return ALN_E_STATE;
  if (b->n_pairs == 0) return ALN_OK;
  if (b->str_count != 0) return ALN_E_STATE;               // enqueued slots must be collected first
  int rc = aln_batch_optimal_strings_enqueue(b, stride);
  if (rc) return rc;
  return aln_batch_optimal_strings_collect(b, scores, identity, status, tlines, qlines, stride, lengths);
}

}  // extern "C"
