// aln_comm.hip — the multi-GPU side of the C ABI: a length-sorted deal of the pair list over ranks and the ONE collective
// of the path, an all-gather of the per-pair scores over RCCL (xGMI).
//
// Pairs are independent DPMatrix objects (dpmatrix.h:104-111), so the data path needs no collective: every rank builds
// its own pairs.  SURVEY 8(e): sort by Q*T descending, deal the sorted list over the devices, gather the fp32 scores once.
// The payload is tiny (config 2: 4 KB per rank; config 5: 8 MiB per rank), so the collective is latency-bound and every
// peer uses its own direct xGMI link; nothing here is sized for a ring.
//
// Two ways to form a communicator, both through aln_comm_create:
//   * one process per GPU (torchrun / MPI): rank 0 calls aln_comm_unique_id, ships the 128 bytes to the others by whatever
//     transport the job has, every process passes its ONE context and its rank;
//   * one process driving several GPUs (the reference's C++11 host is single-threaded): pass all contexts; id may be NULL
//     when the process holds every rank.
// librccl.so is loaded on first use (dlopen), so single-GPU users never pay for it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "aln_internal.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  if (r.so || !r.err.empty()) return &r;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.so) break;
  }
  if (!r.so) { r.err = std::string("dlopen librccl.so: ") + dlerror(); return &r; }
#define ALN_SYM(field, sym) \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, sym)); \
  if (!r.field) { r.err = std::string("dlsym ") + sym; r.so = nullptr; return &r; }
  ALN_SYM(GetUniqueId, "ncclGetUniqueId");
  ALN_SYM(CommInitRank, "ncclCommInitRank");
  ALN_SYM(CommDestroy, "ncclCommDestroy");
  ALN_SYM(AllGather, "ncclAllGather");
  ALN_SYM(GroupStart, "ncclGroupStart");
  ALN_SYM(GroupEnd, "ncclGroupEnd");
  ALN_SYM(GetErrorString, "ncclGetErrorString");
#undef ALN_SYM
  return &r;
}

struct Rec { int32_t index; float score; };   // one gathered element: global pair index (-1 = padding) + its score

}  // namespace

struct aln_comm {
  int n_ranks = 0, first_rank = 0;
  std::vector<aln_ctx*> ctxs;          // this process's contexts = ranks first_rank .. first_rank + n_local - 1
  std::vector<ncclComm_t> comms;
  std::vector<Rec*> d_send, d_recv;    // per local rank, grown on demand
  std::vector<size_t> cap;             // records per rank the buffers hold
  Rec* h_stage = nullptr; size_t h_cap = 0;   // pinned staging (send records of all local ranks, then the gathered result)
  std::string last_error;
};

extern "C" {

int aln_comm_unique_id(void* id_out) {
  if (!id_out) return ALN_E_ARG;
  Rccl* r = rccl();
  if (!r->so) return ALN_E_HIP;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return ALN_E_HIP;
  static_assert(sizeof(ncclUniqueId) == ALN_COMM_ID_BYTES, "ALN_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  memcpy(id_out, &id, sizeof id);
  return ALN_OK;
}

void aln_comm_destroy(aln_comm* c) {
  if (!c) return;
  Rccl* r = rccl();
  for (size_t k = 0; k < c->comms.size(); ++k) {
    if (c->ctxs[k]) (void)hipSetDevice(c->ctxs[k]->device);
    if (c->comms[k] && r->so) r->CommDestroy(c->comms[k]);
    if (k < c->d_send.size()) { hipFree(c->d_send[k]); hipFree(c->d_recv[k]); }
  }
  if (c->h_stage) hipHostFree(c->h_stage);
  delete c;
}

const char* aln_comm_last_error(const aln_comm* c) {
  if (c) return c->last_error.c_str();
  return rccl()->err.c_str();
}

int aln_comm_create(aln_ctx* const* ctxs, int32_t n_local, const void* id128, int32_t n_ranks, int32_t first_rank, aln_comm** out) {
  if (!ctxs || !out || n_local < 1 || n_ranks < n_local || first_rank < 0 || first_rank + n_local > n_ranks) return ALN_E_ARG;
  *out = nullptr;
  for (int k = 0; k < n_local; ++k) if (!ctxs[k]) return ALN_E_ARG;
  if (!id128 && n_local != n_ranks) return ALN_E_ARG;          // the id may be omitted only by a process that holds every rank
  Rccl* r = rccl();
  if (!r->so) return ALN_E_HIP;
  ncclUniqueId id;
  if (id128) memcpy(&id, id128, sizeof id);
  else if (r->GetUniqueId(&id) != ncclSuccess) return ALN_E_HIP;
  aln_comm* c = new aln_comm();
  c->n_ranks = n_ranks; c->first_rank = first_rank;
  c->ctxs.assign(ctxs, ctxs + n_local);
  c->comms.assign(n_local, nullptr);
  c->d_send.assign(n_local, nullptr); c->d_recv.assign(n_local, nullptr); c->cap.assign(n_local, 0);
  ncclResult_t rc = r->GroupStart();
  for (int k = 0; k < n_local && rc == ncclSuccess; ++k) {
    if (hipSetDevice(ctxs[k]->device) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    rc = r->CommInitRank(&c->comms[k], n_ranks, id, first_rank + k);
  }
  const ncclResult_t rc2 = r->GroupEnd();
  if (rc == ncclSuccess) rc = rc2;
  if (rc != ncclSuccess) {
    ctxs[0]->last_error = std::string("ncclCommInitRank: ") + r->GetErrorString(rc);
    aln_comm_destroy(c);
    return ALN_E_HIP;
  }
  *out = c;
  return ALN_OK;
}

int32_t aln_comm_n_ranks(const aln_comm* c) { return c ? c->n_ranks : 0; }

// SURVEY 8(b)'s aln_ctx_create(device_ids, n): one context (private stream) per device of this process + their communicator.
int aln_ctx_create_multi(const int32_t* device_ids, int32_t n, aln_ctx** ctxs_out, aln_comm** comm_out) {
  if (!device_ids || n < 1 || !ctxs_out) return ALN_E_ARG;
  for (int k = 0; k < n; ++k) ctxs_out[k] = nullptr;
  int rc = ALN_OK;
  for (int k = 0; k < n && rc == ALN_OK; ++k) rc = aln_ctx_create(device_ids[k], nullptr, &ctxs_out[k]);
  if (rc == ALN_OK && comm_out) rc = aln_comm_create(ctxs_out, n, nullptr, n, 0, comm_out);
  if (rc != ALN_OK) for (int k = 0; k < n; ++k) { if (ctxs_out[k]) aln_ctx_destroy(ctxs_out[k]); ctxs_out[k] = nullptr; }
  return rc;
}

// The one collective.  Local rank k of this process contributes n_local[k] scores whose positions in the job's global pair
// list are global_index[k][0..n_local[k]).  Every rank calls it with the same n_max (>= every rank's n_local) and n_total;
// afterwards global_out[0..n_total) holds every pair's score on every rank (positions nobody contributed keep their value).
int aln_gather_scores(aln_comm* c, const float* const* local_scores, const int32_t* const* global_index, const int32_t* n_local,
                      int32_t n_max, float* global_out, int64_t n_total) {
  if (!c || !local_scores || !global_index || !n_local || !global_out || n_max < 0 || n_total < 0) return ALN_E_ARG;
  const int nl = (int)c->ctxs.size();
  for (int k = 0; k < nl; ++k) {
    if (n_local[k] < 0 || n_local[k] > n_max) return ALN_E_ARG;
    if (n_local[k] > 0 && (!local_scores[k] || !global_index[k])) return ALN_E_ARG;
  }
  if (n_max == 0) return ALN_OK;
  Rccl* r = rccl();
  const size_t per = (size_t)n_max, all = per * (size_t)c->n_ranks;
#define CTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->last_error = std::string(#expr) + ": " + hipGetErrorString(e_); return ALN_E_HIP; } } while (0)
  const size_t need_h = std::max(per * nl, all);
  if (c->h_cap < need_h) {
    if (c->h_stage) { hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_cap = 0; }
    CTRY(hipHostMalloc((void**)&c->h_stage, need_h * sizeof(Rec)));
    c->h_cap = need_h;
  }
  for (int k = 0; k < nl; ++k) {
    Rec* s = c->h_stage + per * k;
    for (int e = 0; e < n_local[k]; ++e) { s[e].index = global_index[k][e]; s[e].score = local_scores[k][e]; }
    for (int e = n_local[k]; e < n_max; ++e) { s[e].index = -1; s[e].score = 0.f; }
  }
  for (int k = 0; k < nl; ++k) {
    CTRY(hipSetDevice(c->ctxs[k]->device));
    if (c->cap[k] < per) {
      hipFree(c->d_send[k]); hipFree(c->d_recv[k]); c->d_send[k] = c->d_recv[k] = nullptr; c->cap[k] = 0;
      CTRY(hipMalloc((void**)&c->d_send[k], per * sizeof(Rec)));
      CTRY(hipMalloc((void**)&c->d_recv[k], all * sizeof(Rec)));
      c->cap[k] = per;
    }
    CTRY(hipMemcpyAsync(c->d_send[k], c->h_stage + per * k, per * sizeof(Rec), hipMemcpyHostToDevice, c->ctxs[k]->stream));
  }
  ncclResult_t rc = r->GroupStart();
  for (int k = 0; k < nl && rc == ncclSuccess; ++k) {
    if (hipSetDevice(c->ctxs[k]->device) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    rc = r->AllGather(c->d_send[k], c->d_recv[k], per * sizeof(Rec), ncclChar, c->comms[k], c->ctxs[k]->stream);
  }
  const ncclResult_t rc2 = r->GroupEnd();
  if (rc == ncclSuccess) rc = rc2;
  if (rc != ncclSuccess) { c->last_error = std::string("ncclAllGather: ") + r->GetErrorString(rc); return ALN_E_HIP; }
  // every local device now holds the whole list; one device-to-host copy (from the first) feeds the caller's array
  CTRY(hipSetDevice(c->ctxs[0]->device));
  for (int k = 1; k < nl; ++k) { CTRY(hipSetDevice(c->ctxs[k]->device)); CTRY(hipStreamSynchronize(c->ctxs[k]->stream)); }
  CTRY(hipSetDevice(c->ctxs[0]->device));
  CTRY(hipMemcpyAsync(c->h_stage, c->d_recv[0], all * sizeof(Rec), hipMemcpyDeviceToHost, c->ctxs[0]->stream));
  CTRY(hipStreamSynchronize(c->ctxs[0]->stream));
#undef CTRY
  int bad = 0;
  for (size_t e = 0; e < all; ++e) {
    const Rec& x = c->h_stage[e];
    if (x.index < 0) continue;
    if ((int64_t)x.index >= n_total) { bad = 1; continue; }
    global_out[x.index] = x.score;
  }
  return bad ? ALN_E_ARG : ALN_OK;
}

// Length-sorted deal: units (pairs, or query rows of an all-vs-all job) sorted by work = Q*T descending (ties: lower
// index first) and dealt over the ranks in boustrophedon order 0..n-1, n-1..0, ... so that every rank gets one unit of every
// "stratum" and the per-rank sums of work differ by at most one unit's work.  owner[u] = rank of unit u; slot[u] = its position
// in that rank's local list (local lists keep the sorted order: longest first, which also launches the long pairs first).
// Pure host arithmetic, no device.
int aln_deal_units(const int64_t* work, int64_t n_units, int32_t n_ranks, int32_t* owner, int32_t* slot) {
  if ((!work && n_units > 0) || n_units < 0 || n_ranks < 1 || !owner) return ALN_E_ARG;
  std::vector<int64_t> order((size_t)n_units);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return work[a] > work[b]; });
  std::vector<int32_t> count((size_t)n_ranks, 0);
  for (int64_t k = 0; k < n_units; ++k) {
    const int64_t round = k / n_ranks, pos = k % n_ranks;
    const int32_t rk = (int32_t)((round & 1) ? (n_ranks - 1 - pos) : pos);
    owner[order[(size_t)k]] = rk;
    if (slot) slot[order[(size_t)k]] = count[(size_t)rk];
    ++count[(size_t)rk];
  }
  return ALN_OK;
}

}  // extern "C"
