// kmerhash_amd_dist.cpp -- libkmerhash_amd_dist.so: the sharded table (include/kmerhash_amd_dist.h) over RCCL.
//
// Host-only C++: the device work is the kh_* entry points of libkmerhash_amd.so (kh_shard_permute, the streamed insert
// kh_insert_begin/feed/end, kh_find, kh_count, kh_erase) plus RCCL collectives.  The exchange logic is written once
// against a small Transport interface with two implementations: RCCL (one process per GPU, the product) and an in-process
// one (all ranks as threads on one device; what the tests run on a one-GPU box).
#include "../../include/kmerhash_amd_dist.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Transport {
  virtual ~Transport() {}
  virtual int nranks() const = 0;
  virtual int rank() const = 0;
  // every rank gives k values per destination rank (send[dst * k + j]); recv[src * k + j] = what src gave for this rank.
  // ONE collective whatever k is (mxx::all2all, distributed_batched_robinhood_map.hpp:1024)
  virtual bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, hipStream_t stream, std::string& err) = 0;
  // all-to-all-v of `na` arrays that share their split sizes (elements), ONE grouped launch on `stream`, asynchronous
  virtual bool exchange(const void* const* send, void* const* recv, const int* elem_bytes, int na, const uint64_t* send_counts,
                        const uint64_t* recv_counts, hipStream_t stream, std::string& err) = 0;
  virtual bool allreduce_sum(uint64_t* v, hipStream_t stream, std::string& err) = 0;
};

// ---- RCCL ---------------------------------------------------------------------------------------------------------------
struct RcclTransport : Transport {
  ncclComm_t comm;
  int n, r;
  uint64_t* dscratch;       // device: 2 * 64 * 17 u64 (counts in / out)
  uint64_t* hscratch;       // pinned host mirror
  RcclTransport() : comm(nullptr), n(1), r(0), dscratch(nullptr), hscratch(nullptr) {}
  ~RcclTransport() override {
    if (comm) ncclCommDestroy(comm);
    if (dscratch) hipFree(dscratch);
    if (hscratch) hipHostFree(hscratch);
  }
  static const int kMaxCounts = 64 * 17;
  int nranks() const override { return n; }
  int rank() const override { return r; }
  static ncclDataType_t dtype(int bytes) { return bytes == 8 ? ncclUint64 : (bytes == 4 ? ncclUint32 : ncclUint8); }
  bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, hipStream_t stream, std::string& err) override {
    const int m = n * k;
    if (m > kMaxCounts) { err = "too many counts in one exchange"; return false; }
    std::memcpy(hscratch, send, sizeof(uint64_t) * m);
    if (hipMemcpyAsync(dscratch, hscratch, sizeof(uint64_t) * m, hipMemcpyHostToDevice, stream) != hipSuccess) { err = "hipMemcpyAsync(counts)"; return false; }
    ncclResult_t rc = ncclAllToAll(dscratch, dscratch + kMaxCounts, (size_t)k, ncclUint64, comm, stream);
    if (rc != ncclSuccess) { err = std::string("ncclAllToAll(counts): ") + ncclGetErrorString(rc); return false; }
    if (hipMemcpyAsync(hscratch + kMaxCounts, dscratch + kMaxCounts, sizeof(uint64_t) * m, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) { err = "counts read-back"; return false; }
    std::memcpy(recv, hscratch + kMaxCounts, sizeof(uint64_t) * m);
    return true;
  }
  bool exchange(const void* const* send, void* const* recv, const int* elem_bytes, int na, const uint64_t* sc, const uint64_t* rc_,
                hipStream_t stream, std::string& err) override {
    std::vector<size_t> scount(n), sdisp(n), rcount(n), rdisp(n);
    size_t so = 0, ro = 0;
    for (int i = 0; i < n; ++i) { scount[i] = sc[i]; sdisp[i] = so; so += sc[i]; rcount[i] = rc_[i]; rdisp[i] = ro; ro += rc_[i]; }
    // keys and values of a piece in ONE RCCL launch: the all-to-all-v's are grouped (khmxx::distribute_permuted, :1126)
    ncclResult_t rc = ncclGroupStart();
    for (int a = 0; a < na && rc == ncclSuccess; ++a)
      rc = ncclAllToAllv(send[a], scount.data(), sdisp.data(), recv[a], rcount.data(), rdisp.data(), dtype(elem_bytes[a]), comm, stream);
    ncclResult_t re = ncclGroupEnd();
    if (rc == ncclSuccess) rc = re;
    if (rc != ncclSuccess) { err = std::string("ncclAllToAllv: ") + ncclGetErrorString(rc); return false; }
    return true;
  }
  bool allreduce_sum(uint64_t* v, hipStream_t stream, std::string& err) override {
    hscratch[0] = *v;
    if (hipMemcpyAsync(dscratch, hscratch, 8, hipMemcpyHostToDevice, stream) != hipSuccess) { err = "hipMemcpyAsync"; return false; }
    ncclResult_t rc = ncclAllReduce(dscratch, dscratch + 8, 1, ncclUint64, ncclSum, comm, stream);
    if (rc != ncclSuccess) { err = std::string("ncclAllReduce: ") + ncclGetErrorString(rc); return false; }
    if (hipMemcpyAsync(hscratch + 8, dscratch + 8, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
      err = "allreduce read-back"; return false; }
    *v = hscratch[8];
    return true;
  }
};

// ---- all ranks in one process (threads), one device ------------------------------------------------------------------------
struct LocalGroup {
  int n;
  std::mutex mu; std::condition_variable cv; int arrived; unsigned long gen;
  std::vector<const uint64_t*> counts;                 // posted count arrays
  struct Post { const void* const* send; const uint64_t* sc; hipEvent_t ready, done; };
  std::vector<Post> posts;
  std::vector<uint64_t> red;
  explicit LocalGroup(int n_) : n(n_), arrived(0), gen(0), counts(n_), posts(n_), red(n_) {
    for (auto& p : posts) { hipEventCreateWithFlags(&p.ready, hipEventDisableTiming); hipEventCreateWithFlags(&p.done, hipEventDisableTiming); }
  }
  ~LocalGroup() { for (auto& p : posts) { hipEventDestroy(p.ready); hipEventDestroy(p.done); } }
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
struct LocalTransport : Transport {
  std::shared_ptr<LocalGroup> G;
  int r;
  int nranks() const override { return G->n; }
  int rank() const override { return r; }
  bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, hipStream_t, std::string&) override {
    G->counts[r] = send;
    G->barrier();
    for (int src = 0; src < G->n; ++src)
      for (int j = 0; j < k; ++j) recv[src * k + j] = G->counts[src][r * k + j];
    G->barrier();
    return true;
  }
  bool exchange(const void* const* send, void* const* recv, const int* elem_bytes, int na, const uint64_t* sc, const uint64_t* rc,
                hipStream_t stream, std::string& err) override {
    LocalGroup::Post& me = G->posts[r];
    me.send = send; me.sc = sc;
    hipEventRecord(me.ready, stream);                      // what this rank sends was produced on `stream` before this point
    G->barrier();
    bool ok = true;
    uint64_t ro = 0;
    for (int src = 0; src < G->n; ++src) {
      const LocalGroup::Post& p = G->posts[src];
      uint64_t so = 0;
      for (int d = 0; d < r; ++d) so += p.sc[d];
      if (hipStreamWaitEvent(stream, p.ready, 0) != hipSuccess) ok = false;
      for (int a = 0; a < na && rc[src]; ++a)
        if (hipMemcpyAsync(static_cast<char*>(recv[a]) + ro * elem_bytes[a], static_cast<const char*>(p.send[a]) + so * elem_bytes[a],
                           rc[src] * elem_bytes[a], hipMemcpyDeviceToDevice, stream) != hipSuccess) ok = false;
      ro += rc[src];
    }
    hipEventRecord(me.done, stream);
    G->barrier();
    // a sender's buffers stay untouched until every receiver has copied from them
    for (int d = 0; d < G->n; ++d) if (hipStreamWaitEvent(stream, G->posts[d].done, 0) != hipSuccess) ok = false;
    G->barrier();                                           // (the posted pointers may be overwritten by the next exchange)
    if (!ok) err = "local exchange: HIP error";
    return ok;
  }
  bool allreduce_sum(uint64_t* v, hipStream_t, std::string&) override {
    G->red[r] = *v;
    G->barrier();
    uint64_t s = 0;
    for (int i = 0; i < G->n; ++i) s += G->red[i];
    G->barrier();
    *v = s;
    return true;
  }
};

struct Buf {
  void* p; size_t cap;
  Buf() : p(nullptr), cap(0) {}
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) hipFree(p);
    p = nullptr; cap = 0;
    bytes = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    cap = bytes;
    return true;
  }
  void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct Phase { const char* name; hipEvent_t a, b; };

}  // namespace

struct khd_map {
  std::unique_ptr<Transport> tp;
  kh_table* local;
  int device;
  kh_hash dist_hash; uint64_t dist_seed;
  hipStream_t stream, comm;
  // two sets of send / receive buffers: piece i travels while piece i-1 is partitioned
  Buf sk[2], sv[2], rk[2], rv[2], res[2];
  hipEvent_t ev_perm, ev_landed[2], ev_fed[2], ev_sent[2];
  std::vector<Phase> phases;
  std::vector<hipEvent_t> ev_pool;
  std::string err;
};

namespace {

kh_status fail(khd_map* m, kh_status s, const std::string& msg) { if (m) m->err = msg; return s; }
#define KHD_HIP(call) do { if ((call) != hipSuccess) return fail(m, KH_ERR_HIP, #call); } while (0)
#define KHD_KH(call) do { kh_status s__ = (call); if (s__ != KH_OK) return fail(m, s__, std::string(#call) + ": " + kh_last_error(m->local)); } while (0)

hipEvent_t ev_get(khd_map* m) {
  if (!m->ev_pool.empty()) { hipEvent_t e = m->ev_pool.back(); m->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}
struct Span {          // device time of a phase on one stream
  khd_map* m; hipStream_t s; Phase ph;
  Span(khd_map* m_, const char* name, hipStream_t s_) : m(m_), s(s_) { ph.name = name; ph.a = ev_get(m); ph.b = ev_get(m); hipEventRecord(ph.a, s); }
  ~Span() { hipEventRecord(ph.b, s); m->phases.push_back(ph); }
};

kh_status make_map(khd_map** out, std::unique_ptr<Transport> tp, int device, kh_kind kind, kh_hash hash, uint64_t seed, uint64_t capacity,
                   float mn, float mx, kh_hash dist_hash, uint64_t dist_seed) {
  khd_map* m = new khd_map();
  m->tp = std::move(tp); m->local = nullptr; m->device = device; m->dist_hash = dist_hash; m->dist_seed = dist_seed;
  m->stream = nullptr; m->comm = nullptr;
  kh_status s = kh_create(&m->local, kind, 8, 4, hash, seed, capacity, mn, mx, device);
  if (s != KH_OK) { delete m; return s; }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->comm, hipStreamNonBlocking) != hipSuccess) { kh_destroy(m->local); delete m; return KH_ERR_HIP; }
  hipEventCreateWithFlags(&m->ev_perm, hipEventDisableTiming);
  for (int i = 0; i < 2; ++i) {
    hipEventCreateWithFlags(&m->ev_landed[i], hipEventDisableTiming); hipEventCreateWithFlags(&m->ev_fed[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&m->ev_sent[i], hipEventDisableTiming);
  }
  *out = m;
  return KH_OK;
}

// keys (and values) grouped by owner rank into send set `s`; counts[p] on the host.  Synchronises m->stream.
kh_status permute(khd_map* m, const uint64_t* keys, const uint32_t* vals, uint64_t n, int s, uint64_t* counts) {
  const int p = m->tp->nranks();
  if (!m->sk[s].ensure(std::max<uint64_t>(n, 1) * 8) || (vals && !m->sv[s].ensure(std::max<uint64_t>(n, 1) * 4))) return fail(m, KH_ERR_NOMEM, "send buffers");
  kh_status st = kh_shard_permute(m->dist_hash, m->dist_seed, (uint32_t)p, keys, vals, n, static_cast<uint64_t*>(m->sk[s].p),
                                  vals ? static_cast<uint32_t*>(m->sv[s].p) : nullptr, counts, m->device, m->stream);
  if (st != KH_OK) return fail(m, st, "kh_shard_permute");
  return KH_OK;
}

}  // namespace

extern "C" {

kh_status khd_unique_id(void* id128) {
  if (!id128) return KH_ERR_INVALID;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return KH_ERR_HIP;
  static_assert(sizeof(id) == KHD_UNIQUE_ID_BYTES, "communicator id size");
  std::memcpy(id128, &id, sizeof(id));
  return KH_OK;
}

kh_status khd_create(khd_map** out, const void* id128, int nranks, int rank, int device, kh_kind kind, kh_hash hash, uint64_t seed,
                     uint64_t capacity, float mn, float mx, kh_hash dist_hash, uint64_t dist_seed) {
  if (!out || !id128 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return KH_ERR_INVALID;
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) return KH_ERR_HIP;
  std::unique_ptr<RcclTransport> tp(new RcclTransport());
  tp->n = nranks; tp->r = rank;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  if (ncclCommInitRank(&tp->comm, nranks, id, rank) != ncclSuccess) return KH_ERR_HIP;
  if (hipMalloc(reinterpret_cast<void**>(&tp->dscratch), sizeof(uint64_t) * 2 * RcclTransport::kMaxCounts) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&tp->hscratch), sizeof(uint64_t) * 2 * RcclTransport::kMaxCounts) != hipSuccess) return KH_ERR_NOMEM;
  return make_map(out, std::move(tp), device, kind, hash, seed, capacity, mn, mx, dist_hash, dist_seed);
}

kh_status khd_create_local(khd_map** out, int nranks, int device, kh_kind kind, kh_hash hash, uint64_t seed, uint64_t capacity, float mn,
                           float mx, kh_hash dist_hash, uint64_t dist_seed) {
  if (!out || nranks < 1 || nranks > 64) return KH_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return KH_ERR_HIP;
  std::shared_ptr<LocalGroup> G(new LocalGroup(nranks));
  for (int r = 0; r < nranks; ++r) {
    std::unique_ptr<LocalTransport> tp(new LocalTransport());
    tp->G = G; tp->r = r;
    kh_status s = make_map(&out[r], std::move(tp), device, kind, hash, seed, capacity, mn, mx, dist_hash, dist_seed);
    if (s != KH_OK) { for (int i = 0; i < r; ++i) khd_destroy(out[i]); return s; }
  }
  return KH_OK;
}

kh_status khd_destroy(khd_map* m) {
  if (!m) return KH_OK;
  hipSetDevice(m->device);
  hipStreamSynchronize(m->stream); hipStreamSynchronize(m->comm);
  kh_destroy(m->local);
  for (int i = 0; i < 2; ++i) {
    m->sk[i].release(); m->sv[i].release(); m->rk[i].release(); m->rv[i].release(); m->res[i].release();
    hipEventDestroy(m->ev_landed[i]); hipEventDestroy(m->ev_fed[i]); hipEventDestroy(m->ev_sent[i]);
  }
  hipEventDestroy(m->ev_perm);
  for (auto& ph : m->phases) { hipEventDestroy(ph.a); hipEventDestroy(ph.b); }
  for (auto e : m->ev_pool) hipEventDestroy(e);
  hipStreamDestroy(m->comm);
  m->tp.reset();
  delete m;
  return KH_OK;
}

kh_status khd_set_stream(khd_map* m, void* s) {
  if (!m) return KH_ERR_INVALID;
  m->stream = static_cast<hipStream_t>(s);
  return kh_set_stream(m->local, s);
}
const char* khd_last_error(const khd_map* m) { return m ? m->err.c_str() : "null map"; }
kh_table* khd_local(khd_map* m) { return m ? m->local : nullptr; }
int khd_rank(const khd_map* m) { return m ? m->tp->rank() : -1; }
int khd_nranks(const khd_map* m) { return m ? m->tp->nranks() : 0; }

kh_status khd_insert(khd_map* m, const uint64_t* keys, const uint32_t* vals, uint64_t n, int pieces, int reduce_plus, uint64_t* n_inserted) {
  if (!m) return KH_ERR_INVALID;
  if (n_inserted) *n_inserted = 0;
  if ((n && !keys) || (!vals && !reduce_plus) || pieces > 16) return fail(m, KH_ERR_INVALID, "khd_insert: bad arguments");
  KHD_HIP(hipSetDevice(m->device));
  const int p = m->tp->nranks();
  if (p == 1) {       // one rank owns every key
    KHD_KH(reduce_plus ? kh_insert_reduce_plus(m->local, keys, vals, n, KH_MEM_DEVICE, n_inserted) : kh_insert(m->local, keys, vals, n, KH_MEM_DEVICE, n_inserted));
    return KH_OK;
  }
  if (pieces < 1) pieces = 1;
  std::vector<uint64_t> bounds(pieces + 1);
  for (int i = 0; i <= pieces; ++i) bounds[i] = n * (uint64_t)i / (uint64_t)pieces;
  // ---- destination counts of every piece, ONE exchange for all of them: rc[src * pieces + piece].  Up to 8 ranks: one count sweep +
  //      scan + host synchronisation for the whole batch (kh_shard_plan; the pieces are then permuted without counting again,
  //      their boundaries are the plan's: multiples of 4096 pairs); more ranks: a count-only pass per piece
  std::vector<uint64_t> sc((size_t)p * pieces), rc((size_t)p * pieces), tmp(p);
  struct PlanGuard { kh_shard_plan* h = nullptr; ~PlanGuard() { kh_shard_plan_destroy(h); } } plan;
  { Span sp(m, "count_pass", m->stream);
    if (p <= 8) {
      std::vector<uint64_t> pc((size_t)pieces * p);
      kh_status st = kh_shard_plan_create(&plan.h, m->dist_hash, m->dist_seed, KH_XF_IDENTITY, 0, (uint32_t)p, keys, n, (uint32_t)pieces, pc.data(), bounds.data(),
                                          m->device, m->stream);
      if (st != KH_OK) return fail(m, st, "kh_shard_plan_create");
      for (int i = 0; i < pieces; ++i) for (int d = 0; d < p; ++d) sc[(size_t)d * pieces + i] = pc[(size_t)i * p + d];
    } else
    for (int i = 0; i < pieces; ++i) {
      kh_status st = kh_shard_permute(m->dist_hash, m->dist_seed, (uint32_t)p, keys + bounds[i], nullptr, bounds[i + 1] - bounds[i], nullptr, nullptr,
                                      tmp.data(), m->device, m->stream);
      if (st != KH_OK) return fail(m, st, "kh_shard_permute (count only)");
      for (int d = 0; d < p; ++d) sc[(size_t)d * pieces + i] = tmp[d];
    } }
  if (!m->tp->exchange_counts(sc.data(), rc.data(), pieces, m->stream, m->err)) return KH_ERR_HIP;
  uint64_t total = 0, max_piece = 0;
  for (int i = 0; i < pieces; ++i) { uint64_t t = 0; for (int s = 0; s < p; ++s) t += rc[(size_t)s * pieces + i]; total += t; max_piece = std::max(max_piece, t); }
  // every received piece is KEPT (one buffer for the whole batch, piece after piece) until the build has succeeded: the local
  // table may then partition the pieces speculatively (KH_INS_REPEATABLE: no histogram pass, slots shared by all pieces) and
  // ask for them again if that did not hold (KH_ERR_RETRY)
  (void)max_piece;
  if (!m->rk[0].ensure(std::max<uint64_t>(total, 1) * 8) || (vals && !m->rv[0].ensure(std::max<uint64_t>(total, 1) * 4))) return fail(m, KH_ERR_NOMEM, "receive buffers");
  uint64_t* const rk_all = static_cast<uint64_t*>(m->rk[0].p);
  uint32_t* const rv_all = vals ? static_cast<uint32_t*>(m->rv[0].p) : nullptr;
  KHD_KH(kh_insert_begin_ex(m->local, total, (reduce_plus ? KH_INS_REDUCE_PLUS : 0u) | KH_INS_REPEATABLE));
  std::vector<uint64_t> scounts(p), rcounts(p), roff(pieces + 1, 0);
  int landed = -1;
  for (int i = 0; i < pieces; ++i) {
    const int s = i & 1;
    // send set s was last read by the exchange of piece i-2
    KHD_HIP(hipStreamWaitEvent(m->stream, m->ev_sent[s], 0));
    { Span sp(m, "permute", m->stream);
      if (plan.h) {
        const uint64_t np_ = bounds[i + 1] - bounds[i];
        if (!m->sk[s].ensure(std::max<uint64_t>(np_, 1) * 8) || (vals && !m->sv[s].ensure(std::max<uint64_t>(np_, 1) * 4))) return fail(m, KH_ERR_NOMEM, "send buffers");
        kh_status st = kh_shard_plan_permute(plan.h, (uint32_t)i, keys, vals, static_cast<uint64_t*>(m->sk[s].p), vals ? static_cast<uint32_t*>(m->sv[s].p) : nullptr, m->stream);
        if (st != KH_OK) return fail(m, st, "kh_shard_plan_permute");
        for (int d = 0; d < p; ++d) scounts[d] = sc[(size_t)d * pieces + i];
      } else {
        kh_status st = permute(m, keys + bounds[i], vals ? vals + bounds[i] : nullptr, bounds[i + 1] - bounds[i], s, scounts.data());
        if (st != KH_OK) return st;
      } }
    uint64_t rtot = 0;
    for (int src = 0; src < p; ++src) { rcounts[src] = rc[(size_t)src * pieces + i]; rtot += rcounts[src]; }
    roff[i + 1] = roff[i] + rtot;
    KHD_HIP(hipEventRecord(m->ev_perm, m->stream));
    KHD_HIP(hipStreamWaitEvent(m->comm, m->ev_perm, 0));
    { Span sp(m, "exchange", m->comm);
      const void* sb[2] = {m->sk[s].p, m->sv[s].p}; void* rb[2] = {rk_all + roff[i], rv_all ? rv_all + roff[i] : nullptr}; const int eb[2] = {8, 4};
      if (!m->tp->exchange(sb, rb, eb, vals ? 2 : 1, scounts.data(), rcounts.data(), m->comm, m->err)) return KH_ERR_HIP; }
    KHD_HIP(hipEventRecord(m->ev_landed[s], m->comm));
    KHD_HIP(hipEventRecord(m->ev_sent[s], m->comm));
    if (landed >= 0) {      // piece i-1 has landed (or is landing): partition it while piece i travels
      KHD_HIP(hipStreamWaitEvent(m->stream, m->ev_landed[landed], 0));
      { Span sp(m, "feed", m->stream);
        KHD_KH(kh_insert_feed(m->local, rk_all + roff[i - 1], rv_all ? rv_all + roff[i - 1] : nullptr, roff[i] - roff[i - 1], KH_MEM_DEVICE)); }
    }
    landed = s;
  }
  KHD_HIP(hipStreamWaitEvent(m->stream, m->ev_landed[landed], 0));
  { Span sp(m, "feed", m->stream);
    KHD_KH(kh_insert_feed(m->local, rk_all + roff[pieces - 1], rv_all ? rv_all + roff[pieces - 1] : nullptr, roff[pieces] - roff[pieces - 1], KH_MEM_DEVICE)); }
  kh_status est;
  { Span sp(m, "build", m->stream);
    est = kh_insert_end(m->local, n_inserted); }
  if (est == KH_ERR_RETRY) {      // the speculative partition did not hold: the kept pieces, concatenated in feed order, the exact way
    Span sp(m, "refeed", m->stream);
    KHD_KH(kh_insert_begin(m->local, total, reduce_plus));
    KHD_KH(kh_insert_feed(m->local, rk_all, rv_all, total, KH_MEM_DEVICE));
    KHD_KH(kh_insert_end(m->local, n_inserted));
    return KH_OK;
  }
  if (est != KH_OK) return fail(m, est, kh_last_error(m->local));
  return KH_OK;
}

// keys out (grouped by owner), the local query, results back with the swapped counts
static kh_status query(khd_map* m, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals, uint8_t* out_flags, int op, uint64_t* n_local) {
  KHD_HIP(hipSetDevice(m->device));
  const int p = m->tp->nranks();
  if (n_local) *n_local = 0;
  if (p == 1) {
    if (out_keys && n) KHD_HIP(hipMemcpyAsync(out_keys, keys, n * 8, hipMemcpyDeviceToDevice, m->stream));
    Span sp(m, "query", m->stream);
    if (op == 0) KHD_KH(kh_count(m->local, keys, n, KH_MEM_DEVICE, out_flags));
    else if (op == 1) KHD_KH(kh_find(m->local, keys, n, KH_MEM_DEVICE, out_vals, out_flags, nullptr));
    else KHD_KH(kh_erase(m->local, keys, n, KH_MEM_DEVICE, n_local));
    return KH_OK;
  }
  std::vector<uint64_t> sc(p), rc(p);
  { Span sp(m, "permute", m->stream);
    kh_status st = permute(m, keys, nullptr, n, 0, sc.data());
    if (st != KH_OK) return st; }
  if (!m->tp->exchange_counts(sc.data(), rc.data(), 1, m->stream, m->err)) return KH_ERR_HIP;
  uint64_t rtot = 0;
  for (int i = 0; i < p; ++i) rtot += rc[i];
  if (!m->rk[0].ensure(std::max<uint64_t>(rtot, 1) * 8) || !m->res[0].ensure(std::max<uint64_t>(rtot, 1) * 4) || !m->res[1].ensure(std::max<uint64_t>(rtot, 1)))
    return fail(m, KH_ERR_NOMEM, "receive buffers");
  { Span sp(m, "exchange", m->stream);
    const void* sb[1] = {m->sk[0].p}; void* rb[1] = {m->rk[0].p}; const int eb[1] = {8};
    if (!m->tp->exchange(sb, rb, eb, 1, sc.data(), rc.data(), m->stream, m->err)) return KH_ERR_HIP; }
  if (out_keys && n) KHD_HIP(hipMemcpyAsync(out_keys, m->sk[0].p, n * 8, hipMemcpyDeviceToDevice, m->stream));
  uint32_t* lv = static_cast<uint32_t*>(m->res[0].p); uint8_t* lf = static_cast<uint8_t*>(m->res[1].p);
  { Span sp(m, "query", m->stream);
    if (op == 0) KHD_KH(kh_count(m->local, m->rk[0].p, rtot, KH_MEM_DEVICE, lf));
    else if (op == 1) {
      KHD_HIP(hipMemsetAsync(lv, 0, std::max<uint64_t>(rtot, 1) * 4, m->stream));
      KHD_KH(kh_find(m->local, m->rk[0].p, rtot, KH_MEM_DEVICE, lv, lf, nullptr));
    } else { KHD_KH(kh_erase(m->local, m->rk[0].p, rtot, KH_MEM_DEVICE, n_local)); return KH_OK; } }
  { Span sp(m, "exchange", m->stream);       // results return with the swapped counts (:1495), values and flags in one launch
    if (op == 1) {
      const void* sb[2] = {lv, lf}; void* rb[2] = {out_vals, out_flags}; const int eb[2] = {4, 1};
      if (!m->tp->exchange(sb, rb, eb, 2, rc.data(), sc.data(), m->stream, m->err)) return KH_ERR_HIP;
    } else {
      const void* sb[1] = {lf}; void* rb[1] = {out_flags}; const int eb[1] = {1};
      if (!m->tp->exchange(sb, rb, eb, 1, rc.data(), sc.data(), m->stream, m->err)) return KH_ERR_HIP;
    } }
  KHD_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

kh_status khd_count(khd_map* m, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint8_t* out01) {
  if (!m || (n && (!keys || !out01))) return KH_ERR_INVALID;
  return query(m, keys, n, out_keys, nullptr, out01, 0, nullptr);
}
kh_status khd_find(khd_map* m, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals, uint8_t* out_found) {
  if (!m || (n && (!keys || !out_vals || !out_found))) return KH_ERR_INVALID;
  return query(m, keys, n, out_keys, out_vals, out_found, 1, nullptr);
}
kh_status khd_erase(khd_map* m, const uint64_t* keys, uint64_t n, uint64_t* n_erased_local) {
  if (!m || (n && !keys)) return KH_ERR_INVALID;
  return query(m, keys, n, nullptr, nullptr, nullptr, 2, n_erased_local);
}
kh_status khd_size(khd_map* m, uint64_t* global_size) {
  if (!m || !global_size) return KH_ERR_INVALID;
  uint64_t v = 0;
  KHD_KH(kh_size(m->local, &v));
  if (m->tp->nranks() > 1 && !m->tp->allreduce_sum(&v, m->stream, m->err)) return KH_ERR_HIP;
  *global_size = v;
  return KH_OK;
}

kh_status khd_phase_ms(khd_map* m, char* buf, uint64_t cap) {
  if (!m || !buf || !cap) return KH_ERR_INVALID;
  hipSetDevice(m->device);
  hipStreamSynchronize(m->stream); hipStreamSynchronize(m->comm);
  std::vector<std::pair<std::string, double> > acc;
  for (auto& ph : m->phases) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, ph.a, ph.b);
    m->ev_pool.push_back(ph.a); m->ev_pool.push_back(ph.b);
    bool found = false;
    for (auto& a : acc) if (a.first == ph.name) { a.second += ms; found = true; break; }
    if (!found) acc.push_back(std::make_pair(std::string(ph.name), (double)ms));
  }
  m->phases.clear();
  std::string s;
  char line[128];
  for (auto& a : acc) { snprintf(line, sizeof(line), "%s %.6f\n", a.first.c_str(), a.second); s += line; }
  snprintf(buf, cap, "%s", s.c_str());
  return KH_OK;
}

}  // extern "C"
