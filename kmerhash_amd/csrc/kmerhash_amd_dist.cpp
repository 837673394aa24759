// kmerhash_amd_dist.cpp -- libkmerhash_amd_dist.so: the sharded table (include/kmerhash_amd_dist.h) over RCCL.
//
// Host-only C++: the device work is the kh_* entry points of libkmerhash_amd.so (kh_shard_plan_*, the streamed insert
// kh_insert_begin/feed/end, kh_find, kh_count, kh_erase) plus RCCL collectives.  The exchange logic is written once
// against a small Transport interface with two implementations: RCCL (one process per GPU, the product) and an in-process
// one (all ranks as threads on one device; what the tests run on a one-GPU box).
//
// Failure protocol (a rank that fails locally must not leave its peers inside a collective):
//   * every collective entry point runs the SAME sequence of collectives on every rank whatever happens locally;
//   * a local failure (allocation, a kh_* call) is only NOTED; the rank goes on taking part in the collectives that are still
//     to come (its buffers exist: they are allocated before the vote that precedes their use) and skips its local work;
//   * votes make the failure known to all ranks: the status word that travels with the count exchange (stage 1: everything before
//     any payload is sized), an all-reduce(max) before the first payload exchange (stage 2: the receive side), one after the last
//     payload exchange of insert / erase (stage 3: nobody builds or erases with what a failed rank sent) and one at the end
//     (stage 4: the build / the local erase).  find / count stay asynchronous: the status word of their local work rides with the last result
//     exchange and is reported by khd_synchronize or by the next collective call ("late status");
//   * a failure of the transport itself (RCCL error, a peer that never arrives: bounded wait, KHD_OPT_TIMEOUT_MS) ends the map:
//     the communicator is aborted and every later call returns an error.
#include "../../include/kmerhash_amd_dist.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one array of an all-to-all-v: counts and displacements in ELEMENTS; sd / rd == nullptr: exclusive prefix sums of sc / rc
struct XArr { const void* send; void* recv; int elem_bytes; const uint64_t *sc, *sd, *rc, *rd; };

struct Transport {
  virtual ~Transport() {}
  virtual int nranks() const = 0;
  virtual int rank() const = 0;
  // every rank gives k values per destination rank (send[dst * k + j]) and ONE status word; recv[src * k + j] = what src gave for
  // this rank, *worst = the largest status word of all ranks.  ONE collective whatever k is (mxx::all2all,
  // distributed_batched_robinhood_map.hpp:1024).  Synchronises `stream`.
  virtual bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, uint64_t status, uint64_t* worst, hipStream_t stream, std::string& err) = 0;
  // all-to-all-v of `na` arrays, ONE grouped launch on `stream`, asynchronous (khmxx::distribute_permuted, :1126)
  virtual bool exchange(const XArr* a, int na, hipStream_t stream, std::string& err) = 0;
  // v[0..n): max (is_max) or sum over the ranks of host values.  Synchronises `stream`.
  virtual bool allreduce(uint64_t* v, int n, bool is_max, hipStream_t stream, std::string& err) = 0;
  // bounded wait for everything queued on `stream` (a collective in it may be waiting for a peer that never arrives)
  virtual bool wait(hipStream_t stream, std::string& err) = 0;
  virtual void abandon() = 0;          // this rank leaves for good: nobody may wait for it any more
  long long timeout_ms = 300000;
};

// ---- RCCL ---------------------------------------------------------------------------------------------------------------
struct RcclTransport : Transport {
  ncclComm_t comm;
  int n, r;
  uint64_t* dscratch;       // device: 2 * kMaxCounts u64 (counts in / out)
  uint64_t* hscratch;       // pinned host mirror
  RcclTransport() : comm(nullptr), n(1), r(0), dscratch(nullptr), hscratch(nullptr) {}
  ~RcclTransport() override {
    if (comm) ncclCommDestroy(comm);
    if (dscratch) hipFree(dscratch);
    if (hscratch) hipHostFree(hscratch);
  }
  static const int kMaxCounts = 64 * 18;
  int nranks() const override { return n; }
  int rank() const override { return r; }
  static ncclDataType_t dtype(int bytes) { return bytes == 8 ? ncclUint64 : (bytes == 4 ? ncclUint32 : ncclUint8); }
  void abandon() override { if (comm) { ncclCommAbort(comm); comm = nullptr; } }
  bool wait(hipStream_t stream, std::string& err) override {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
      const hipError_t e = hipStreamQuery(stream);
      if (e == hipSuccess) return true;
      if (e != hipErrorNotReady) { (void)hipGetLastError(); err = std::string("hipStreamQuery: ") + hipGetErrorString(e); return false; }
      if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spins & 1023u) == 1023u) {
        if (comm) { ncclResult_t ae = ncclSuccess; if (ncclCommGetAsyncError(comm, &ae) == ncclSuccess && ae != ncclSuccess) { err = std::string("RCCL asynchronous error: ") + ncclGetErrorString(ae); return false; } }
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > timeout_ms) { err = "timed out waiting for a collective (a peer rank never arrived); communicator aborted"; return false; }
      }
    }
  }
  bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, uint64_t status, uint64_t* worst, hipStream_t stream, std::string& err) override {
    const int k1 = k + 1, m = n * k1;
    if (!comm) { err = "communicator aborted"; return false; }
    if (m > kMaxCounts) { err = "too many counts in one exchange"; return false; }
    for (int d = 0; d < n; ++d) { std::memcpy(hscratch + (size_t)d * k1, send + (size_t)d * k, sizeof(uint64_t) * k); hscratch[(size_t)d * k1 + k] = status; }
    if (hipMemcpyAsync(dscratch, hscratch, sizeof(uint64_t) * m, hipMemcpyHostToDevice, stream) != hipSuccess) { err = "hipMemcpyAsync(counts)"; return false; }
    ncclResult_t rc = ncclAllToAll(dscratch, dscratch + kMaxCounts, (size_t)k1, ncclUint64, comm, stream);
    if (rc != ncclSuccess) { err = std::string("ncclAllToAll(counts): ") + ncclGetErrorString(rc); return false; }
    if (hipMemcpyAsync(hscratch + kMaxCounts, dscratch + kMaxCounts, sizeof(uint64_t) * m, hipMemcpyDeviceToHost, stream) != hipSuccess) { err = "counts read-back"; return false; }
    if (!wait(stream, err)) return false;
    uint64_t w = 0;
    for (int s = 0; s < n; ++s) { std::memcpy(recv + (size_t)s * k, hscratch + kMaxCounts + (size_t)s * k1, sizeof(uint64_t) * k); w = std::max(w, hscratch[kMaxCounts + (size_t)s * k1 + k]); }
    *worst = w;
    return true;
  }
  bool exchange(const XArr* a, int na, hipStream_t stream, std::string& err) override {
    if (!comm) { err = "communicator aborted"; return false; }
    std::vector<size_t> v((size_t)4 * n * na);
    for (int k = 0; k < na; ++k) {
      size_t *scount = &v[(size_t)4 * n * k], *sdisp = scount + n, *rcount = sdisp + n, *rdisp = rcount + n;
      size_t so = 0, ro = 0;
      for (int i = 0; i < n; ++i) {
        scount[i] = a[k].sc[i]; sdisp[i] = a[k].sd ? a[k].sd[i] : so; so += a[k].sc[i];
        rcount[i] = a[k].rc[i]; rdisp[i] = a[k].rd ? a[k].rd[i] : ro; ro += a[k].rc[i];
      }
    }
    // all arrays of a piece in ONE RCCL launch: the all-to-all-v's are grouped
    ncclResult_t rc = ncclGroupStart();
    for (int k = 0; k < na && rc == ncclSuccess; ++k) {
      size_t *scount = &v[(size_t)4 * n * k], *sdisp = scount + n, *rcount = sdisp + n, *rdisp = rcount + n;
      rc = ncclAllToAllv(a[k].send, scount, sdisp, a[k].recv, rcount, rdisp, dtype(a[k].elem_bytes), comm, stream);
    }
    ncclResult_t re = ncclGroupEnd();
    if (rc == ncclSuccess) rc = re;
    if (rc != ncclSuccess) { err = std::string("ncclAllToAllv: ") + ncclGetErrorString(rc); return false; }
    return true;
  }
  bool allreduce(uint64_t* v, int cnt, bool is_max, hipStream_t stream, std::string& err) override {
    if (!comm) { err = "communicator aborted"; return false; }
    if (cnt > 64) { err = "allreduce: too many values"; return false; }
    std::memcpy(hscratch, v, sizeof(uint64_t) * cnt);
    if (hipMemcpyAsync(dscratch, hscratch, sizeof(uint64_t) * cnt, hipMemcpyHostToDevice, stream) != hipSuccess) { err = "hipMemcpyAsync"; return false; }
    ncclResult_t rc = ncclAllReduce(dscratch, dscratch + 64, (size_t)cnt, ncclUint64, is_max ? ncclMax : ncclSum, comm, stream);
    if (rc != ncclSuccess) { err = std::string("ncclAllReduce: ") + ncclGetErrorString(rc); return false; }
    if (hipMemcpyAsync(hscratch + 64, dscratch + 64, sizeof(uint64_t) * cnt, hipMemcpyDeviceToHost, stream) != hipSuccess) { err = "allreduce read-back"; return false; }
    if (!wait(stream, err)) return false;
    std::memcpy(v, hscratch + 64, sizeof(uint64_t) * cnt);
    return true;
  }
};

// ---- all ranks in one process (threads), one device ------------------------------------------------------------------------
struct LocalGroup {
  int n;
  std::mutex mu; std::condition_variable cv; int arrived; unsigned long gen; bool failed;
  std::vector<const uint64_t*> counts;                 // posted count arrays
  std::vector<uint64_t> status;
  struct Post { const XArr* a; int na; hipEvent_t ready, done; };
  std::vector<Post> posts;
  std::vector<const uint64_t*> red;
  explicit LocalGroup(int n_) : n(n_), arrived(0), gen(0), failed(false), counts(n_), status(n_), posts(n_), red(n_) {
    for (auto& p : posts) { hipEventCreateWithFlags(&p.ready, hipEventDisableTiming); hipEventCreateWithFlags(&p.done, hipEventDisableTiming); }
  }
  ~LocalGroup() { for (auto& p : posts) { hipEventDestroy(p.ready); hipEventDestroy(p.done); } }
  // false: a rank has left the group, or did not arrive within timeout_ms (nobody will ever complete this barrier)
  bool barrier(long long timeout_ms) {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const unsigned long g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else if (!cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return gen != g || failed; })) { failed = true; cv.notify_all(); }
    return !failed;
  }
  void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
};
struct LocalTransport : Transport {
  std::shared_ptr<LocalGroup> G;
  int r;
  int nranks() const override { return G->n; }
  int rank() const override { return r; }
  void abandon() override { G->fail(); }
  bool wait(hipStream_t stream, std::string& err) override {
    if (hipStreamSynchronize(stream) != hipSuccess) { err = "hipStreamSynchronize"; return false; }
    return true;
  }
  bool gone(std::string& err) { err = "a rank of the in-process group has left it or never arrived at a collective"; return false; }
  bool exchange_counts(const uint64_t* send, uint64_t* recv, int k, uint64_t status, uint64_t* worst, hipStream_t, std::string& err) override {
    G->counts[r] = send; G->status[r] = status;
    if (!G->barrier(timeout_ms)) return gone(err);
    uint64_t w = 0;
    for (int src = 0; src < G->n; ++src) {
      for (int j = 0; j < k; ++j) recv[src * k + j] = G->counts[src][r * k + j];
      w = std::max(w, G->status[src]);
    }
    *worst = w;
    if (!G->barrier(timeout_ms)) return gone(err);
    return true;
  }
  bool exchange(const XArr* a, int na, hipStream_t stream, std::string& err) override {
    LocalGroup::Post& me = G->posts[r];
    me.a = a; me.na = na;
    hipEventRecord(me.ready, stream);                      // what this rank sends was produced on `stream` before this point
    if (!G->barrier(timeout_ms)) return gone(err);
    bool ok = true;
    for (int k = 0; k < na; ++k) {
      uint64_t ro = 0;
      for (int src = 0; src < G->n; ++src) {
        const LocalGroup::Post& p = G->posts[src];
        uint64_t so = 0;
        if (p.a[k].sd) so = p.a[k].sd[r]; else for (int d = 0; d < r; ++d) so += p.a[k].sc[d];
        const uint64_t cnt = a[k].rc[src], dst = a[k].rd ? a[k].rd[src] : ro;
        if (k == 0 && hipStreamWaitEvent(stream, p.ready, 0) != hipSuccess) ok = false;
        if (cnt && hipMemcpyAsync(static_cast<char*>(a[k].recv) + dst * a[k].elem_bytes, static_cast<const char*>(p.a[k].send) + so * p.a[k].elem_bytes,
                                  cnt * a[k].elem_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) ok = false;
        ro += cnt;
      }
    }
    hipEventRecord(me.done, stream);
    if (!G->barrier(timeout_ms)) return gone(err);
    // a sender's buffers stay untouched until every receiver has copied from them
    for (int d = 0; d < G->n; ++d) if (hipStreamWaitEvent(stream, G->posts[d].done, 0) != hipSuccess) ok = false;
    if (!G->barrier(timeout_ms)) return gone(err);                    // (the posted pointers may be overwritten by the next exchange)
    if (!ok) err = "local exchange: HIP error";
    return ok;
  }
  bool allreduce(uint64_t* v, int cnt, bool is_max, hipStream_t, std::string& err) override {
    G->red[r] = v;
    if (!G->barrier(timeout_ms)) return gone(err);
    std::vector<uint64_t> s(cnt, 0);
    for (int i = 0; i < G->n; ++i) for (int j = 0; j < cnt; ++j) s[j] = is_max ? std::max(s[j], G->red[i][j]) : s[j] + G->red[i][j];
    if (!G->barrier(timeout_ms)) return gone(err);
    for (int j = 0; j < cnt; ++j) v[j] = s[j];             // (nobody reads a peer's v after the second barrier)
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

const int kMaxQP = 8;         // pieces of a pipelined query

}  // namespace

struct khd_map {
  std::unique_ptr<Transport> tp;
  kh_table* local;
  int device;
  kh_hash dist_hash; uint64_t dist_seed;
  hipStream_t stream, comm;
  // two sets of send / receive buffers: piece i travels while piece i-1 is partitioned
  Buf sk[2], sv[2], rk[2], rv[2], res[2];
  Buf dstat;                    // device: [0] this rank's status word, [1 .. p] the words received with the last result exchange
  uint64_t* hstat;              // pinned mirror
  bool late_pending;            // a find / count is in flight whose peers' status words have not been looked at
  bool force, dead;
  int query_pieces, fail_stage;
  hipEvent_t ev_perm, ev_landed[2], ev_fed[2], ev_sent[2];
  hipEvent_t ev_p[kMaxQP], ev_k[kMaxQP], ev_q[kMaxQP], ev_done;
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

// the transport failed (or a peer never arrived): this rank leaves, the map is finished
kh_status dead_end(khd_map* m) {
  m->dead = true;
  m->tp->abandon();
  return KH_ERR_HIP;          // (m->err holds the transport's message)
}
const char* status_name(uint64_t s) {
  static const char* const names[] = {"KH_OK", "KH_ERR_INVALID", "KH_ERR_NOMEM", "KH_ERR_FULL", "KH_ERR_PROBE_OVERFLOW", "KH_ERR_HIP", "KH_ERR_UNSUPPORTED", "KH_ERR_RETRY"};
  return s < 8 ? names[s] : "error";
}
// the local status and the reason of the FIRST local failure of a collective call
struct Note {
  kh_status st = KH_OK; std::string why;
  void operator()(kh_status s, const std::string& msg) { if (st == KH_OK && s != KH_OK) { st = s; why = msg; } }
  bool ok() const { return st == KH_OK; }
};
// after a vote: this rank's own failure, else the worst status among the peers
kh_status voted(khd_map* m, const Note& mine, uint64_t worst, const char* stage) {
  if (!mine.ok()) return fail(m, mine.st, mine.why);
  return fail(m, worst < 8 ? (kh_status)worst : KH_ERR_HIP, std::string("a peer rank failed (") + status_name(worst) + ") " + stage + "; nothing further was exchanged");
}

// status words of the last find / count: valid once the comm stream has drained
kh_status late_check(khd_map* m) {
  if (!m->late_pending) return KH_OK;
  if (!m->tp->wait(m->comm, m->err)) return dead_end(m);
  m->late_pending = false;
  uint64_t w = 0;
  for (int i = 0; i < m->tp->nranks(); ++i) w = std::max(w, m->hstat[1 + i]);
  if (w != KH_OK) return fail(m, w < 8 ? (kh_status)w : KH_ERR_HIP, std::string("a rank failed (") + status_name(w) + ") in the local part of the previous find / count: its results are invalid");
  return KH_OK;
}

kh_status make_map(khd_map** out, std::unique_ptr<Transport> tp, int device, kh_kind kind, kh_hash hash, uint64_t seed, uint64_t capacity,
                   float mn, float mx, kh_hash dist_hash, uint64_t dist_seed) {
  khd_map* m = new khd_map();
  m->tp = std::move(tp); m->local = nullptr; m->device = device; m->dist_hash = dist_hash; m->dist_seed = dist_seed;
  m->stream = nullptr; m->comm = nullptr; m->hstat = nullptr; m->late_pending = false; m->dead = false; m->query_pieces = 0; m->fail_stage = 0;
  const char* f = getenv("KH_DIST_FORCE_COLLECTIVES");
  m->force = f && f[0] == '1';
  const char* to = getenv("KHD_TIMEOUT_MS");
  if (to && atoll(to) > 0) m->tp->timeout_ms = atoll(to);
  kh_status s = kh_create(&m->local, kind, 8, 4, hash, seed, capacity, mn, mx, device);
  if (s != KH_OK) { delete m; return s; }
  const int p = m->tp->nranks();
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->comm, hipStreamNonBlocking) != hipSuccess ||
      !m->dstat.ensure(sizeof(uint64_t) * (p + 1)) || hipHostMalloc(reinterpret_cast<void**>(&m->hstat), sizeof(uint64_t) * (p + 1)) != hipSuccess) {
    kh_destroy(m->local); delete m; return KH_ERR_HIP;
  }
  std::memset(m->hstat, 0, sizeof(uint64_t) * (p + 1));
  hipEventCreateWithFlags(&m->ev_perm, hipEventDisableTiming);
  hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming);
  for (int i = 0; i < 2; ++i) {
    hipEventCreateWithFlags(&m->ev_landed[i], hipEventDisableTiming); hipEventCreateWithFlags(&m->ev_fed[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&m->ev_sent[i], hipEventDisableTiming);
  }
  for (int i = 0; i < kMaxQP; ++i) {
    hipEventCreateWithFlags(&m->ev_p[i], hipEventDisableTiming); hipEventCreateWithFlags(&m->ev_k[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&m->ev_q[i], hipEventDisableTiming);
  }
  *out = m;
  return KH_OK;
}

struct PlanGuard { kh_shard_plan* h = nullptr; ~PlanGuard() { kh_shard_plan_destroy(h); } };
// an open streamed insert never outlives the call that opened it
struct InsertGuard { kh_table* t; bool on; ~InsertGuard() { if (on) kh_insert_abort(t); } };

bool inject(khd_map* m, int stage) { if (m->fail_stage == stage) { m->fail_stage = 0; return true; } return false; }

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
  if (!m->dead) { hipStreamSynchronize(m->stream); hipStreamSynchronize(m->comm); }
  kh_destroy(m->local);
  for (int i = 0; i < 2; ++i) {
    m->sk[i].release(); m->sv[i].release(); m->rk[i].release(); m->rv[i].release(); m->res[i].release();
    hipEventDestroy(m->ev_landed[i]); hipEventDestroy(m->ev_fed[i]); hipEventDestroy(m->ev_sent[i]);
  }
  for (int i = 0; i < kMaxQP; ++i) { hipEventDestroy(m->ev_p[i]); hipEventDestroy(m->ev_k[i]); hipEventDestroy(m->ev_q[i]); }
  m->dstat.release();
  if (m->hstat) hipHostFree(m->hstat);
  hipEventDestroy(m->ev_perm); hipEventDestroy(m->ev_done);
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
kh_status khd_set_option(khd_map* m, int option, long long value) {
  if (!m) return KH_ERR_INVALID;
  switch (option) {
    case KHD_OPT_FORCE_COLLECTIVES: m->force = value != 0; return KH_OK;
    case KHD_OPT_QUERY_PIECES: if (value < 0 || value > kMaxQP) return fail(m, KH_ERR_INVALID, "query pieces: 0 (auto) .. 8"); m->query_pieces = (int)value; return KH_OK;
    case KHD_OPT_TIMEOUT_MS: if (value <= 0) return fail(m, KH_ERR_INVALID, "timeout must be positive"); m->tp->timeout_ms = value; return KH_OK;
    default: return fail(m, KH_ERR_INVALID, "unknown option");
  }
}
kh_status khd_debug_fail_next(khd_map* m, int stage) { if (!m) return KH_ERR_INVALID; m->fail_stage = stage; return KH_OK; }
const char* khd_last_error(const khd_map* m) { return m ? m->err.c_str() : "null map"; }
kh_table* khd_local(khd_map* m) { return m ? m->local : nullptr; }
int khd_rank(const khd_map* m) { return m ? m->tp->rank() : -1; }
int khd_nranks(const khd_map* m) { return m ? m->tp->nranks() : 0; }

kh_status khd_synchronize(khd_map* m) {
  if (!m) return KH_ERR_INVALID;
  if (m->dead) return fail(m, KH_ERR_HIP, "the map's communicator was aborted after a transport failure");
  KHD_HIP(hipSetDevice(m->device));
  if (!m->tp->wait(m->comm, m->err) || !m->tp->wait(m->stream, m->err)) return dead_end(m);
  return late_check(m);
}

kh_status khd_insert(khd_map* m, const uint64_t* keys, const uint32_t* vals, uint64_t n, int pieces, int reduce_plus, uint64_t* n_inserted) {
  if (!m) return KH_ERR_INVALID;
  if (n_inserted) *n_inserted = 0;
  if ((n && !keys) || (!vals && !reduce_plus) || pieces > 16) return fail(m, KH_ERR_INVALID, "khd_insert: bad arguments");
  if (m->dead) return fail(m, KH_ERR_HIP, "the map's communicator was aborted after a transport failure");
  KHD_HIP(hipSetDevice(m->device));
  { kh_status ls = late_check(m); if (ls != KH_OK) return ls; }
  const int p = m->tp->nranks();
  if (p == 1 && !m->force) {       // one rank owns every key
    KHD_KH(reduce_plus ? kh_insert_reduce_plus(m->local, keys, vals, n, KH_MEM_DEVICE, n_inserted) : kh_insert(m->local, keys, vals, n, KH_MEM_DEVICE, n_inserted));
    return KH_OK;
  }
  Transport& T = *m->tp;
  Note note;
  if (pieces < 1) pieces = 1;
  std::vector<uint64_t> bounds(pieces + 1);
  for (int i = 0; i <= pieces; ++i) bounds[i] = n * (uint64_t)i / (uint64_t)pieces;
  // ---- stage 1 (local): destination counts of every piece, send buffers.  Up to 8 ranks: one count sweep + scan + host
  //      synchronisation for the whole batch (kh_shard_plan; the pieces are then permuted without counting again, their boundaries
  //      are the plan's: multiples of 4096 pairs); more ranks: a count-only pass per piece
  std::vector<uint64_t> sc((size_t)p * pieces, 0), rc((size_t)p * pieces, 0), tmp(p);
  PlanGuard plan;
  if (inject(m, 1)) note(KH_ERR_NOMEM, "injected failure (stage 1)");
  if (note.ok()) {
    Span sp(m, "count_pass", m->stream);
    if (p <= 8) {
      std::vector<uint64_t> pc((size_t)pieces * p);
      kh_status st = kh_shard_plan_create(&plan.h, m->dist_hash, m->dist_seed, KH_XF_IDENTITY, 0, (uint32_t)p, keys, n, (uint32_t)pieces, pc.data(), bounds.data(),
                                          m->device, m->stream);
      note(st, "kh_shard_plan_create");
      if (st == KH_OK) for (int i = 0; i < pieces; ++i) for (int d = 0; d < p; ++d) sc[(size_t)d * pieces + i] = pc[(size_t)i * p + d];
    } else
    for (int i = 0; i < pieces && note.ok(); ++i) {
      kh_status st = kh_shard_permute(m->dist_hash, m->dist_seed, (uint32_t)p, keys + bounds[i], nullptr, bounds[i + 1] - bounds[i], nullptr, nullptr,
                                      tmp.data(), m->device, m->stream);
      note(st, "kh_shard_permute (count only)");
      for (int d = 0; d < p; ++d) sc[(size_t)d * pieces + i] = tmp[d];
    }
  }
  { uint64_t mx = 1;
    for (int i = 0; i < pieces; ++i) mx = std::max(mx, bounds[i + 1] - bounds[i]);
    for (int s = 0; s < (pieces > 1 ? 2 : 1); ++s)
      if (!m->sk[s].ensure(mx * 8) || (vals && !m->sv[s].ensure(mx * 4))) note(KH_ERR_NOMEM, "send buffers"); }
  // ---- ONE exchange for the counts of all pieces (rc[src * pieces + piece]); the status word of stage 1 travels with it
  uint64_t worst = 0;
  if (!T.exchange_counts(sc.data(), rc.data(), pieces, (uint64_t)note.st, &worst, m->stream, m->err)) return dead_end(m);
  if (worst != KH_OK) return voted(m, note, worst, "before the count exchange");
  // ---- stage 2 (local): the receive side.  Every received piece is KEPT (one buffer for the whole batch, piece after piece) until
  //      the build has succeeded: the local table may then partition the pieces speculatively (KH_INS_REPEATABLE: no histogram pass,
  //      slots shared by all pieces) and ask for them again if that did not hold (KH_ERR_RETRY)
  uint64_t total = 0;
  for (int i = 0; i < pieces; ++i) for (int s = 0; s < p; ++s) total += rc[(size_t)s * pieces + i];
  if (inject(m, 2)) note(KH_ERR_NOMEM, "injected failure (stage 2)");
  if (note.ok() && (!m->rk[0].ensure(std::max<uint64_t>(total, 1) * 8) || (vals && !m->rv[0].ensure(std::max<uint64_t>(total, 1) * 4)))) note(KH_ERR_NOMEM, "receive buffers");
  uint64_t* const rk_all = static_cast<uint64_t*>(m->rk[0].p);
  uint32_t* const rv_all = vals ? static_cast<uint32_t*>(m->rv[0].p) : nullptr;
  InsertGuard guard{m->local, false};
  if (note.ok()) {
    kh_status st = kh_insert_begin_ex(m->local, total, (reduce_plus ? KH_INS_REDUCE_PLUS : 0u) | KH_INS_REPEATABLE);
    note(st, std::string("kh_insert_begin_ex: ") + kh_last_error(m->local));
    guard.on = st == KH_OK;
  }
  { uint64_t v = (uint64_t)note.st;
    if (!T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
    if (v != KH_OK) return voted(m, note, v, "while preparing to receive"); }
  // ---- stage 3: the pieces.  A local failure from here on is noted; the rank keeps exchanging (its peers expect its pairs: they get
  //      whatever the send buffers hold and learn of the failure in the final vote) and skips its own local work
  std::vector<uint64_t> scounts(p), rcounts(p), roff(pieces + 1, 0);
  int landed = -1;
  for (int i = 0; i < pieces; ++i) {
    const int s = i & 1;
    // send set s was last read by the exchange of piece i-2
    if (hipStreamWaitEvent(m->stream, m->ev_sent[s], 0) != hipSuccess) note(KH_ERR_HIP, "hipStreamWaitEvent");
    if (i == 0 && inject(m, 3)) note(KH_ERR_NOMEM, "injected failure (stage 3)");      // (the rest of the loop: a failed rank keeps exchanging)
    if (plan.h) {
      if (note.ok()) {
        Span sp(m, "permute", m->stream);
        note(kh_shard_plan_permute(plan.h, (uint32_t)i, keys, vals, static_cast<uint64_t*>(m->sk[s].p), vals ? static_cast<uint32_t*>(m->sv[s].p) : nullptr, m->stream), "kh_shard_plan_permute");
      }
      for (int d = 0; d < p; ++d) scounts[d] = sc[(size_t)d * pieces + i];
    } else {
      for (int d = 0; d < p; ++d) scounts[d] = sc[(size_t)d * pieces + i];
      if (note.ok()) {
        Span sp(m, "permute", m->stream);
        note(kh_shard_permute(m->dist_hash, m->dist_seed, (uint32_t)p, keys + bounds[i], vals ? vals + bounds[i] : nullptr, bounds[i + 1] - bounds[i],
                              static_cast<uint64_t*>(m->sk[s].p), vals ? static_cast<uint32_t*>(m->sv[s].p) : nullptr, tmp.data(), m->device, m->stream), "kh_shard_permute");
      }
    }
    uint64_t rtot = 0;
    for (int src = 0; src < p; ++src) { rcounts[src] = rc[(size_t)src * pieces + i]; rtot += rcounts[src]; }
    roff[i + 1] = roff[i] + rtot;
    if (hipEventRecord(m->ev_perm, m->stream) != hipSuccess || hipStreamWaitEvent(m->comm, m->ev_perm, 0) != hipSuccess) note(KH_ERR_HIP, "event");
    { Span sp(m, "exchange", m->comm);
      const XArr xa[2] = {{m->sk[s].p, rk_all + roff[i], 8, scounts.data(), nullptr, rcounts.data(), nullptr},
                          {m->sv[s].p, rv_all ? rv_all + roff[i] : nullptr, 4, scounts.data(), nullptr, rcounts.data(), nullptr}};
      if (!T.exchange(xa, vals ? 2 : 1, m->comm, m->err)) return dead_end(m); }
    if (hipEventRecord(m->ev_landed[s], m->comm) != hipSuccess || hipEventRecord(m->ev_sent[s], m->comm) != hipSuccess) note(KH_ERR_HIP, "event");
    if (landed >= 0 && note.ok()) {      // piece i-1 has landed (or is landing): partition it while piece i travels
      if (hipStreamWaitEvent(m->stream, m->ev_landed[landed], 0) != hipSuccess) note(KH_ERR_HIP, "hipStreamWaitEvent");
      Span sp(m, "feed", m->stream);
      kh_status st = kh_insert_feed(m->local, rk_all + roff[i - 1], rv_all ? rv_all + roff[i - 1] : nullptr, roff[i] - roff[i - 1], KH_MEM_DEVICE);
      note(st, std::string("kh_insert_feed: ") + kh_last_error(m->local));
    }
    landed = s;
  }
  if (note.ok()) {
    if (hipStreamWaitEvent(m->stream, m->ev_landed[landed], 0) != hipSuccess) note(KH_ERR_HIP, "hipStreamWaitEvent");
    Span sp(m, "feed", m->stream);
    kh_status st = kh_insert_feed(m->local, rk_all + roff[pieces - 1], rv_all ? rv_all + roff[pieces - 1] : nullptr, roff[pieces] - roff[pieces - 1], KH_MEM_DEVICE);
    note(st, std::string("kh_insert_feed: ") + kh_last_error(m->local));
  }
  // ---- vote: did every rank send real pairs and feed what it received?  If not, nobody builds: what a failed rank sent in place of
  //      its pairs must not reach any table
  { uint64_t v = (uint64_t)note.st;
    if (!note.ok() && !T.wait(m->comm, m->err)) return dead_end(m);
    if (!T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
    if (v != KH_OK) {
      if (guard.on) { kh_insert_abort(m->local); guard.on = false; }
      return voted(m, note, v, "while the pieces were exchanged; nothing was inserted on any rank");
    } }
  // ---- stage 4: the build
  if (inject(m, 4)) note(KH_ERR_NOMEM, "injected failure (stage 4)");
  if (note.ok()) {
    kh_status est;
    { Span sp(m, "build", m->stream);
      est = kh_insert_end(m->local, n_inserted); }
    guard.on = false;
    if (est == KH_ERR_RETRY) {      // the speculative partition did not hold: the kept pieces, concatenated in feed order, the exact way
      Span sp(m, "refeed", m->stream);
      est = kh_insert_begin(m->local, total, reduce_plus);
      if (est == KH_OK) {
        guard.on = true;
        est = kh_insert_feed(m->local, rk_all, rv_all, total, KH_MEM_DEVICE);
        if (est == KH_OK) { est = kh_insert_end(m->local, n_inserted); guard.on = false; }
      }
    }
    if (est != KH_OK) note(est, kh_last_error(m->local));
  }
  if (!note.ok()) {       // nothing of this rank's share was inserted
    if (guard.on) { kh_insert_abort(m->local); guard.on = false; }
    if (n_inserted) *n_inserted = 0;
  }
  { uint64_t v = (uint64_t)note.st;
    if (!T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
    if (v != KH_OK) return voted(m, note, v, "in the build: the ranks that did not fail hold their share of the batch"); }
  return KH_OK;
}

// keys out (grouped by owner, in pieces), the local query of a piece while the next one travels, results back with the swapped
// counts (khmxx::ialltoallv_and_query_one_to_one, incremental_mxx.hpp:4403-4669).  op: 0 count, 1 find, 2 erase
static kh_status query(khd_map* m, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals, uint8_t* out_flags, int op, uint64_t* n_local) {
  if (m->dead) return fail(m, KH_ERR_HIP, "the map's communicator was aborted after a transport failure");
  KHD_HIP(hipSetDevice(m->device));
  { kh_status ls = late_check(m); if (ls != KH_OK) return ls; }
  const int p = m->tp->nranks();
  if (n_local) *n_local = 0;
  if (p == 1 && !m->force) {
    if (out_keys && n) KHD_HIP(hipMemcpyAsync(out_keys, keys, n * 8, hipMemcpyDeviceToDevice, m->stream));
    Span sp(m, "query", m->stream);
    if (op == 0) KHD_KH(kh_count(m->local, keys, n, KH_MEM_DEVICE, out_flags));
    else if (op == 1) KHD_KH(kh_find(m->local, keys, n, KH_MEM_DEVICE, out_vals, out_flags, nullptr));
    else KHD_KH(kh_erase(m->local, keys, n, KH_MEM_DEVICE, n_local));
    return KH_OK;
  }
  Transport& T = *m->tp;
  Note note;
  // pieces of THIS rank's queries: its own choice (its own n); the count exchange always carries kMaxQP counts per destination plus
  // the choice, and every rank then runs as many rounds as the rank with the most pieces (zero-sized parts for the others).
  // An erase re-lays the table out once: one piece.
  // (measured over RCCL, one rank, self-exchange, 10^7 finds: every extra piece costs ~0.045 ms of launches -- 0.82 / 0.87 / 0.96 / 1.33 ms
  //  for 1 / 2 / 4 / 8 pieces -- against ~0.4 ms of exchange that can hide behind the probes: 4 pieces from 2^23 keys, 2 from 2^22)
  int my_pieces = m->query_pieces > 0 ? m->query_pieces : (n >= (uint64_t(1) << 23) ? 4 : (n >= (uint64_t(1) << 22) ? 2 : 1));
  if (op == 2 || p > 8) my_pieces = 1;
  // ---- stage 1 (local): where the permuted keys go, the plan (ONE count sweep + synchronisation), the counts
  uint64_t* pk = out_keys;
  if (!pk) { if (!m->sk[0].ensure(std::max<uint64_t>(n, 1) * 8)) note(KH_ERR_NOMEM, "send buffer"); pk = static_cast<uint64_t*>(m->sk[0].p); }
  if (inject(m, 1)) note(KH_ERR_NOMEM, "injected failure (stage 1)");
  PlanGuard plan;
  const int K = kMaxQP + 1;
  std::vector<uint64_t> off((size_t)p * (my_pieces + 1), 0), sc((size_t)p * K, 0), rc((size_t)p * K, 0), bounds(my_pieces + 1, 0);
  bool permuted = false;
  if (note.ok()) {
    Span sp(m, "count_pass", m->stream);
    if (p <= 8) {
      std::vector<uint64_t> pc((size_t)my_pieces * p);
      kh_status st = kh_shard_plan_create(&plan.h, m->dist_hash, m->dist_seed, KH_XF_IDENTITY, 0, (uint32_t)p, keys, n, (uint32_t)my_pieces, pc.data(), bounds.data(), m->device, m->stream);
      note(st, "kh_shard_plan_create");
      if (st == KH_OK) note(kh_shard_plan_offsets(plan.h, off.data()), "kh_shard_plan_offsets");
    } else {        // one stable permutation gives the counts too
      std::vector<uint64_t> cnt(p);
      kh_status st = kh_shard_permute(m->dist_hash, m->dist_seed, (uint32_t)p, keys, nullptr, n, pk, nullptr, cnt.data(), m->device, m->stream);
      note(st, "kh_shard_permute");
      uint64_t run = 0;
      for (int d = 0; d < p; ++d) { off[(size_t)d * 2] = run; run += cnt[d]; off[(size_t)d * 2 + 1] = run; }
      permuted = true;
    }
    if (note.ok()) for (int d = 0; d < p; ++d) for (int i = 0; i < my_pieces; ++i) sc[(size_t)d * K + i] = off[(size_t)d * (my_pieces + 1) + i + 1] - off[(size_t)d * (my_pieces + 1) + i];
  }
  for (int d = 0; d < p; ++d) sc[(size_t)d * K + kMaxQP] = (uint64_t)my_pieces;
  uint64_t worst = 0;
  if (!T.exchange_counts(sc.data(), rc.data(), K, (uint64_t)note.st, &worst, m->stream, m->err)) return dead_end(m);
  if (worst != KH_OK) return voted(m, note, worst, "before the count exchange");
  int rounds = 1;
  for (int s = 0; s < p; ++s) rounds = std::max<int>(rounds, (int)std::min<uint64_t>(rc[(size_t)s * K + kMaxQP], kMaxQP));
  // ---- stage 2 (local): the receive side: keys and results of all rounds, round after round
  std::vector<uint64_t> roff(rounds + 1, 0);
  for (int i = 0; i < rounds; ++i) { uint64_t t = 0; for (int s = 0; s < p; ++s) t += rc[(size_t)s * K + i]; roff[i + 1] = roff[i] + t; }
  const uint64_t rtot = roff[rounds];
  if (inject(m, 2)) note(KH_ERR_NOMEM, "injected failure (stage 2)");
  if (note.ok() && (!m->rk[0].ensure(std::max<uint64_t>(rtot, 1) * 8) || !m->res[0].ensure(std::max<uint64_t>(rtot, 1) * 4) || !m->res[1].ensure(std::max<uint64_t>(rtot, 1))))
    note(KH_ERR_NOMEM, "receive buffers");
  { uint64_t v = (uint64_t)note.st;
    if (!T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
    if (v != KH_OK) return voted(m, note, v, "while preparing to receive"); }
  // ---- stage 3: everything is queued at once; no host synchronisation (an erase returns a host scalar and votes at the end)
  uint64_t* const rkeys = static_cast<uint64_t*>(m->rk[0].p);
  uint32_t* const lv = static_cast<uint32_t*>(m->res[0].p); uint8_t* const lf = static_cast<uint8_t*>(m->res[1].p);
  if (op == 1 && hipMemsetAsync(lv, 0, std::max<uint64_t>(rtot, 1) * 4, m->stream) != hipSuccess) note(KH_ERR_HIP, "hipMemsetAsync");
  if (inject(m, 3)) note(KH_ERR_NOMEM, "injected failure (stage 3)");
  for (int i = 0; i < rounds; ++i) {       // permutation of every piece first: the exchanges can start as early as possible
    if (!permuted && i < my_pieces && note.ok()) {
      Span sp(m, "permute", m->stream);
      note(kh_shard_plan_permute_global(plan.h, (uint32_t)i, keys, nullptr, pk, nullptr, m->stream), "kh_shard_plan_permute_global");
    }
    if (hipEventRecord(m->ev_p[i], m->stream) != hipSuccess) note(KH_ERR_HIP, "hipEventRecord");
  }
  // per round: what this rank sends to / receives from every peer, and where
  std::vector<uint64_t> scn((size_t)rounds * p, 0), sdn((size_t)rounds * p, 0), rcn((size_t)rounds * p, 0), ones(p, 1), zeros(p, 0);
  for (int i = 0; i < rounds; ++i)
    for (int d = 0; d < p; ++d) {
      if (i < my_pieces) { scn[(size_t)i * p + d] = sc[(size_t)d * K + i]; sdn[(size_t)i * p + d] = off[(size_t)d * (my_pieces + 1) + i]; }
      rcn[(size_t)i * p + d] = rc[(size_t)d * K + i];
    }
  auto keys_out = [&](int i) -> bool {
    if (hipStreamWaitEvent(m->comm, m->ev_p[i], 0) != hipSuccess) note(KH_ERR_HIP, "hipStreamWaitEvent");
    Span sp(m, "exchange", m->comm);
    const XArr xa[1] = {{pk, rkeys + roff[i], 8, &scn[(size_t)i * p], &sdn[(size_t)i * p], &rcn[(size_t)i * p], nullptr}};
    if (!T.exchange(xa, 1, m->comm, m->err)) return false;
    if (hipEventRecord(m->ev_k[i], m->comm) != hipSuccess) note(KH_ERR_HIP, "hipEventRecord");
    return true;
  };
  if (!keys_out(0)) return dead_end(m);
  for (int i = 0; i < rounds; ++i) {
    if (i + 1 < rounds && !keys_out(i + 1)) return dead_end(m);       // round i+1 travels while round i is probed
    if (hipStreamWaitEvent(m->stream, m->ev_k[i], 0) != hipSuccess) note(KH_ERR_HIP, "hipStreamWaitEvent");
    const uint64_t cnt = roff[i + 1] - roff[i];
    if (op == 2) {       // did every rank send real keys?  (what a failed rank sent in place of its keys must not be erased anywhere)
      uint64_t v = (uint64_t)note.st;
      if (!T.wait(m->comm, m->err) || !T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
      if (v != KH_OK) return voted(m, note, v, "while the keys were exchanged; nothing was erased on any rank");
      if (inject(m, 4)) note(KH_ERR_NOMEM, "injected failure (stage 4)");
    }
    if (note.ok()) {
      Span sp(m, "query", m->stream);
      kh_status st = KH_OK;
      if (op == 0) st = kh_count(m->local, rkeys + roff[i], cnt, KH_MEM_DEVICE, lf + roff[i]);
      else if (op == 1) st = kh_find(m->local, rkeys + roff[i], cnt, KH_MEM_DEVICE, lv + roff[i], lf + roff[i], nullptr);
      else st = kh_erase(m->local, rkeys + roff[i], cnt, KH_MEM_DEVICE, n_local);
      note(st, std::string("local query: ") + kh_last_error(m->local));
    }
    if (op == 2) continue;
    if (hipEventRecord(m->ev_q[i], m->stream) != hipSuccess || hipStreamWaitEvent(m->comm, m->ev_q[i], 0) != hipSuccess) note(KH_ERR_HIP, "event");
    // results return with the swapped counts (:1495) to their place in the permuted order; values and flags in one launch, and
    // with the last round the status word of this rank's local work
    const bool last = i == rounds - 1;
    if (last) {
      m->hstat[0] = (uint64_t)note.st;
      if (hipMemcpyAsync(m->dstat.p, m->hstat, 8, hipMemcpyHostToDevice, m->comm) != hipSuccess) note(KH_ERR_HIP, "hipMemcpyAsync");
    }
    { Span sp(m, "exchange", m->comm);
      XArr xa[3]; int na = 0;
      if (op == 1) xa[na++] = XArr{lv + roff[i], out_vals, 4, &rcn[(size_t)i * p], nullptr, &scn[(size_t)i * p], &sdn[(size_t)i * p]};
      xa[na++] = XArr{lf + roff[i], out_flags, 1, &rcn[(size_t)i * p], nullptr, &scn[(size_t)i * p], &sdn[(size_t)i * p]};
      if (last) xa[na++] = XArr{m->dstat.p, static_cast<uint64_t*>(m->dstat.p) + 1, 8, ones.data(), zeros.data(), ones.data(), nullptr};
      if (!T.exchange(xa, na, m->comm, m->err)) return dead_end(m); }
    if (last) {
      if (hipMemcpyAsync(m->hstat + 1, static_cast<uint64_t*>(m->dstat.p) + 1, sizeof(uint64_t) * p, hipMemcpyDeviceToHost, m->comm) != hipSuccess) note(KH_ERR_HIP, "hipMemcpyAsync");
      m->late_pending = true;
    }
  }
  if (op == 2) {       // the erase counts are on the host already; one vote tells every rank whether all of them succeeded
    uint64_t v = (uint64_t)note.st;
    if (!T.allreduce(&v, 1, true, m->stream, m->err)) return dead_end(m);
    if (v != KH_OK) return voted(m, note, v, "during the erase: the ranks that did not fail erased their share");
    return KH_OK;
  }
  // the results are complete in the order of the table's stream; nothing has been waited for on the host
  if (hipEventRecord(m->ev_done, m->comm) != hipSuccess || hipStreamWaitEvent(m->stream, m->ev_done, 0) != hipSuccess) note(KH_ERR_HIP, "event");
  if (!note.ok()) return fail(m, note.st, note.why);
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
  if (m->dead) return fail(m, KH_ERR_HIP, "the map's communicator was aborted after a transport failure");
  { kh_status ls = late_check(m); if (ls != KH_OK) return ls; }
  uint64_t v = 0;
  KHD_KH(kh_size(m->local, &v));
  if ((m->tp->nranks() > 1 || m->force) && !m->tp->allreduce(&v, 1, false, m->stream, m->err)) return dead_end(m);
  *global_size = v;
  return KH_OK;
}

kh_status khd_phase_ms(khd_map* m, char* buf, uint64_t cap) {
  if (!m || !buf || !cap) return KH_ERR_INVALID;
  hipSetDevice(m->device);
  if (!m->dead) { hipStreamSynchronize(m->stream); hipStreamSynchronize(m->comm); }
  std::vector<std::pair<std::string, double> > acc;
  for (auto& ph : m->phases) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ph.a, ph.b) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
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
