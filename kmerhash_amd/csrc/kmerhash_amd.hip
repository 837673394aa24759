// kmerhash_amd.hip -- host side of libkmerhash_amd.so: table state machine + C-ABI (include/kmerhash_amd.h).
//
// Host logic restates the reference's *observable* contract (sizes, float load thresholds, the
// doubling/halving rules, first-value-wins) around the chunked GPU kernels of kh_kernels.h:
//   insert  : sample -> partition the batch by chunk -> one-launch build (k_build_fused; speculates that the capacity the
//             reference's rule yields is the predicted one) or, when a speculation fails, LDS de-dup + membership test (k_dedup)
//             -> exact capacity decision (hashmap_robinhood.hpp:530 rule, evaluated in closed form) -> chunk rebuild into a
//             fresh buffer.  The old buffer stays valid until the new one is complete, so a failing batch (probe distance >= 128)
//             leaves the table untouched.  Batches of up to 16 keys and mid-size batches are applied in place.
//   erase   : RH partitions the erase keys by chunk and drops them inside a one-launch re-layout (fall-back: mark hits, re-lay
//             out without them); LP writes tombstones in place.
//   find/count : sector probing with in-launch compaction (k_find).
#include "kh_kernels.h"
#include "../../include/kmerhash_amd.h"

#include <string>
#include <vector>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>

#define KH_VERSION_STR "kmerhash_amd 0.1 (gfx950)"

// ---- process-wide device memory cache -------------------------------------------------------------
// hipMalloc/hipFree of multi-GB buffers cost milliseconds to hundreds of milliseconds and synchronise the
// device; tables are typically created, filled and destroyed in a loop (one per benchmark repeat, one per
// file batch), so freed table buffers and workspaces are kept and handed out again (best fit).
namespace {
struct DevPool {
  std::mutex m;
  std::multimap<size_t, void*> free_;                 // size -> block
  std::unordered_map<void*, size_t> size_of;          // every live or cached block
  size_t cached = 0;
};
DevPool g_pool[16];
const size_t kPoolGranule = size_t(2) << 20;

void pool_trim(int dev) {
  DevPool& P = g_pool[dev & 15];
  std::lock_guard<std::mutex> g(P.m);
  for (auto& kv : P.free_) { hipFree(kv.second); P.size_of.erase(kv.second); }
  P.free_.clear();
  P.cached = 0;
}
hipError_t pool_alloc(int dev, size_t bytes, void** out) {
  DevPool& P = g_pool[dev & 15];
  bytes = (bytes + kPoolGranule - 1) / kPoolGranule * kPoolGranule;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_.lower_bound(bytes);
    if (it != P.free_.end() && it->first <= bytes + bytes / 4 + (size_t(16) << 20)) {
      *out = it->second;
      P.cached -= it->first;
      P.free_.erase(it);
      return hipSuccess;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {            // give cached blocks back to the driver and retry once
    (void)hipGetLastError();
    pool_trim(dev);
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(P.m);
  P.size_of[p] = bytes;
  *out = p;
  return hipSuccess;
}
void pool_free(int dev, void* p) {
  if (!p) return;
  DevPool& P = g_pool[dev & 15];
  std::lock_guard<std::mutex> g(P.m);
  auto it = P.size_of.find(p);
  if (it == P.size_of.end()) { hipFree(p); return; }
  P.free_.insert(std::make_pair(it->second, p));
  P.cached += it->second;
}
// pinned host scratch blocks (512 B, one per table) and profiling events are recycled too: hipHostFree synchronises the
// device (0.2 ms per table destroyed, 4% of a benchmark step)
struct HostPool { std::mutex m; std::vector<uint64_t*> pinned; std::vector<hipEvent_t> events[16]; };   // events: per device
HostPool g_host;
uint64_t* pinned_get() {
  { std::lock_guard<std::mutex> g(g_host.m);
    if (!g_host.pinned.empty()) { uint64_t* p = g_host.pinned.back(); g_host.pinned.pop_back(); return p; } }
  uint64_t* p = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&p), 64 * sizeof(uint64_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void pinned_put(uint64_t* p) { if (p) { std::lock_guard<std::mutex> g(g_host.m); g_host.pinned.push_back(p); } }
hipEvent_t event_get(int dev) {      // the caller has made `dev` current
  { std::lock_guard<std::mutex> g(g_host.m);
    auto& v = g_host.events[dev & 15];
    if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; } }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}
void event_put(int dev, hipEvent_t e) { if (e) { std::lock_guard<std::mutex> g(g_host.m); g_host.events[dev & 15].push_back(e); } }
}  // namespace

namespace {

struct ProfRec { const char* name; hipEvent_t a, b; };

struct Block { char* p; size_t cap; };

}  // namespace

// histogram-free layout shared by the pieces of a streamed insert: every piece appends to the SAME slots (cursors, overflow flag
// and the final buffer live across the feeds), so that the build at the end sees one source
struct SlackShared { unsigned long long* cur2; uint64_t* starts; uint32_t* ovf; uint64_t slot; };
struct kh_table {
  int kind, hash, device;
  KhSeed seed;      // storage hash seed + key transform (kh_set_key_transform)
  hipStream_t stream;
  float min_lf, max_lf;
  uint64_t min_load, max_load, lsize;
  KhSlots cur, spare;
  // workspace arena (grow-only, reset per operation)
  std::vector<Block> blocks;
  size_t blk, off;
  uint64_t* hpin;   // pinned host scratch (64 x u64)
  std::string err;
  // streamed insert (kh_insert_begin / feed / end)
  struct {
    bool active, fallback, nodup; int mode;       // nodup: a sample of the FIRST feed found no duplicate key (the build then skips its fold, speculatively)
    // KH_INS_REPEATABLE: the caller can feed the same pieces again, so the speculative forms are allowed: histogram-free partition
    // into slots shared by all pieces (one source for the build), 12-byte records when the sample found no duplicate
    bool repeatable, slack, rec12;
    SlackShared sh;
    uint64_t n_total, fed, cap_u; uint32_t PB;
    ulonglong2 *tmp, *fin;
    KhSrcSet S;
    uint64_t* stage_k; uint32_t* stage_v;
  } ins;
  uint32_t* part_overflow;      // device flag of the histogram-free partition feeding the operation in flight (or null)
  double batch_vf;              // variance factor E[m^2]/E[m] the duplicate sample gave for the batch in flight (1: no duplicate seen)
  bool batch_nodup;             // a sample of the batch in flight found no duplicate key (k_sample_dups)
  bool prof;
  std::vector<ProfRec> recs;
  std::vector<std::pair<std::string, std::pair<double, uint64_t> > > prof_acc;
};

namespace {

inline uint64_t next_pow2(uint64_t x) {   // math_utils.hpp:64-69 (x<=1 -> 1)
  if (x <= 1) return 1;
  return uint64_t(1) << (64 - __builtin_clzll(x - 1));
}
inline uint64_t threshold(uint64_t buckets, float lf) {   // hashmap_robinhood.hpp:263,269: float arithmetic
  return static_cast<uint64_t>(static_cast<float>(buckets) * lf);
}
inline uint32_t log2u(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

kh_status fail(kh_table* t, kh_status s, const std::string& msg) {
  if (t) t->err = msg;
  return s;
}
#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      return fail(t, e__ == hipErrorOutOfMemory ? KH_ERR_NOMEM : KH_ERR_HIP,                          \
                  std::string(#call) + ": " + hipGetErrorString(e__));                                \
  } while (0)

// ---- workspace arena ---------------------------------------------------------------------------
void arena_reset(kh_table* t) { t->blk = 0; t->off = 0; }
kh_status arena_take(kh_table* t, size_t bytes, void** out) {
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes == 0) bytes = 256;
  while (t->blk < t->blocks.size()) {
    Block& b = t->blocks[t->blk];
    if (t->off + bytes <= b.cap) { *out = b.p + t->off; t->off += bytes; return KH_OK; }
    ++t->blk; t->off = 0;
  }
  size_t cap = std::max(bytes, size_t(64) << 20);
  void* vp = nullptr;
  if (getenv("KH_DEBUG_ARENA")) fprintf(stderr, "[kmerhash_amd] arena grows by %.3f GB (block %zu)\n", cap / 1e9, t->blocks.size());
  HIPCHK(pool_alloc(t->device, cap, &vp));
  char* p = static_cast<char*>(vp);
  Block b; b.p = p; b.cap = cap;
  t->blocks.push_back(b);
  t->blk = t->blocks.size() - 1;
  t->off = bytes;
  *out = p;
  return KH_OK;
}
// Before an operation: make the arena ONE block of at least `bytes` (an upper estimate of what the
// operation takes), so that the steady state allocates nothing; arena_take still grows on demand.
kh_status arena_prepare(kh_table* t, size_t bytes) {
  arena_reset(t);
  if (t->blocks.size() == 1 && t->blocks[0].cap >= bytes) return KH_OK;
  size_t total = 0;
  for (auto& b : t->blocks) total += b.cap;
  if (!t->blocks.empty()) {
    hipStreamSynchronize(t->stream);
    for (auto& b : t->blocks) pool_free(t->device, b.p);
    t->blocks.clear();
  }
  void* p = nullptr;
  // (3 % of headroom: a caller's batches differ by a few reads -- without it the NEXT batch, a megabyte larger, frees and allocates the whole
  //  arena again: 0.27 s for 35 GB where hipMalloc is slow)
  const size_t want = std::max(bytes + bytes / 32, total);
  if (getenv("KH_DEBUG_ARENA")) fprintf(stderr, "[kmerhash_amd] arena prepared: %.3f GB (asked %.3f, held %.3f)\n", want / 1e9, bytes / 1e9, total / 1e9);
  HIPCHK(pool_alloc(t->device, want, &p));
  Block b; b.p = static_cast<char*>(p); b.cap = want;
  t->blocks.push_back(b);
  return KH_OK;
}
void arena_consolidate(kh_table*) {}
#define TAKE(ptr, type, count)                                                                  \
  do {                                                                                          \
    void* p__ = nullptr;                                                                        \
    kh_status s__ = arena_take(t, sizeof(type) * size_t(count), &p__);                          \
    if (s__ != KH_OK) return s__;                                                               \
    ptr = static_cast<type*>(p__);                                                              \
  } while (0)

// ---- slots ---------------------------------------------------------------------------------------
const KhSlots kNoSlots = KhSlots{nullptr, 0};
const bool g_poison = getenv("KH_DEBUG_POISON") != nullptr;      // test hook: destination buffers start as garbage
void free_slots(kh_table* t, KhSlots& s) {
  pool_free(t->device, s.s);
  s = kNoSlots;
}
kh_status alloc_slots(kh_table* t, uint64_t cap, KhSlots& s) {
  s = kNoSlots;
  hipError_t e = pool_alloc(t->device, std::max<uint64_t>(cap, 16) * sizeof(KhSlot), reinterpret_cast<void**>(&s.s));
  if (e != hipSuccess) { s = kNoSlots; return fail(t, KH_ERR_NOMEM, std::string("table allocation: ") + hipGetErrorString(e)); }
  s.cap = cap;
  return KH_OK;
}
kh_status fill_empty(kh_table* t, KhSlots s) {
  const uint32_t grid = (uint32_t)std::min<uint64_t>((s.cap + 255) / 256, 256 * 16);
  if (t->kind == KHK_RH) hipLaunchKernelGGL((k_fill_empty<KHK_RH>), dim3(grid), dim3(256), 0, t->stream, s);
  else hipLaunchKernelGGL((k_fill_empty<KHK_LP>), dim3(grid), dim3(256), 0, t->stream, s);
  HIPCHK(hipGetLastError());
  return KH_OK;
}
// a destination buffer of capacity `cap`.  It is NOT cleared: a re-layout writes every slot of its destination, occupied or
// empty, exactly once (the slices of the chunk workgroups tile the circular table), so clearing 16 B x capacity first
// would only add a 2 GB memset per build of a 2^27-bucket table
kh_status fresh_slots(kh_table* t, uint64_t cap, KhSlots& s) {
  if (t->spare.cap == cap && t->spare.s) { s = t->spare; t->spare = kNoSlots; }
  else {
    kh_status st = alloc_slots(t, cap, s);
    if (st != KH_OK) return st;
  }
  if (g_poison) {
    hipLaunchKernelGGL(k_poison, dim3((uint32_t)std::min<uint64_t>((cap + 255) / 256, 4096)), dim3(256), 0, t->stream, s);
    HIPCHK(hipGetLastError());
  }
  return KH_OK;
}
void retire_slots(kh_table* t, KhSlots& s) {   // keep one spare buffer for ping-pong rebuilds
  if (!s.s) return;
  if (t->spare.s) { hipStreamSynchronize(t->stream); free_slots(t, t->spare); }
  t->spare = s;
  s = kNoSlots;
}

// ---- profiling -----------------------------------------------------------------------------------
struct Launch {
  kh_table* t; const char* name; hipEvent_t a, b; bool on;
  Launch(kh_table* t_, const char* n) : t(t_), name(n), a(nullptr), b(nullptr), on(t_ && t_->prof) {
    if (on) { a = event_get(t->device); b = event_get(t->device); hipEventRecord(a, t->stream); }
  }
  ~Launch() {
    if (on) { hipEventRecord(b, t->stream); ProfRec r; r.name = name; r.a = a; r.b = b; t->recs.push_back(r); }
  }
};
void prof_collect(kh_table* t) {
  if (t->recs.empty()) return;
  hipStreamSynchronize(t->stream);
  for (auto& r : t->recs) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    event_put(t->device, r.a); event_put(t->device, r.b);
    bool found = false;
    for (auto& a : t->prof_acc) if (a.first == r.name) { a.second.first += ms; a.second.second += 1; found = true; break; }
    if (!found) t->prof_acc.push_back(std::make_pair(std::string(r.name), std::make_pair(double(ms), uint64_t(1))));
  }
  t->recs.clear();
}

#define KH_SWITCH_HASH(h, ...)                                                        \
  switch (h) {                                                                        \
    case KHH_IDENTITY: { constexpr int HASH = KHH_IDENTITY; __VA_ARGS__; } break;     \
    case KHH_MURMUR3_X86: { constexpr int HASH = KHH_MURMUR3_X86; __VA_ARGS__; } break; \
    case KHH_MURMUR3_X64: { constexpr int HASH = KHH_MURMUR3_X64; __VA_ARGS__; } break; \
    default: { constexpr int HASH = KHH_FARM; __VA_ARGS__; } break;                   \
  }
#define KH_SWITCH_KIND_HASH(k, h, ...)                                                \
  if ((k) == KHK_RH) { constexpr int KIND = KHK_RH; KH_SWITCH_HASH(h, __VA_ARGS__) }  \
  else { constexpr int KIND = KHK_LP; KH_SWITCH_HASH(h, __VA_ARGS__) }

inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t maxblocks = 256 * 16) {
  uint64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > maxblocks) g = maxblocks;
  return (uint32_t)g;
}

// stage a host array on the device (workspace) or pass a device pointer through
template <typename T>
kh_status stage_in(kh_table* t, const void* p, uint64_t n, kh_mem where, const T** out) {
  if (where == KH_MEM_DEVICE || p == nullptr) { *out = static_cast<const T*>(p); return KH_OK; }
  T* d = nullptr;
  TAKE(d, T, n);
  HIPCHK(hipMemcpyAsync(d, p, sizeof(T) * n, hipMemcpyHostToDevice, t->stream));
  *out = d;
  return KH_OK;
}

// workspace a rebuild at capacity `cap` takes (home counts + per-chunk scan arrays)
inline size_t ws_rebuild(uint64_t cap) { return size_t(cap) * 2 + (cap > KH_L ? (cap >> KH_LB) : 1) * 48 + (size_t(1) << 16); }

// ---- chunk rebuild ---------------------------------------------------------------------------------
// Lays out (live elements of t->cur, minus `erased`) U (new distinct elements) at capacity new_cap in
// a fresh buffer and makes it current.  On KH_ERR_* the current table is unchanged.
struct PreCount { uint16_t* homecnt; long long* sumA; long long* sumN; };   // chunk counts already produced by k_dedup

const bool g_disable_fused_rebuild = getenv("KH_DISABLE_FUSED_BUILD") != nullptr || getenv("KH_DISABLE_FUSED_REBUILD") != nullptr;   // test hooks

// The three speculative one-launch forms share their launch sequence: k_build_fused<KIND, HASH, SRC> over all chunks, the
// reduction of the per-chunk totals, and the tail launch that places chunk 0 once the last chunk's run-over is known.
// SRC 0: bulk build into an empty table from partition records; 1: re-layout of the current table (+ new distinct lists);
// 2: current table + partition records folded together.  Returns with the stream synchronised and, in t->hpin,
// [0] elements placed, [1] max(first-occurrence position + 1), bytes 32..: the vote words, bytes 64..: the flags.
const long long g_poll_limit = getenv("KH_DEBUG_POLL_LIMIT") ? atoll(getenv("KH_DEBUG_POLL_LIMIT")) : (1ll << 23);     // test hook: look-back time-out
struct FusedRun { unsigned long long* totals; uint32_t* flags; };
const bool g_disable_lean = getenv("KH_DISABLE_LEAN_BUILD") != nullptr;      // test hook / A-B: the general one-launch build for 12-byte records too
kh_status launch_fused(kh_table* t, int src, KhFusedParams& F, const KhSlots& nw, uint32_t PB_tail, const char* name, FusedRun* out) {
  const uint32_t nch = (uint32_t)(nw.cap >> KH_LB);
  char* blk; uint32_t* maxidx; uint64_t* ck0; uint32_t* cv0; uint16_t* hc0; long long* xc0; uint64_t* noff0; uint32_t* ncnt0;
  const size_t sz_pub = (size_t)nch * 8, sz_ctl = sz_pub + 256, sz_all = sz_ctl + (size_t)nch * 4;      // granules, control words, maxidx: ONE fill
  TAKE(blk, char, sz_all);
  maxidx = reinterpret_cast<uint32_t*>(blk + sz_ctl);
  TAKE(ck0, uint64_t, KH_DD_M); TAKE(cv0, uint32_t, KH_DD_M); TAKE(hc0, uint16_t, KH_L); TAKE(xc0, long long, 1);
  TAKE(noff0, uint64_t, 2); TAKE(ncnt0, uint32_t, 1);
  HIPCHK(hipMemsetAsync(blk, 0, sz_all, t->stream));
  F.New = nw; F.seed = t->seed;
  F.pub = reinterpret_cast<unsigned long long*>(blk);
  unsigned long long* totals = reinterpret_cast<unsigned long long*>(blk + sz_pub);   // 2 x u64 (k_fused_totals)
  F.maxidx = maxidx; F.ck0 = ck0; F.cv0 = cv0; F.homecnt0 = hc0;
  F.est = reinterpret_cast<unsigned long long*>(blk + sz_pub + 32);          // 2 x u64
  F.flags = reinterpret_cast<uint32_t*>(blk + sz_pub + 64);                  // KH_NFLAGS x u32
  F.poll_limit = g_poll_limit;
  F.R.New = nw; F.R.seed = t->seed; F.R.flags = F.flags;
  { Launch L(t, name);
    // (the benchmark case -- a duplicate-free sample, one source of 12-byte records -- has its own kernel with 17 KB less LDS: 4 workgroups per CU)
    // (... as long as a chunk of mean + 5 sigma records fits its smaller staging arrays: a table filled to 0.9 would fail it in a chunk or two
    //  of 65536 and pay for the discarded launch)
    const double mean_c = (double)F.n_total / (double)nch;
    const bool lean_fits = mean_c + 5.0 * std::sqrt(mean_c) < (double)KH_LEAN_M;
    if (src == 0 && F.nodup && F.src.rec12 == 1 && F.src.n == 1 && lean_fits && !g_disable_lean) { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_build_lean<KIND, HASH>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }
    else if (src == 0) { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_build_fused<KIND, HASH, 0>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }
    else if (src == 1) { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_build_fused<KIND, HASH, 1>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }
    else if (src == 3) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_build_fused<KHK_RH, HASH, 3>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }      // (batch erase: Robin Hood only)
    else if (src == 4) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_erase_stream<HASH>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }              // (batch erase as an ordered stream)
    else if (src == 5) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_insert_stream<HASH>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); }             // (insert into a loaded table as an ordered stream)
    else { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_build_fused<KIND, HASH, 2>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, F)); } }
  { Launch L(t, "k_fused_totals");
    hipLaunchKernelGGL(k_fused_totals, dim3(std::max<uint32_t>(1u, std::min<uint32_t>(64u, nch / 1024u))), dim3(1024), 0, t->stream, F.pub, maxidx, nch, totals); }
  { // chunk 0: placed now that the last chunk's run-over is known (one workgroup of the general placement kernel)
    Launch L(t, "k_fused_tail");
    hipLaunchKernelGGL(k_fused_tail_carry, dim3(1), dim3(64), 0, t->stream, F.pub, nch, xc0, noff0, ncnt0);      // (+ list offsets {0, 0}, list length = count field of pub[0])
    KhRebuildParams T0;
    memset(&T0, 0, sizeof(T0));
    T0.Old = kNoSlots; T0.New = nw; T0.ck = ck0; T0.cv = cv0; T0.noff = noff0; T0.ncnt = ncnt0; T0.PB = PB_tail;
    T0.seed = t->seed; T0.homecnt = hc0; T0.xcarry = xc0; T0.flags = F.flags;
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_chunk_place<KIND, HASH>), dim3(1), dim3(KH_CHUNK_THREADS), 0, t->stream, T0));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, blk + sz_pub, 128, hipMemcpyDeviceToHost, t->stream));
  t->hpin[30] = 0;
  if (t->part_overflow) HIPCHK(hipMemcpyAsync(t->hpin + 30, t->part_overflow, 4, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  out->totals = totals; out->flags = F.flags;
  return KH_OK;
}

kh_status rebuild(kh_table* t, uint64_t new_cap, const uint64_t* ck, const uint32_t* cv, const uint64_t* noff,
                  const uint32_t* ncnt, uint32_t PB, bool drop_marked, uint64_t total_after, const PreCount* pre = nullptr) {
  if (total_after > new_cap)
    return fail(t, KH_ERR_FULL, "table would hold more elements than buckets (no slot to insert into)");
  KhSlots nw;
  kh_status st = fresh_slots(t, new_cap, nw);
  if (st != KH_OK) return st;
  const uint32_t nch = new_cap > KH_L ? (uint32_t)(new_cap >> KH_LB) : 1u;
  // ---- one-launch rebuild (k_build_fused with the current table as its source): Robin Hood, same or doubled capacity.
  // Speculative like the bulk build: a chunk denser than the staging area, a carry chain or a poll time-out raise a flag,
  // and the three-kernel path below redoes the work into the same buffer.
  // Also taken with an EMPTY source table (either kind): the batch was de-duplicated on the general path and only its
  // distinct keys (ck/cv lists) have to be laid out.
  const bool from_empty = t->lsize == 0;
  if (!pre && !g_disable_fused_rebuild && new_cap >= 2 * (uint64_t)KH_L && total_after <= threshold(new_cap, 0.9f) &&
      (!noff || PB >= log2u(new_cap >> KH_LB)) &&
      (from_empty ? noff != nullptr
                  : (t->cur.cap >= 2 * (uint64_t)KH_L && (new_cap == t->cur.cap || new_cap == 2 * t->cur.cap)))) {
    const size_t keep_blk = t->blk, keep_off = t->off;
    KhFusedParams F;
    memset(&F, 0, sizeof(F));
    F.PB = PB; F.mode = KH_DEDUP_FIRST;
    F.R.Old = t->cur; F.R.drop_marked = drop_marked ? 1 : 0; F.R.ck = ck; F.R.cv = cv; F.R.noff = noff; F.R.ncnt = ncnt; F.R.PB = PB;
    if (from_empty) F.R.Old.cap = 0;     // nothing to carry over: the source scan is skipped
    FusedRun run;
    // (the parked list of chunk 0 is its own: one partition per chunk)
    { kh_status fs = launch_fused(t, 1, F, nw, log2u(new_cap >> KH_LB), "k_rebuild_fused", &run); if (fs != KH_OK) return fs; }
    const uint32_t* ff = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 64);
    bool bad = false;
    for (int i = 0; i < KH_NFLAGS; ++i) bad = bad || ff[i] != 0;
    if (!bad && t->hpin[0] == total_after) {
      KhSlots old = t->cur;
      t->cur = nw;
      retire_slots(t, old);
      t->min_load = threshold(new_cap, t->min_lf);
      t->max_load = threshold(new_cap, t->max_lf);
      return KH_OK;
    }
    if (getenv("KH_DEBUG_FUSED"))
      fprintf(stderr, "[kmerhash_amd] fused rebuild rejected: placed %llu expected %llu flags=%u %u %u %u %u\n", (unsigned long long)t->hpin[0],
              (unsigned long long)total_after, ff[0], ff[1], ff[2], ff[3], ff[4]);
    if (ff[KH_FLAG_PROBE_OVERFLOW] && !ff[KH_FLAG_FUSE_INVALID]) {   // a genuine 7-bit overflow: the general path would find the same
      KhSlots tmp = nw;
      retire_slots(t, tmp);
      return fail(t, KH_ERR_PROBE_OVERFLOW, "Robin Hood probe distance would exceed 127 (7-bit info field, hashmap_robinhood.hpp:142-144,556)");
    }
    t->blk = keep_blk; t->off = keep_off;      // scratch of the failed attempt is reused by the general path
  }
  uint16_t* homecnt; long long *sumA, *sumN, *xcarry; KhMP* ptmp; uint32_t* flags;
  if (pre) { homecnt = pre->homecnt; sumA = pre->sumA; sumN = pre->sumN; }
  else { TAKE(homecnt, uint16_t, new_cap); TAKE(sumA, long long, nch); TAKE(sumN, long long, nch); }
  TAKE(xcarry, long long, nch);
  TAKE(ptmp, KhMP, nch);
  TAKE(flags, uint32_t, KH_NFLAGS);
  HIPCHK(hipMemsetAsync(flags, 0, sizeof(uint32_t) * KH_NFLAGS, t->stream));
  KhRebuildParams P;
  P.Old = t->cur; P.drop_marked = drop_marked ? 1 : 0; P.New = nw; P.ck = ck; P.cv = cv; P.noff = noff; P.ncnt = ncnt; P.PB = PB;
  if (t->lsize == 0) P.Old.cap = 0;   // nothing to carry over: the chunk kernels skip the source scan
  P.seed = t->seed; P.homecnt = homecnt; P.sumA = sumA; P.sumN = sumN; P.xcarry = xcarry; P.flags = flags;
  if (!pre) { Launch L(t, "k_chunk_count");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_chunk_count<KIND, HASH>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, P)); }
  { Launch L(t, "k_chunk_carry");
    hipLaunchKernelGGL(k_chunk_carry, dim3(1), dim3(1024), 0, t->stream, sumA, sumN, nch, (long long)new_cap, xcarry, ptmp); }
  { Launch L(t, "k_chunk_place");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_chunk_place<KIND, HASH>), dim3(nch), dim3(KH_CHUNK_THREADS), 0, t->stream, P)); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, flags, sizeof(uint32_t) * KH_NFLAGS, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  const uint32_t* f = reinterpret_cast<const uint32_t*>(t->hpin);
  if (f[KH_FLAG_PROBE_OVERFLOW] || f[KH_FLAG_REGION_OVERFLOW] || f[KH_FLAG_COUNT_OVERFLOW] || f[KH_FLAG_INTERNAL]) {
    KhSlots tmp = nw;
    retire_slots(t, tmp);
    if (f[KH_FLAG_PROBE_OVERFLOW])
      return fail(t, KH_ERR_PROBE_OVERFLOW, "Robin Hood probe distance would exceed 127 (7-bit info field, hashmap_robinhood.hpp:142-144,556)");
    if (f[KH_FLAG_REGION_OVERFLOW])
      return fail(t, KH_ERR_PROBE_OVERFLOW, "cluster longer than one chunk (2048 slots) without an empty slot");
    if (f[KH_FLAG_COUNT_OVERFLOW])
      return fail(t, KH_ERR_PROBE_OVERFLOW, "more than 65535 keys share one home bucket");
    return fail(t, KH_ERR_HIP, "internal: de-duplication set overflow");
  }
  KhSlots old = t->cur;
  t->cur = nw;
  retire_slots(t, old);
  t->min_load = threshold(new_cap, t->min_lf);
  t->max_load = threshold(new_cap, t->max_lf);
  return KH_OK;
}

// rehash(b): hashmap_robinhood.hpp:432-464 / hashmap_linearprobe.hpp:324-349
kh_status do_rehash(kh_table* t, uint64_t b) {
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  uint64_t n = next_pow2(b);
  if (n == t->cur.cap) return KH_OK;
  if (t->kind == KHK_RH) {
    // RH copy() re-inserts through insert(), which doubles whenever size >= max_load (:530):
    // the capacity that results is the doubling rule applied to lsize distinct inserts from n.
    uint64_t c = n;
    if (t->lsize > 0) {
      if (0 >= threshold(c, t->max_lf)) c <<= 1;
      while (t->lsize > threshold(c, t->max_lf)) c <<= 1;
    }
    n = c;
    if (n == t->cur.cap) return KH_OK;
  } else if (t->lsize > n) {
    return fail(t, KH_ERR_FULL, "ERROR: did not find any place to insert.  should not have happend (hashmap_linearprobe.hpp:408)");
  }
  { kh_status ps = arena_prepare(t, ws_rebuild(n)); if (ps != KH_OK) return ps; }
  kh_status st = rebuild(t, n, nullptr, nullptr, nullptr, nullptr, 0, false, t->lsize);
  arena_consolidate(t);
  return st;
}
kh_status do_reserve(kh_table* t, uint64_t n) {   // :421-426 / :313-318
  if (n > t->max_load) return do_rehash(t, static_cast<uint64_t>(static_cast<float>(n) / t->max_lf));
  return KH_OK;
}

// ---- radix partition of a batch by bit-reversed chunk id ------------------------------------------
struct Partitioned {
  ulonglong2* rec;                              // records grouped by partition: (key, idx<<32|val)
  uint64_t* part_off;                           // [nparts+1]
  uint32_t PB, nparts;
  ulonglong2* spare;                            // the other record buffer (free for outputs)
  uint64_t slot;                                // != 0: histogram-free layout: partition q = rec[q * slot, cursor[q])
  const unsigned long long* cursor;
  uint32_t* overflow;                           // device flag: a partition outgrew its slot (the batch must be redone with exact offsets)
  int rec12;                                    // 0: 16-byte records; 1: 12-byte (key, value) (histogram-free layout only); 2: 8-byte keys (counting insert)
};
// slot of a histogram-free partition with mean m records: m + 7 sigma (hashed keys: Poisson) + a little
// (vf: variance of a partition's fill over its mean -- 1 for distinct keys (Poisson in the records); a batch with duplicates fills its partitions
//  key by key, E[m^2] / E[m] records at a time: insert_core estimates that factor from the duplicate sample)
inline uint64_t slack_slot(double mean, double vf = 1.0) { return (uint64_t)(mean + 7.0 * std::sqrt(mean * vf) + 16.0); }
const bool g_disable_slack = getenv("KH_DISABLE_SLACK_PARTITION") != nullptr;      // test hook: exact offsets always
// eight consecutive partition tiles per XCD (blocks b, b + 8, ... share one): the adjacent output runs of consecutive tiles meet in
// one L2 (-3 % scatter time, measured A/B); KH_DISABLE_XCD_SWIZZLE=1 turns it off
const int g_xcd_swizzle = getenv("KH_DISABLE_XCD_SWIZZLE") ? 0 : 1;
// records the two buffers of partition_batch must hold
inline uint64_t part_buffer_records(uint64_t n, uint32_t PB, bool allow_slack, double vf = 1.0) {
  if (!allow_slack || g_disable_slack || PB <= 11 || n < (uint64_t(256) << PB)) return n;
  return (slack_slot((double)n / (double)(uint64_t(1) << PB), vf) << PB) + KH_PART_TILE;        // slots + the dump area of the scatter
}

// Partitions n input pairs into `fin` (n records); `tmp` (n records) is scratch for the first of two passes.  idx_base =
// stream position of the first pair (pairs fed before it in a streamed insert).  Asynchronous on the table's stream.
kh_status partition_batch(kh_table* t, const char* kbase, uint32_t kstride, const char* vbase, uint32_t vstride,
                          uint32_t vconst, uint64_t n, uint64_t idx_base, uint32_t PB, ulonglong2* tmp, ulonglong2* fin, Partitioned& out,
                          bool allow_slack = false, int rec12 = 0, const SlackShared* shared = nullptr, bool force_slack = false, double vf = 1.0) {
  const uint32_t nparts = 1u << PB;
  out.slot = 0; out.cursor = nullptr; out.overflow = nullptr; out.rec12 = 0;
  // (force_slack: the caller has sized tmp / fin for the fixed slots itself -- (slack_slot(n / 2^PB) << PB) + KH_PART_TILE records each --
  //  whatever n is: the erase keys of a batch erase, where the histogram sweep would cost more than the partition)
  if (shared || force_slack || part_buffer_records(n, PB, allow_slack, vf) != n) {
    // ---- histogram-free two-pass partition: hashed keys fill the 2^PB partitions evenly, so every partition
    // gets a fixed slot of mean + 7 sigma records and the passes reserve space with their cursors alone: no histogram sweep
    // over the keys (0.33 ms per 1e8), no offset scan.  Level-1 buckets are the unions of their partitions' slots.
    const uint32_t B1 = (PB + 1) / 2, B2 = PB - B1, nb1 = 1u << B1, nb2 = 1u << B2;
    // (a piece of a streamed insert: its level-1 buckets are sized for the piece, the final slots -- shared -- for the whole batch)
    const uint64_t slot = shared ? shared->slot : slack_slot((double)n / (double)nparts, vf);
    const uint64_t slot1 = shared ? slack_slot((double)n / (double)nb1) : slot * nb2;
    unsigned long long *cur1, *cur2; uint64_t* starts; uint32_t* ovf;
    TAKE(cur1, unsigned long long, nb1);
    if (shared) { cur2 = shared->cur2; starts = shared->starts; ovf = shared->ovf; }
    else { TAKE(cur2, unsigned long long, nparts); TAKE(starts, uint64_t, (size_t)nparts + 1); TAKE(ovf, uint32_t, 1); }
    // the second pass cuts every level-1 slot into tiles by arithmetic (no tile list, see KhPartParams::slot_in) -- up to mean + 9 sigma of a
    // level-1 bucket's fill, which is well inside the slot (the union of its partitions' slots: 256 x 7 sigma of THEIR fill); a bucket
    // that holds more raises the overflow flag like a slot that runs over
    const double mean1 = (double)n / (double)nb1;
    const uint64_t fill_cap = std::min<uint64_t>(slot1, (uint64_t)(mean1 + 9.0 * std::sqrt(mean1 * vf) + 16.0));
    const uint32_t tps = (uint32_t)((fill_cap + KH_PART_TILE - 1) / KH_PART_TILE);
    const uint32_t max_tiles = nb1 * tps;
    if (shared) hipLaunchKernelGGL(k_init_cursors, dim3((nb1 + 255) / 256), dim3(256), 0, t->stream, cur1, (uint64_t*)nullptr, (uint64_t)nb1, slot1);
    else hipLaunchKernelGGL(k_init_cursors2, dim3((nparts + 256) / 256), dim3(256), 0, t->stream, cur1, (uint64_t)nb1, slot1, cur2, starts, (uint64_t)nparts, slot, ovf);
    KhPartParams P;
    memset(&P, 0, sizeof(P));
    P.idx_base = idx_base;
    P.xcd_swizzle = g_xcd_swizzle;
    P.kbase = kbase; P.kstride = kstride; P.vbase = vbase; P.vstride = vstride; P.vconst = vconst; P.n = n;
    P.ntiles = (uint32_t)((n + KH_PART_TILE - 1) / KH_PART_TILE);
    P.seed = t->seed; P.PB = PB; P.shift = B2; P.nb = nb1; P.cursor = cur1; P.orec = tmp; P.slot = slot1; P.overflow = ovf; P.dump = slot1 * nb1;
    { Launch L(t, "k_part_scatter");
      if (rec12 == 1) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 1>), dim3(P.ntiles), dim3(KH_PART_THREADS), ((nb1 + 1u) & ~1u) * 8 + nb1 * 8, t->stream, P)); }
      else if (rec12 == 2) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 2>), dim3(P.ntiles), dim3(KH_PART_THREADS), ((nb1 + 1u) & ~1u) * 8 + nb1 * 8, t->stream, P)); }
      else { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 0>), dim3(P.ntiles), dim3(KH_PART_THREADS), ((nb1 + 1u) & ~1u) * 8 + nb1 * 8, t->stream, P)); } }
    KhPartParams Q = P;
    Q.kbase = nullptr; Q.kstride = 0; Q.vbase = nullptr; Q.vstride = 0; Q.rec_in = tmp;
    Q.tiles = nullptr; Q.ntiles_dev = nullptr; Q.ntiles = max_tiles; Q.slot_in = slot1; Q.cur_in = cur1; Q.tps = tps;
    Q.shift = 0; Q.nb = nb2; Q.cursor = cur2; Q.orec = fin; Q.slot = slot; Q.dump = slot * nparts;
    { Launch L(t, "k_part_scatter");
      if (rec12 == 1) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 1>), dim3(max_tiles), dim3(KH_PART_THREADS), ((nb2 + 1u) & ~1u) * 8 + nb2 * 8, t->stream, Q)); }
      else if (rec12 == 2) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 2>), dim3(max_tiles), dim3(KH_PART_THREADS), ((nb2 + 1u) & ~1u) * 8 + nb2 * 8, t->stream, Q)); }
      else { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 0>), dim3(max_tiles), dim3(KH_PART_THREADS), ((nb2 + 1u) & ~1u) * 8 + nb2 * 8, t->stream, Q)); } }
    HIPCHK(hipGetLastError());
    out.rec = fin; out.part_off = starts; out.PB = PB; out.nparts = nparts; out.spare = tmp;
    out.slot = slot; out.cursor = cur2; out.overflow = ovf; out.rec12 = rec12;
    return KH_OK;
  }
  if (rec12 == 1) return fail(t, KH_ERR_HIP, "internal: record kind not available with exact offsets");
  ulonglong2* ar = nullptr; ulonglong2* br = fin;
  const uint32_t B1 = PB <= 11 ? PB : (PB + 1) / 2, B2 = PB - B1;
  const uint32_t nb1 = 1u << B1, nb2 = 1u << B2;
  ar = B2 ? tmp : fin;                          // a single pass writes the final buffer directly
  uint32_t* counts1; uint64_t* off1; unsigned long long* cur1;
  TAKE(counts1, uint32_t, nb1); TAKE(off1, uint64_t, nb1 + 1); TAKE(cur1, unsigned long long, nb1);
  KhPartParams P;
  memset(&P, 0, sizeof(P));
  P.idx_base = idx_base;
  P.xcd_swizzle = g_xcd_swizzle;
  P.kbase = kbase; P.kstride = kstride; P.vbase = vbase; P.vstride = vstride; P.vconst = vconst; P.rec_in = nullptr; P.n = n;
  P.tiles = nullptr; P.ntiles_dev = nullptr; P.ntiles = (uint32_t)((n + KH_PART_TILE - 1) / KH_PART_TILE);
  P.seed = t->seed; P.PB = PB; P.shift = B2; P.nb = nb1; P.counts = counts1; P.cursor = cur1;
  P.orec = ar;
  KhTile* tiles = nullptr; uint32_t* ntiles_dev = nullptr; uint32_t* counts2 = nullptr; uint64_t* off2 = nullptr; unsigned long long* cur2 = nullptr;
  const uint32_t max_tiles = (uint32_t)(n / KH_PART_TILE) + nb1 + 1;
  if (B2 > 0) {
    TAKE(tiles, KhTile, max_tiles); TAKE(ntiles_dev, uint32_t, 1);
    TAKE(counts2, uint32_t, nparts); TAKE(off2, uint64_t, nparts + 1); TAKE(cur2, unsigned long long, nparts);
    HIPCHK(hipMemsetAsync(counts2, 0, sizeof(uint32_t) * nparts, t->stream));
  }
  const bool full_hist = B2 > 0 && PB <= 18;     // 2^16 16-bit bins = 128 KB of LDS per sweep; up to four sweeps (slices of the id space)
  if (full_hist) {
    // one sweep over the keys gives the histogram of the full partition id; both levels' offsets follow from one scan
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, t->device);
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ncu, (n + 4095) / 4096));
    const uint32_t slice_bits = PB > 16 ? PB - 16 : 0;
    { Launch L(t, "k_part_hist");
      const size_t smem = (size_t)((nparts >> slice_bits) / 2) * 4;
      KH_SWITCH_HASH(t->hash,
                     if (smem > 65536) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_hist_full<HASH>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                     for (uint32_t sl = 0; sl < (1u << slice_bits); ++sl)
                       hipLaunchKernelGGL((k_part_hist_full<HASH>), dim3(grid), dim3(KH_FULLHIST_THREADS), smem, t->stream,
                                          kbase, kstride, n, t->seed, PB, sl, slice_bits, counts2)); }
    { Launch L(t, "k_scan");
      hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, t->stream, counts2, (uint64_t)nparts, off2); }
    hipLaunchKernelGGL(k_seg_offsets, dim3((nb1 + 256) / 256), dim3(256), 0, t->stream, off2, nb1, nb2, off1, cur1);
  } else {
    HIPCHK(hipMemsetAsync(counts1, 0, sizeof(uint32_t) * nb1, t->stream));
    const uint32_t hist_grid = std::min<uint32_t>(P.ntiles, 1024);
    { Launch L(t, "k_part_hist");
      KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_hist<HASH>), dim3(hist_grid), dim3(KH_PART_THREADS), nb1 * 4, t->stream, P)); }
    { Launch L(t, "k_scan");
      hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, t->stream, counts1, (uint64_t)nb1, off1); }
    HIPCHK(hipMemcpyAsync(cur1, off1, sizeof(uint64_t) * nb1, hipMemcpyDeviceToDevice, t->stream));
  }
  { Launch L(t, "k_part_scatter");
    if (rec12 == 2) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 2>), dim3(P.ntiles), dim3(KH_PART_THREADS), ((nb1 + 1u) & ~1u) * 8 + nb1 * 8, t->stream, P)); }
    else { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 0>), dim3(P.ntiles), dim3(KH_PART_THREADS), ((nb1 + 1u) & ~1u) * 8 + nb1 * 8, t->stream, P)); } }
  out.rec12 = rec12 == 2 ? 2 : 0;
  if (B2 == 0) {
    out.rec = ar; out.part_off = off1; out.PB = PB; out.nparts = nparts;
    out.spare = tmp;
    HIPCHK(hipGetLastError());
    return KH_OK;
  }
  // second pass inside every first-pass segment
  { Launch L(t, "k_make_tiles");
    hipLaunchKernelGGL(k_make_tiles, dim3(1), dim3(1024), 0, t->stream, off1, nb1, tiles, ntiles_dev); }
  KhPartParams Q = P;
  Q.kbase = nullptr; Q.kstride = 0;
  Q.vbase = nullptr; Q.vstride = 0; Q.rec_in = ar;
  Q.tiles = tiles; Q.ntiles_dev = ntiles_dev; Q.ntiles = max_tiles;
  Q.shift = 0; Q.nb = nb2; Q.counts = counts2; Q.cursor = cur2; Q.orec = br;
  if (!full_hist) {
    { Launch L(t, "k_part_hist");
      if (rec12 == 2) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_hist<HASH, true>), dim3(std::min<uint32_t>(max_tiles, 1024)), dim3(KH_PART_THREADS), nb2 * 4, t->stream, Q)); }
      else { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_hist<HASH>), dim3(std::min<uint32_t>(max_tiles, 1024)), dim3(KH_PART_THREADS), nb2 * 4, t->stream, Q)); } }
    { Launch L(t, "k_scan");
      hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, t->stream, counts2, (uint64_t)nparts, off2); }
  }
  HIPCHK(hipMemcpyAsync(cur2, off2, sizeof(uint64_t) * nparts, hipMemcpyDeviceToDevice, t->stream));
  { Launch L(t, "k_part_scatter");
    if (rec12 == 2) { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 2>), dim3(max_tiles), dim3(KH_PART_THREADS), ((nb2 + 1u) & ~1u) * 8 + nb2 * 8, t->stream, Q)); }
    else { KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_scatter<HASH, 0>), dim3(max_tiles), dim3(KH_PART_THREADS), ((nb2 + 1u) & ~1u) * 8 + nb2 * 8, t->stream, Q)); } }
  HIPCHK(hipGetLastError());
  out.rec = br; out.part_off = off2; out.PB = PB; out.nparts = nparts;
  out.spare = tmp;
  return KH_OK;
}

// capacity after `ncalls` insert() calls that add `dnew` new keys, the last of which first occurs at
// call index `last_first` (hashmap_robinhood.hpp:530 / hashmap_linearprobe.hpp:439: ONE doubling per
// call made while size >= max_load, duplicates included).  Precondition: at most one doubling is
// pending at the start (size < max_load(2*cap)); the caller peels single calls otherwise.
uint64_t capacity_after(const kh_table* t, uint64_t cap, uint64_t lsize, uint64_t ncalls, uint64_t dnew, uint64_t last_first) {
  if (ncalls == 0) return cap;
  uint64_t c = cap;
  if (lsize >= threshold(c, t->max_lf)) c <<= 1;                 // call 0
  const uint64_t fin = lsize + dnew;
  while (fin > threshold(c, t->max_lf)) c <<= 1;                 // thresholds passed on the way: a later new key follows
  if (dnew > 0 && fin == threshold(c, t->max_lf) && last_first + 1 < ncalls) c <<= 1;   // reached exactly, a later call follows
  return c;
}

enum { INS_FIRST = 0, INS_UPDATE = 1, INS_PLUS = 2 };
// Pairs one internal pass takes.  A batch is the reference's SEQUENCE of insert() calls (one doubling decision per call), so cutting it
// into consecutive passes changes nothing observable; what it bounds is the workspace -- ~58 B per pair of a pass: a k-mer counter's
// file batch of 1.7e9 k-mers would otherwise ask for a 100 GB arena, whose hipMalloc alone costs 0.7 s (measured, scripts/kc_scale_probe.py).
// 2^30: a second pass is an insert into a LOADED table -- a re-layout of all of it (19 ms at 2^31 buckets) -- so 10^9 keys in one call stay one pass
// (scripts/scale_1e9.py: 40.9 ms with passes of 2^29, against the partition + build of a single pass).
const uint64_t g_max_pass = getenv("KH_MAX_PASS_RECORDS") ? std::max<uint64_t>(1024, std::min<uint64_t>(strtoull(getenv("KH_MAX_PASS_RECORDS"), nullptr, 10), 0xFFFFFFF0ull))
                                                          : (uint64_t(1) << 30);
bool g_disable_fused = getenv("KH_DISABLE_FUSED_BUILD") != nullptr;   // test hook: force the general path

// second half of an insert: the n pairs have been partitioned (one source per feed); de-dup, capacity decision, build
const kh_status KH_RETRY_EXACT = static_cast<kh_status>(100);      // internal: a histogram-free partition overflowed a slot
kh_status insert_finish(kh_table* t, KhSrcSet S, uint64_t n, uint32_t PB, uint64_t cap_u, int mode, uint64_t forced_cap,
                        ulonglong2* spare, uint64_t* n_new_out, uint64_t list_cap = 0);

// does the one-launch bulk build (k_build_fused, SRC 0) apply, and may it skip the LDS fold after a duplicate-free sample?
inline bool fused_build_applies(const kh_table* t, uint64_t cap_u, uint32_t PB) {
  return t->lsize == 0 && cap_u >= 2 * (uint64_t)KH_L && t->max_lf <= 0.9f && PB == log2u(cap_u >> KH_LB) && !g_disable_fused;
}
inline bool nodup_build_applies(const kh_table* t, uint64_t cap_u, uint32_t PB, int mode) {
  return fused_build_applies(t, cap_u, PB) && t->batch_nodup && mode != INS_UPDATE && !getenv("KH_DISABLE_NODUP");
}

// core of insert/update for one batch of device-resident input (n < 2^32 - 16)
kh_status insert_core(kh_table* t, const char* kbase, uint32_t kstride, const char* vbase, uint32_t vstride,
                      uint64_t n, int mode, uint64_t forced_cap, uint64_t* n_new_out) {
  *n_new_out = 0;
  if (n == 0) return KH_OK;
  const uint64_t cap_u = forced_cap ? forced_cap : capacity_after(t, t->cur.cap, t->lsize, n, n, n - 1);
  const uint32_t PB = cap_u > KH_L ? log2u(cap_u >> KH_LB) : 0u;
  if (PB > 22) return fail(t, KH_ERR_UNSUPPORTED, "batch would need more than 2^22 partitions");
  const size_t keep_blk = t->blk, keep_off = t->off;
  int first_attempt = 0;
  bool dup_heavy = false;
  double vf = 1.0;            // variance factor of the histogram-free slots (duplicates: see below)
  if (part_buffer_records(n, PB, true) != n) {
    // histogram-free partition only for batches a sample finds (nearly) free of duplicates
    unsigned long long* sset; uint32_t* dups;
    TAKE(sset, unsigned long long, KH_SAMPLE_SET); TAKE(dups, uint32_t, 1);
    HIPCHK(hipMemsetAsync(sset, 0, sizeof(unsigned long long) * KH_SAMPLE_SET, t->stream));
    HIPCHK(hipMemsetAsync(dups, 0, 4, t->stream));
    { Launch L(t, "k_sample_dups");
      hipLaunchKernelGGL(k_sample_dups, dim3(KH_SAMPLE_N / 256), dim3(256), 0, t->stream, kbase, kstride, n, sset, dups); }
    HIPCHK(hipMemcpyAsync(t->hpin + 31, dups, 4, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    // ANY duplicate among the 65536 sampled keys sends the batch down the exact path: the fixed slots are sized for partition counts
    // that are Poisson in the number of RECORDS, and already a mean multiplicity of 1.5 (a k-mer counter's file batch at low coverage:
    // ~7 duplicate pairs in the sample, birthday-bound) widens them enough to overflow a slot somewhere among 2^19 partitions --
    // measured: every such batch paid a discarded 7.5 ms partition attempt; the histogram sweep of the exact path costs 2.5 ms
    // ... UNLESS the slots are sized for the duplicates: a partition then fills key by key, E[m^2] / E[m] records at a time, and that factor
    // follows from the sample -- d duplicates among S sampled keys of n say sum m (m - 1) / n = 2 n d / S^2 (the reference benchmark's x5.5 input:
    // 129 duplicates -> 7.0; a k-mer counter's batch at coverage 1.5: 7 -> 2.5).  Taken at its upper end (d + 3 sqrt(d) + 3) and only while the
    // slot stays below twice the mean; an overflow after all repeats the batch with exact offsets as before.
    const uint32_t dsample = (uint32_t)t->hpin[31];
    if (dsample >= 1u) {
      dup_heavy = true;
      const double d_ub = (double)dsample + 3.0 * std::sqrt((double)dsample) + 3.0;
      const double f = 1.0 + 2.0 * (double)n * d_ub / ((double)KH_SAMPLE_N * (double)KH_SAMPLE_N);
      const double mean = (double)n / (double)(uint64_t(1) << PB);
      if (!getenv("KH_DISABLE_DUP_SLACK") && (double)slack_slot(mean, f) <= 2.0 * mean) vf = f;
      else first_attempt = 1;
    }
    t->batch_nodup = dsample == 0u;
    t->blk = keep_blk; t->off = keep_off;
  } else first_attempt = 1;
  for (int attempt = first_attempt; attempt < 2; ++attempt) {
    const bool slack = attempt == 0;          // histogram-free first; exact offsets if a partition outgrew its slot (skewed keys)
    const uint64_t m = part_buffer_records(n, PB, slack, vf);
    if (!slack && attempt == 1) { t->blk = keep_blk; t->off = keep_off; }
    // no duplicate in the sample and an empty table ahead of the one-launch build: the records need no stream position
    // (12 bytes instead of 16); whatever that build cannot take (a duplicate after all, a dense chunk) repeats the batch
    const bool rec12 = slack && t->batch_nodup && nodup_build_applies(t, cap_u, PB, mode);
    // a counting insert (Reducer = std::plus, every value 1) of a batch the sample found heavy in duplicates -- the k-mer counter's
    // batches: the records are the keys alone (8 bytes: neither value nor position is needed), and the general path takes them
    // directly (the one-launch forms speculate on few duplicates: hopeless here)
    const bool rec8 = !rec12 && dup_heavy && mode == INS_PLUS && vbase == nullptr && !getenv("KH_DISABLE_REC8");
    const int kind = rec12 ? 1 : rec8 ? 2 : 0;
    ulonglong2 *tmp, *fin, *spare;
    { char *a, *b; const size_t rb = kind == 1 ? sizeof(KhRec12) : kind == 2 ? sizeof(KhRec8) : sizeof(ulonglong2);
      TAKE(a, char, m * rb); TAKE(b, char, m * rb);
      tmp = reinterpret_cast<ulonglong2*>(a); fin = reinterpret_cast<ulonglong2*>(b); spare = tmp;
      if (kind == 2) { char* c; TAKE(c, char, m * 12); spare = reinterpret_cast<ulonglong2*>(c); } }     // (lists of distinct keys + counts: 12 bytes per entry)
    Partitioned R;
    kh_status st = partition_batch(t, kbase, kstride, vbase, vstride, mode == INS_PLUS ? 1u : 0u, n, 0, PB, tmp, fin, R, slack, kind, nullptr, false, vf);
    if (st != KH_OK) return st;
    KhSrcSet S;
    memset(&S, 0, sizeof(S));
    S.rec[0] = R.rec; S.off[0] = R.part_off; S.n = 1; S.merged_off = R.part_off; S.slot[0] = R.slot; S.cur[0] = R.cursor;
    S.rec12 = (uint32_t)R.rec12;
    t->part_overflow = R.overflow; t->batch_vf = vf;
    st = insert_finish(t, S, n, PB, cap_u, mode, forced_cap, spare, n_new_out, m);
    t->part_overflow = nullptr; t->batch_nodup = false; t->batch_vf = 1.0;
    if (st != KH_RETRY_EXACT) return st;
  }
  return fail(t, KH_ERR_HIP, "internal: exact partition reported a slot overflow");
}

kh_status insert_finish(kh_table* t, KhSrcSet S, uint64_t n, uint32_t PB, uint64_t cap_u, int mode, uint64_t forced_cap,
                        ulonglong2* spare, uint64_t* n_new_out, uint64_t list_cap) {
  *n_new_out = 0;
  if (list_cap < n) list_cap = n;       // entries the per-partition output lists span (slots of a histogram-free partition: > n)
  kh_status st = KH_OK;
  const uint32_t nparts = 1u << PB;
  struct { uint32_t nparts; } R; R.nparts = nparts;
  // ---- fused bulk build: empty table, moderate load factor, at least two chunks.  Speculates that the capacity the
  // reference's rule yields equals cap_u (true when the batch holds few duplicates); otherwise falls through.
  // (not for a batch whose sample says that a key comes E[m^2]/E[m] >= 2 times: the build speculates that every pair is a new key, its first 64
  //  chunks would vote it down -- 0.11 ms of launches for the reference benchmark's own input; a batch that is mostly distinct after all is
  //  merely built by the general path)
  if (fused_build_applies(t, cap_u, PB) && S.rec12 != 2 && !(t->batch_vf >= 2.0)) {
    const uint32_t nch = (uint32_t)(cap_u >> KH_LB);
    KhSlots nw;
    st = fresh_slots(t, cap_u, nw);
    if (st != KH_OK) return st;
    KhFusedParams F;
    memset(&F, 0, sizeof(F));
    F.src = S; F.PB = PB;
    F.mode = mode == INS_PLUS ? KH_DEDUP_PLUS : KH_DEDUP_FIRST;
    F.base_size = 0;
    F.n_total = n;
    F.nodup = nodup_build_applies(t, cap_u, PB, mode) ? 1 : 0;
    // giving up early only makes sense if a smaller capacity is possible at all (an insert never shrinks the table)
    F.half_max_load = (cap_u >> 1) >= t->cur.cap ? threshold(cap_u >> 1, t->max_lf) : 0;
    FusedRun run;
    { kh_status fs = launch_fused(t, 0, F, nw, PB, "k_build_fused", &run); if (fs != KH_OK) return fs; }
    if (t->part_overflow && (uint32_t)t->hpin[30]) { retire_slots(t, nw); return KH_RETRY_EXACT; }
    unsigned long long* totals = run.totals;
    const uint64_t fd = t->hpin[0];
    // (12-byte records: accepted only if all n keys were distinct, so the last call is the last first occurrence)
    const uint64_t flast = (mode == INS_PLUS || S.rec12) ? n - 1 : (t->hpin[1] ? t->hpin[1] - 1 : 0);
    const uint32_t* ff = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 64);
    bool bad = S.rec12 && fd != n;
    for (int i = 0; i < KH_NFLAGS; ++i) bad = bad || ff[i] != 0;
    if (!bad && capacity_after(t, t->cur.cap, t->lsize, n, fd, flast) == cap_u) {
      KhSlots old = t->cur;
      t->cur = nw;
      retire_slots(t, old);
      t->min_load = threshold(cap_u, t->min_lf);
      t->max_load = threshold(cap_u, t->max_lf);
      t->lsize = fd;
      *n_new_out = fd;
      if (mode == INS_UPDATE) {
        KhDedupParams A;
        memset(&A, 0, sizeof(A));
        A.src = S; A.T = t->cur; A.seed = t->seed; A.table_empty = 0; A.mode = KH_DEDUP_LAST;
        uint32_t* cn; TAKE(cn, uint32_t, R.nparts);
        A.cnt_new = cn; A.max_idx_plus1 = totals; A.flags = F.flags; A.count_cap = 0; A.PB = PB;
        Launch L(t, "k_dedup_assign");
        KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, A));
        HIPCHK(hipGetLastError());
      }
      return KH_OK;
    }
    if (getenv("KH_DEBUG_FUSED"))
      fprintf(stderr, "[kmerhash_amd] fused build rejected: n=%llu dnew=%llu cap_u=%llu cap_rule=%llu flags=%u %u %u %u %u est=%u/%u/%u\n", (unsigned long long)n,
              (unsigned long long)fd, (unsigned long long)cap_u, (unsigned long long)capacity_after(t, t->cur.cap, t->lsize, n, fd, flast), ff[0], ff[1], ff[2], ff[3], ff[4],
              reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 32)[1], reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 32)[0],
              reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 32)[2]);
    retire_slots(t, nw);      // speculation failed (duplicates, skew): the buffer becomes the spare, general path below
    if (S.rec12 == 1) return KH_RETRY_EXACT;      // (the general path needs the stream positions: repartition with 16-byte records)
  }
  if (S.rec12 == 1) return fail(t, KH_ERR_HIP, "internal: 12-byte records outside the bulk build");
  // ---- fused insert into a NON-empty Robin Hood table: every chunk stages its current elements (home from the info byte)
  // next to the batch's records of the same chunk, folds them together (an element of the table beats every record), and
  // lays the chunk out -- no membership probes at random into HBM (k_dedup), no separate re-layout.  Speculates, like the
  // bulk build, that the capacity the reference's rule yields equals cap_u.
  if (t->lsize > 0 && S.rec12 != 2 && (mode == INS_FIRST || mode == INS_PLUS) && !forced_cap && !g_disable_fused_rebuild &&
      t->cur.cap >= 2 * (uint64_t)KH_L && (cap_u == t->cur.cap || cap_u == 2 * t->cur.cap) && t->max_lf <= 0.9f &&
      PB == log2u(cap_u >> KH_LB) && t->lsize + n <= threshold(cap_u, 0.92f)) {
    const size_t keep_blk = t->blk, keep_off = t->off;
    const uint32_t nch = (uint32_t)(cap_u >> KH_LB);
    KhSlots nw;
    st = fresh_slots(t, cap_u, nw);
    if (st != KH_OK) return st;
    KhFusedParams F;
    memset(&F, 0, sizeof(F));
    F.src = S; F.PB = PB;
    F.mode = mode == INS_PLUS ? KH_DEDUP_PLUS : KH_DEDUP_FIRST;
    F.n_total = n;
    F.base_size = t->lsize;
    // giving up early only makes sense if the smaller capacity is possible at all (an insert never shrinks the table)
    F.half_max_load = (cap_u >> 1) >= t->cur.cap ? threshold(cap_u >> 1, t->max_lf) : 0;
    F.R.Old = t->cur; F.R.PB = PB;
    FusedRun run;
    // (Robin Hood at equal capacity, one source of 16-byte records, a chunk's records fit two per lane: the ordered-stream kernel; else the staging form)
    const bool ordered = t->kind == KHK_RH && cap_u == t->cur.cap && S.n == 1 && S.rec12 == 0 &&
                         (S.slot[0] ? S.slot[0] <= KH_IS_MAXR : n / (uint64_t(1) << PB) <= KH_IS_MAXR / 2) && !getenv("KH_DISABLE_ORDERED_INSERT");
    { kh_status fs = launch_fused(t, ordered ? 5 : 2, F, nw, PB, "k_insert_fused", &run); if (fs != KH_OK) return fs; }
    if (t->part_overflow && (uint32_t)t->hpin[30]) { retire_slots(t, nw); return KH_RETRY_EXACT; }
    const uint64_t total = t->hpin[0];
    const uint64_t fd = total >= t->lsize ? total - t->lsize : 0;
    const uint64_t flast = mode == INS_PLUS ? n - 1 : (t->hpin[1] ? t->hpin[1] - 1 : 0);
    const uint32_t* ff = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 64);
    bool bad = total < t->lsize;
    for (int i = 0; i < KH_NFLAGS; ++i) bad = bad || ff[i] != 0;
    if (!bad && capacity_after(t, t->cur.cap, t->lsize, n, fd, flast) == cap_u) {
      KhSlots old = t->cur;
      t->cur = nw;
      retire_slots(t, old);
      t->min_load = threshold(cap_u, t->min_lf);
      t->max_load = threshold(cap_u, t->max_lf);
      t->lsize = total;
      *n_new_out = fd;
      return KH_OK;
    }
    if (getenv("KH_DEBUG_FUSED"))
      fprintf(stderr, "[kmerhash_amd] fused insert rejected: n=%llu size=%llu total=%llu cap_u=%llu cap_rule=%llu flags=%u %u %u %u %u\n", (unsigned long long)n,
              (unsigned long long)t->lsize, (unsigned long long)total, (unsigned long long)cap_u,
              (unsigned long long)capacity_after(t, t->cur.cap, t->lsize, n, fd, flast), ff[0], ff[1], ff[2], ff[3], ff[4]);
    retire_slots(t, nw);
    t->blk = keep_blk; t->off = keep_off;
  }
  uint32_t* cnt_new; uint64_t* noff; unsigned long long* scal; uint32_t* flags;
  TAKE(cnt_new, uint32_t, R.nparts); TAKE(noff, uint64_t, R.nparts + 1); TAKE(scal, unsigned long long, 4);
  TAKE(flags, uint32_t, KH_NFLAGS);
  HIPCHK(hipMemsetAsync(scal, 0, sizeof(unsigned long long) * 4, t->stream));
  HIPCHK(hipMemsetAsync(flags, 0, sizeof(uint32_t) * KH_NFLAGS, t->stream));
  KhDedupParams D;
  D.src = S;
  // outputs go into the scratch record buffer (16 B per input record): keys in its first half, values behind them
  D.nk = reinterpret_cast<uint64_t*>(spare); D.nv = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(spare) + list_cap * 8);
  D.cnt_new = cnt_new; D.max_idx_plus1 = scal;
  // speculate that the capacity decided below equals cap_u (true whenever the batch holds few duplicates): then the
  // de-dup kernel already produces the chunk counts and k_chunk_count is skipped
  PreCount pre; pre.homecnt = nullptr; pre.sumA = nullptr; pre.sumN = nullptr;
  const bool fuse = t->lsize == 0;
  if (fuse) {
    TAKE(pre.homecnt, uint16_t, cap_u); TAKE(pre.sumA, long long, R.nparts); TAKE(pre.sumN, long long, R.nparts);
  }
  D.count_cap = fuse ? cap_u : 0; D.PB = PB; D.homecnt = pre.homecnt; D.sumA = pre.sumA; D.sumN = pre.sumN;
  D.T = t->cur; D.seed = t->seed; D.table_empty = t->lsize == 0 ? 1 : 0; D.mode = mode == INS_PLUS ? KH_DEDUP_PLUS : KH_DEDUP_FIRST; D.flags = flags;
  D.xcd_group = 0;
  // Reducer = std::plus into a non-empty table: k_dedup only LISTS the sums of the keys the table already holds; they are added
  // (k_apply_plus) once nothing can discard this attempt any more -- a histogram-free partition that overflowed repeats the whole
  // batch, and a repeatable streamed insert promises "table unchanged" with KH_ERR_RETRY
  const bool plus_live = mode == INS_PLUS && t->lsize > 0;
  D.cnt_upd = nullptr;
  if (plus_live) TAKE(D.cnt_upd, uint32_t, R.nparts);
  // with exact partition offsets nothing can discard the attempt before the re-layout: the sums are added inside k_dedup (its
  // membership probes have the slot in hand) and the list only serves to take them back should the re-layout fail
  // ... and so they are behind a histogram-free partition once its overflow flag has been read clean: the partition is complete in stream
  // order, so the flag is final -- one small copy + wait (tens of microseconds) instead of a k_apply_plus pass at random over the table
  // (7 ms per 4.5e8-k-mer batch of the k-mer counter, measured)
  bool overflow_clean = t->part_overflow == nullptr;
  if (plus_live && !overflow_clean && !getenv("KH_DISABLE_PLUS_IMMEDIATE")) {
    t->hpin[30] = 0;
    HIPCHK(hipMemcpyAsync(t->hpin + 30, t->part_overflow, 4, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    if ((uint32_t)t->hpin[30]) return KH_RETRY_EXACT;
    overflow_clean = true;
  }
  D.plus_immediate = (plus_live && overflow_clean && !getenv("KH_DISABLE_PLUS_IMMEDIATE")) ? 1 : 0;
  if (t->lsize > 0 && t->cur.cap > KH_L && !getenv("KH_DISABLE_XCD_GROUP")) {
    // partitions are cut for cap_u, the probes go to the (smaller) current table: 2^(PB - k) consecutive partitions share one of its chunks
    const uint32_t k_tab = log2u(t->cur.cap >> KH_LB);
    if (PB > k_tab) {
      const uint32_t G = std::min<uint32_t>(1u << (PB - k_tab), 16u);
      if (R.nparts % (8u * G) == 0) D.xcd_group = G;
    }
  }
  { Launch L(t, "k_dedup");
    if (S.rec12 == 2) { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH, true>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, D)); }
    else { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, D)); } }
  { Launch L(t, "k_scan");
    hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, t->stream, cnt_new, (uint64_t)R.nparts, noff); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, noff + R.nparts, 8, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipMemcpyAsync(t->hpin + 1, scal, 8, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipMemcpyAsync(t->hpin + 2, flags, sizeof(uint32_t) * KH_NFLAGS, hipMemcpyDeviceToHost, t->stream));
  t->hpin[30] = 0;
  if (t->part_overflow) HIPCHK(hipMemcpyAsync(t->hpin + 30, t->part_overflow, 4, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  if (t->part_overflow && (uint32_t)t->hpin[30]) return KH_RETRY_EXACT;
  const uint64_t dnew = t->hpin[0];
  const uint64_t last_first = mode == INS_PLUS ? n - 1 : (t->hpin[1] ? t->hpin[1] - 1 : 0);
  if (reinterpret_cast<const uint32_t*>(t->hpin + 2)[KH_FLAG_INTERNAL])
    return fail(t, KH_ERR_HIP, "internal: de-duplication set overflow");
  auto apply_plus = [&](int sign) {
    Launch L(t, "k_apply_plus");
    hipLaunchKernelGGL(k_apply_plus, dim3(std::min<uint32_t>(R.nparts, 4096u)), dim3(256), 0, t->stream, t->cur.s, S.merged_off, (const uint32_t*)D.cnt_upd,
                       (const uint64_t*)D.nk, (const uint32_t*)D.nv, R.nparts, sign);
  };
  if (plus_live) apply_plus(+1);
  const uint64_t new_cap = forced_cap ? forced_cap : capacity_after(t, t->cur.cap, t->lsize, n, dnew, last_first);
  if (dnew > 0 || new_cap != t->cur.cap) {
    // a chunk of the new table owns 2^(PB-k) consecutive partitions; read their lists in place when that is a
    // handful, gather them into one dense list when the capacity shrank far below the partitioning capacity
    const uint32_t k_new = new_cap > KH_L ? log2u(new_cap >> KH_LB) : 0u;
    const uint64_t* ck = nullptr; const uint32_t* cv = nullptr; const uint64_t* lo = nullptr; const uint32_t* lc = nullptr;
    if (dnew > 0) {
      if (PB - k_new <= 3) { ck = D.nk; cv = D.nv; lo = S.merged_off; lc = cnt_new; }
      else {
        uint64_t* gk; uint32_t* gv;
        TAKE(gk, uint64_t, dnew); TAKE(gv, uint32_t, dnew);
        Launch L(t, "k_gather_new");
        hipLaunchKernelGGL(k_gather_new, dim3(R.nparts), dim3(256), 0, t->stream, S.merged_off, noff, D.nk, D.nv, gk, gv);
        ck = gk; cv = gv; lo = noff; lc = nullptr;
      }
    }
    const bool pre_ok = fuse && new_cap == cap_u && !reinterpret_cast<const uint32_t*>(t->hpin + 2)[KH_FLAG_FUSE_INVALID];
    st = rebuild(t, new_cap, ck, cv, lo, lc, PB, false, t->lsize + dnew, pre_ok ? &pre : nullptr);
    if (st != KH_OK) {        // the table keeps its layout: it must keep its values too
      if (plus_live) { apply_plus(-1); hipStreamSynchronize(t->stream); }
      return st;
    }
    t->lsize += dnew;
  }
  if (mode == INS_UPDATE) {   // update(k,v): existing keys take the value of their LAST occurrence in the batch
    D.T = t->cur; D.table_empty = 0; D.mode = KH_DEDUP_LAST; D.count_cap = 0;
    Launch L(t, "k_dedup_assign");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, D));
    HIPCHK(hipGetLastError());
  }
  *n_new_out = dnew;
  return KH_OK;
}

// device-resident pairs -> table; the batch is cut only where the reference's one-doubling-per-call rule or the 32-bit
// record index demands it.  What the arena holds at entry stays (staged input); the rest is reused per sub-batch.
kh_status insert_device(kh_table* t, const char* kb, uint32_t kstride, const char* vb, uint32_t vstride, uint64_t n, int mode,
                        uint64_t* total_new_out) {
  const size_t keep_blk = t->blk, keep_off = t->off;
  uint64_t total_new = 0, done = 0;
  kh_status st = KH_OK;
  while (done < n && st == KH_OK) {
    // more than one doubling pending (only after set_max_load_factor / an LP shrink): the reference doubles
    // once per insert() call, so peel single calls until at most one is pending
    uint64_t take = n - done;
    if (t->lsize >= threshold(t->cur.cap << 1, t->max_lf)) take = 1;
    if (take > g_max_pass) take = g_max_pass;          // 32-bit record indices; bounded workspace (see g_max_pass)
    t->blk = keep_blk; t->off = keep_off;
    uint64_t nn = 0;
    if (take == 1 && t->lsize >= threshold(t->cur.cap << 1, t->max_lf)) {
      // single call: insert() runs rehash(2c) once (:530 / LP :439).  The LP rehash copies without looking at the load: exactly
      // one doubling.  The RH rehash re-inserts through insert() (:432-464), whose own doubling check fires again while the copy
      // is under way: the capacity ends where all lsize elements fit below max_load (the rule do_rehash applies)
      uint64_t c = t->cur.cap << 1;
      if (t->kind == KHK_RH) while (t->lsize > threshold(c, t->max_lf)) c <<= 1;
      st = insert_core(t, kb + done * kstride, kstride, vb ? vb + done * vstride : nullptr, vstride, 1, mode, c, &nn);
    } else {
      st = insert_core(t, kb + done * kstride, kstride, vb ? vb + done * vstride : nullptr, vstride, take, mode, 0, &nn);
    }
    total_new += nn;
    done += take;
  }
  *total_new_out = total_new;
  return st;
}

// A handful of keys applied in place by one lane (k_small_batch) instead of re-laying out the table.  *done = keys applied
// (all of them unless a Robin Hood displacement chain would pass distance 127: the caller continues on the general path).
#define KH_SMALL_N 16
const bool g_disable_small = getenv("KH_DISABLE_SMALL_BATCH") != nullptr;      // test hook
kh_status small_batch(kh_table* t, const char* kb, uint32_t kstride, const char* vb, uint32_t vstride, uint32_t vconst, uint32_t n, int op,
                      uint64_t* changed, uint32_t* done) {
  unsigned long long* out; uint32_t* flags;
  TAKE(out, unsigned long long, 2); TAKE(flags, uint32_t, KH_NFLAGS);
  HIPCHK(hipMemsetAsync(out, 0, 16, t->stream));
  HIPCHK(hipMemsetAsync(flags, 0, sizeof(uint32_t) * KH_NFLAGS, t->stream));
  { Launch L(t, "k_small_batch");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_small_batch<KIND, HASH>), dim3(1), dim3(64), 0, t->stream, t->cur, kb, kstride, vb, vstride,
                                                             vconst, n, op, t->seed, out, flags)); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, out, 16, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipMemcpyAsync(t->hpin + 2, flags, sizeof(uint32_t) * KH_NFLAGS, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  if (reinterpret_cast<const uint32_t*>(t->hpin + 2)[KH_FLAG_INTERNAL]) return fail(t, KH_ERR_FULL, "ERROR: did not find a slot to insert into (hashmap_linearprobe.hpp:503)");
  *changed = t->hpin[0];
  *done = (uint32_t)t->hpin[1];
  return KH_OK;
}

// ---- batches of middle size applied in place (Robin Hood; kh_kernels.h "Batches of middle size") --------------------------
// Taken when no call of the batch can trigger a doubling (size + n <= max_load), the batch is small against the table (at
// most ~1 key per region of 512 slots) and the load stays moderate (clusters much shorter than a region): cost O(batch)
// instead of the O(table) re-layout.
const bool g_disable_inplace = getenv("KH_DISABLE_INPLACE") != nullptr;      // test hook: force the re-layout
inline bool inplace_ok(const kh_table* t, uint64_t n) {
  return !g_disable_inplace && t->kind == KHK_RH && n > KH_SMALL_N && t->lsize > 0 && t->cur.cap >= 16 * (uint64_t)KH_IP_L &&
         n <= (t->cur.cap >> KH_IP_LB) && t->lsize + n <= threshold(t->cur.cap, 0.85f);
}
inline size_t ws_inplace(const kh_table* t, uint64_t n) { return n * 32 + (n <= 32768 ? n * 32 * 32 : 0) + (t->cur.cap >> KH_IP_LB) * (size_t)(KH_IP_CAP * 16 + 8) + (size_t(1) << 20); }
// result block of the in-place passes (device, zero at launch; copied to t->hpin in one piece)
struct IpResult { unsigned long long defer1, defer2, done, n_in; uint32_t flags[KH_NFLAGS]; };
// the three passes over the input list `src` describes (fields in_* / part_* / n of P0): regions, regions shifted by half, one lane
template <int OP>
kh_status inplace_passes(kh_table* t, const KhInplaceParams& src, uint64_t n_max, IpResult** res_out) {
  const uint32_t regions = (uint32_t)(t->cur.cap >> KH_IP_LB);
  // one zero-initialised block: result + the two counter arrays
  char* z; ulonglong2 *bins, *defer1, *defer2;
  const size_t zbytes = sizeof(IpResult) + sizeof(uint32_t) * 2 * (size_t)regions;
  TAKE(z, char, zbytes);
  TAKE(bins, ulonglong2, (size_t)regions * KH_IP_CAP);
  TAKE(defer1, ulonglong2, n_max); TAKE(defer2, ulonglong2, n_max);
  HIPCHK(hipMemsetAsync(z, 0, zbytes, t->stream));
  IpResult* res = reinterpret_cast<IpResult*>(z);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(z + sizeof(IpResult));
  KhInplaceParams P = src;
  P.T = t->cur; P.seed = t->seed; P.bins = bins; P.n_done = &res->done; P.n_in = &res->n_in; P.flags = res->flags;
  const uint32_t apply_grid = (regions + KH_IP_THREADS - 1) / KH_IP_THREADS;
  // pass 1: regions [r L, (r+1) L)
  P.ofs = 0; P.cnt = cnt; P.defer = defer1; P.n_defer = &res->defer1;
  { Launch L(t, "k_ip_bin");
    const uint32_t grid = P.part_off ? std::min<uint32_t>(P.nparts, 4096u) : grid_for(n_max, 256);
    KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_ip_bin<HASH>), dim3(grid), dim3(256), 0, t->stream, P)); }
  { Launch L(t, "k_ip_apply");
    KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_ip_apply<HASH, OP>), dim3(apply_grid), dim3(KH_IP_THREADS), 0, t->stream, P)); }
  // pass 2: what crossed a boundary, regions shifted by half a region
  P.ofs = KH_IP_L / 2; P.in_k = nullptr; P.in_v = nullptr; P.part_off = nullptr; P.in_rec = defer1; P.n = 0; P.n_dev = &res->defer1;
  P.cnt = cnt + regions; P.defer = defer2; P.n_defer = &res->defer2;
  { Launch L(t, "k_ip_bin2");
    KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_ip_bin<HASH>), dim3(grid_for(std::max<uint64_t>(n_max / 8, 256), 256)), dim3(256), 0, t->stream, P)); }
  { Launch L(t, "k_ip_apply2");
    KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_ip_apply<HASH, OP>), dim3(apply_grid), dim3(KH_IP_THREADS), 0, t->stream, P)); }
  // pass 3: the rest, one lane
  P.in_rec = defer2; P.n_dev = &res->defer2;
  { Launch L(t, "k_ip_serial");
    KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_ip_serial<HASH, OP>), dim3(1), dim3(64), 0, t->stream, P)); }
  HIPCHK(hipGetLastError());
  *res_out = res;
  return KH_OK;
}

kh_status insert_inplace(kh_table* t, const char* kb, uint32_t kstride, const char* vb, uint32_t vstride, uint64_t n, int mode, uint64_t* n_new_out) {
  *n_new_out = 0;
  // equal keys must meet: partition by a few hash bits into pieces of ~1024 records (one pass), fold duplicates in LDS and
  // test membership against the table (k_dedup: updates / sums of EXISTING keys happen there); what comes out are the
  // batch's distinct NEW keys, one list per partition, which the binning kernel reads where they lie
  uint32_t PB = 0;
  while (PB < 11 && (n >> PB) > 1024) ++PB;
  Partitioned R;
  kh_status st = KH_OK;
  uint64_t list_cap = n;                       // entries the per-partition output lists of k_dedup span
  if (n <= 32768) {
    // one launch: workgroup q keeps partition q in its own slot of n records (k_part_direct)
    const uint32_t nparts = 1u << PB;
    list_cap = (uint64_t)nparts * n;
    ulonglong2 *rec, *spare; unsigned long long* cur; uint64_t* starts;
    TAKE(rec, ulonglong2, list_cap); TAKE(spare, ulonglong2, list_cap); TAKE(cur, unsigned long long, nparts); TAKE(starts, uint64_t, (size_t)nparts + 1);
    { Launch L(t, "k_part_direct");
      KH_SWITCH_HASH(t->hash, hipLaunchKernelGGL((k_part_direct<HASH>), dim3(nparts), dim3(512), 0, t->stream, kb, kstride, vb, vstride,
                                                 mode == INS_PLUS ? 1u : 0u, n, t->seed, PB, rec, cur, starts)); }
    HIPCHK(hipGetLastError());
    R.rec = rec; R.part_off = starts; R.PB = PB; R.nparts = nparts; R.spare = spare; R.slot = n; R.cursor = cur; R.overflow = nullptr;
  } else {
    ulonglong2 *tmp, *fin;
    TAKE(tmp, ulonglong2, n); TAKE(fin, ulonglong2, n);
    st = partition_batch(t, kb, kstride, vb, vstride, mode == INS_PLUS ? 1u : 0u, n, 0, PB, tmp, fin, R);
    if (st != KH_OK) return st;
  }
  KhSrcSet S;
  memset(&S, 0, sizeof(S));
  S.rec[0] = R.rec; S.off[0] = R.part_off; S.n = 1; S.merged_off = R.part_off; S.slot[0] = R.slot; S.cur[0] = R.cursor;
  uint32_t* cnt_new;
  TAKE(cnt_new, uint32_t, R.nparts);
  KhInplaceParams src;
  memset(&src, 0, sizeof(src));
  src.in_k = reinterpret_cast<uint64_t*>(R.spare); src.in_v = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(R.spare) + list_cap * 8);
  src.part_off = R.part_off; src.part_cnt = cnt_new; src.nparts = R.nparts;
  KhDedupParams D;
  memset(&D, 0, sizeof(D));
  D.src = S;
  D.nk = const_cast<uint64_t*>(src.in_k); D.nv = const_cast<uint32_t*>(src.in_v);
  D.cnt_new = cnt_new; D.count_cap = 0; D.PB = PB;
  D.T = t->cur; D.seed = t->seed; D.table_empty = 0; D.mode = mode == INS_PLUS ? KH_DEDUP_PLUS : KH_DEDUP_FIRST;
  char* zpre;                   // k_dedup's scalar and flag words
  TAKE(zpre, char, 64);
  HIPCHK(hipMemsetAsync(zpre, 0, 64, t->stream));
  D.max_idx_plus1 = reinterpret_cast<unsigned long long*>(zpre); D.flags = reinterpret_cast<uint32_t*>(zpre + 32);
  if (mode == INS_PLUS) { TAKE(D.cnt_upd, uint32_t, R.nparts); D.plus_immediate = getenv("KH_DISABLE_PLUS_IMMEDIATE") ? 0 : 1; }      // (nothing discards an in-place batch)
  { Launch L(t, "k_dedup");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, D)); }
  if (mode == INS_PLUS) {      // sums of the keys the table already holds (listed by k_dedup; nothing can discard an in-place batch)
    Launch L(t, "k_apply_plus");
    hipLaunchKernelGGL(k_apply_plus, dim3(std::min<uint32_t>(R.nparts, 4096u)), dim3(256), 0, t->stream, t->cur.s, (const uint64_t*)R.part_off, (const uint32_t*)D.cnt_upd,
                       (const uint64_t*)D.nk, (const uint32_t*)D.nv, R.nparts, 1);
  }
  IpResult* res = nullptr;
  st = inplace_passes<KH_IP_INSERT>(t, src, n, &res);
  if (st != KH_OK) return st;
  if (mode == INS_UPDATE) {   // update(k,v): every key of the batch is in the table now; it takes the value of its LAST occurrence
    D.mode = KH_DEDUP_LAST;
    Launch L(t, "k_dedup_assign");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_dedup<KIND, HASH>), dim3(R.nparts), dim3(KH_CHUNK_THREADS), 0, t->stream, D));
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(t->hpin, res, sizeof(IpResult), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipMemcpyAsync(t->hpin + 32, zpre, 64, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  const IpResult* h = reinterpret_cast<const IpResult*>(t->hpin);
  const uint64_t dnew = h->n_in, placed = h->done;
  t->lsize += placed;
  *n_new_out = placed;
  if (reinterpret_cast<const uint32_t*>(t->hpin + 32 + 4)[KH_FLAG_INTERNAL]) return fail(t, KH_ERR_HIP, "internal: de-duplication set overflow");
  if (placed != dnew || h->flags[KH_FLAG_PROBE_OVERFLOW])
    return fail(t, KH_ERR_PROBE_OVERFLOW, "Robin Hood probe distance would exceed 127 (7-bit info field, hashmap_robinhood.hpp:142-144,556); "
                                          "the keys of the batch that fit were applied in place");
  return KH_OK;
}

kh_status do_insert(kh_table* t, const void* keys, uint32_t kstride, const void* vals, uint32_t vstride, uint64_t n,
                    kh_mem where, int mode, uint64_t* n_inserted, bool tail_reserve = true) {
  if (n_inserted) *n_inserted = 0;
  if (n && !keys) return fail(t, KH_ERR_INVALID, "null keys");
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  { const uint64_t np_ = std::min<uint64_t>(n, g_max_pass);       // (pairs of one internal pass: insert_device)
    const uint64_t cu = capacity_after(t, t->cur.cap, t->lsize, np_ ? np_ : 1, np_, np_ ? np_ - 1 : 0);
    const bool ip = inplace_ok(t, n) && t->lsize + n <= t->max_load;      // in place: no re-layout workspace, bins instead
    kh_status ps = ip ? arena_prepare(t, (where == KH_MEM_HOST ? n * 16 : 0) + n * 48 + ws_inplace(t, n))
                      // (68 B per pair: two buffers of 16-byte records in histogram-free slots of up to twice the mean fill -- a duplicate-heavy batch,
                      //  insert_core -- + the sample set; a duplicate-free batch takes 28 B per pair of it, an exact partition 58)
                      : arena_prepare(t, (where == KH_MEM_HOST ? n * 16 : 0) + np_ * 68 + ws_rebuild(cu) + cu * 2 + (cu > KH_L ? (cu >> KH_LB) : 1) * 64 + (np_ / KH_PART_TILE + 4096) * 16 + (size_t(1) << 20));
    if (ps != KH_OK) return ps; }
  const char* kb = static_cast<const char*>(keys);
  const char* vb = static_cast<const char*>(vals);
  if (where == KH_MEM_HOST && n) {
    char* d = nullptr;
    if (kstride == 16) {           // pair array: one copy, values live at +8
      TAKE(d, char, n * 16);
      HIPCHK(hipMemcpyAsync(d, keys, n * 16, hipMemcpyHostToDevice, t->stream));
      kb = d; vb = d + 8;
    } else {
      TAKE(d, char, n * 8);
      HIPCHK(hipMemcpyAsync(d, keys, n * 8, hipMemcpyHostToDevice, t->stream));
      kb = d;
      if (vals) {
        char* dv = nullptr;
        TAKE(dv, char, n * 4);
        HIPCHK(hipMemcpyAsync(dv, vals, n * 4, hipMemcpyHostToDevice, t->stream));
        vb = dv;
      }
    }
  }
  uint64_t total_new = 0;
  kh_status st = KH_OK;
  // a handful of keys into a table with room for all of them (no insert() call of the batch can trigger the doubling):
  // applied in place, one after the other, by the reference's own single-key algorithms
  if (n > 0 && n <= KH_SMALL_N && t->lsize > 0 && t->lsize + n <= t->max_load && !g_disable_small) {
    uint64_t ch = 0; uint32_t done = 0;
    st = small_batch(t, kb, kstride, vb, vstride, mode == INS_PLUS ? 1u : 0u, (uint32_t)n,
                     mode == INS_UPDATE ? KH_SMALL_UPDATE : (mode == INS_PLUS ? KH_SMALL_PLUS : KH_SMALL_FIRST), &ch, &done);
    if (st != KH_OK) return st;
    t->lsize += ch;
    total_new = ch;
    kb += (uint64_t)done * kstride;
    if (vb) vb += (uint64_t)done * vstride;
    n -= done;                      // > 0 only when a displacement chain would pass distance 127: the general path reports it
  }
  if (n > 0 && inplace_ok(t, n) && t->lsize + n <= t->max_load) {
    uint64_t more = 0;
    st = insert_inplace(t, kb, kstride, vb, vstride, n, mode, &more);
    total_new += more;
    n = 0;
  }
  if (n > 0) {
    uint64_t more = 0;
    st = insert_device(t, kb, kstride, vb, vstride, n, mode, &more);
    total_new += more;
  }
  // trailing reserve(lsize) of insert(Iter,Iter) (:672 / :546): a no-op unless size > max_load (after set_max_load_factor).
  // kh_update stands for a sequence of update(k,v) calls (:1274), which has no such tail
  if (st == KH_OK && mode != INS_UPDATE && tail_reserve) st = do_reserve(t, t->lsize);
  if (st == KH_OK) HIPCHK(hipStreamSynchronize(t->stream));
  arena_consolidate(t);
  if (n_inserted) *n_inserted = total_new;
  return st;
}

// generic compaction: flags/q/vals (device) -> out (device); returns the number of hits
kh_status compact(kh_table* t, const uint8_t* flags, const uint64_t* q, const uint32_t* vals, uint64_t n,
                  uint64_t* out_keys, uint32_t* out_vals, uint8_t* out_pairs, uint64_t* n_out) {
  *n_out = 0;
  if (n == 0) return KH_OK;
  const uint64_t ntl = (n + KH_CMP_TILE - 1) / KH_CMP_TILE;
  uint32_t* sums; uint64_t* offs;
  TAKE(sums, uint32_t, ntl); TAKE(offs, uint64_t, ntl + 1);
  { Launch L(t, "k_flag_tile_sums");
    hipLaunchKernelGGL(k_flag_tile_sums, dim3((uint32_t)ntl), dim3(256), 0, t->stream, flags, n, sums); }
  { Launch L(t, "k_scan");
    hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, t->stream, sums, ntl, offs); }
  if (out_keys || out_pairs) {
    Launch L(t, "k_compact_hits");
    hipLaunchKernelGGL(k_compact_hits, dim3((uint32_t)ntl), dim3(256), 0, t->stream, flags, q, vals, n, offs, out_keys, out_vals, out_pairs);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, offs + ntl, 8, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  *n_out = t->hpin[0];
  return KH_OK;
}

// launches k_find<KIND, HASH, OUT> over n device-resident queries; *hits_dev (device, 8 B) receives the number of hits
kh_status launch_find(kh_table* t, int out_mode, const uint64_t* q, uint64_t n, uint32_t* dvals, uint8_t* dfound, uint64_t* dkeys, uint8_t* dpairs,
                      unsigned long long** hits_dev) {
  const uint64_t ntiles = (n + KH_Q_TILE - 1) / KH_Q_TILE;
  // control block: [0] hit count, [1] ticket, then one look-back granule per tile (compacted forms)
  unsigned long long* ctl;
  const size_t nctl = 2 + ((out_mode == KH_FIND_COMPACT || out_mode == KH_FIND_PAIRS) ? ntiles : 0);
  TAKE(ctl, unsigned long long, nctl);
  HIPCHK(hipMemsetAsync(ctl, 0, nctl * 8, t->stream));
  KhFindParams F;
  memset(&F, 0, sizeof(F));
  F.T = t->cur; F.q = q; F.n = n; F.seed = t->seed;
  F.out_vals = dvals; F.out_found = dfound; F.out_keys = dkeys; F.out_pairs16 = dpairs;
  F.n_found = (out_mode == KH_FIND_COUNT && !hits_dev) ? nullptr : ctl;      // count(Iter,Iter) returns no total
  F.ticket = reinterpret_cast<uint32_t*>(ctl + 1); F.tile_state = ctl + 2;
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, t->device);
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, (uint64_t)ncu * 8));
  const char* name = out_mode == KH_FIND_COUNT ? "k_count" : "k_find";
  { Launch L(t, name);
#define KH_FIND_LAUNCH(OUT)                                                                                                            \
    if (t->seed.xk) { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_find<KIND, HASH, OUT, true>), dim3(grid), dim3(KH_Q_THREADS), 0, t->stream, F)); }   \
    else { KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_find<KIND, HASH, OUT, false>), dim3(grid), dim3(KH_Q_THREADS), 0, t->stream, F)); }
    switch (out_mode) {
      case KH_FIND_PERQUERY: KH_FIND_LAUNCH(KH_FIND_PERQUERY) break;
      case KH_FIND_COMPACT: KH_FIND_LAUNCH(KH_FIND_COMPACT) break;
      case KH_FIND_PAIRS: KH_FIND_LAUNCH(KH_FIND_PAIRS) break;
      default: KH_FIND_LAUNCH(KH_FIND_COUNT) break;
    }
#undef KH_FIND_LAUNCH
  }
  HIPCHK(hipGetLastError());
  if (hits_dev) *hits_dev = ctl;
  return KH_OK;
}

kh_status do_find(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint32_t* out_vals, uint8_t* out_found,
                  uint64_t* out_ckeys, uint32_t* out_cvals, void* out_pairs, bool compacted, uint64_t* n_found) {
  if (n_found) *n_found = 0;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  if (n == 0) return KH_OK;
  if (!keys) return fail(t, KH_ERR_INVALID, "null keys");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, n * 40 + (n / KH_Q_TILE + 8) * 8 + (size_t(1) << 20)); if (ps != KH_OK) return ps; }
  const uint64_t* q;
  kh_status st = stage_in<uint64_t>(t, keys, n, where, &q);
  if (st != KH_OK) return st;
  const bool host = where == KH_MEM_HOST;
  unsigned long long* hits_dev = nullptr;
  uint64_t hits = 0;
  if (!compacted) {
    uint32_t* dv = out_vals; uint8_t* df = out_found;
    if (host && dv) TAKE(dv, uint32_t, n);
    if (host || !df) TAKE(df, uint8_t, n);
    st = launch_find(t, KH_FIND_PERQUERY, q, n, dv, df, nullptr, nullptr, &hits_dev);
    if (st != KH_OK) return st;
    HIPCHK(hipMemcpyAsync(t->hpin, hits_dev, 8, hipMemcpyDeviceToHost, t->stream));
    if (host) {
      // values of misses stay untouched in the caller's buffer: copy through a flag-selective host loop
      std::vector<uint32_t> hv(out_vals ? n : 0); std::vector<uint8_t> hf(n);
      if (out_vals) HIPCHK(hipMemcpyAsync(hv.data(), dv, n * 4, hipMemcpyDeviceToHost, t->stream));
      HIPCHK(hipMemcpyAsync(hf.data(), df, n, hipMemcpyDeviceToHost, t->stream));
      HIPCHK(hipStreamSynchronize(t->stream));
      for (uint64_t i = 0; i < n; ++i) { if (out_found) out_found[i] = hf[i]; if (hf[i] && out_vals) out_vals[i] = hv[i]; }
    } else if (n_found) HIPCHK(hipStreamSynchronize(t->stream));
    hits = t->hpin[0];
  } else {
    uint64_t* ck = out_ckeys; uint32_t* cv = out_cvals; uint8_t* cp = static_cast<uint8_t*>(out_pairs);
    if (host) {
      if (out_pairs) TAKE(cp, uint8_t, n * 16);
      else { TAKE(ck, uint64_t, n); TAKE(cv, uint32_t, n); }
    }
    st = launch_find(t, out_pairs ? KH_FIND_PAIRS : KH_FIND_COMPACT, q, n, out_pairs ? nullptr : cv, nullptr, out_pairs ? nullptr : ck,
                     out_pairs ? cp : nullptr, &hits_dev);
    if (st != KH_OK) return st;
    HIPCHK(hipMemcpyAsync(t->hpin, hits_dev, 8, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    hits = t->hpin[0];
    if (host && hits) {
      if (out_pairs) HIPCHK(hipMemcpyAsync(out_pairs, cp, hits * 16, hipMemcpyDeviceToHost, t->stream));
      else {
        HIPCHK(hipMemcpyAsync(out_ckeys, ck, hits * 8, hipMemcpyDeviceToHost, t->stream));
        HIPCHK(hipMemcpyAsync(out_cvals, cv, hits * 4, hipMemcpyDeviceToHost, t->stream));
      }
      HIPCHK(hipStreamSynchronize(t->stream));
    }
  }
  if (n_found) *n_found = hits;
  arena_consolidate(t);
  return KH_OK;
}

// erase_no_resize over a batch; the caller applies the form-specific resize rule
kh_status erase_core(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint64_t* n_erased) {
  *n_erased = 0;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  if (n == 0) return KH_OK;
  if (!keys) return fail(t, KH_ERR_INVALID, "null keys");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = inplace_ok(t, n) ? arena_prepare(t, n * 8 + ws_inplace(t, n))
                                    : arena_prepare(t, n * 8 + 2 * 8 * ((slack_slot((double)n / (double)std::max<uint64_t>(1, t->cur.cap >> KH_LB)) * std::max<uint64_t>(1, t->cur.cap >> KH_LB)) + KH_PART_TILE + n) +
                                                       ws_rebuild(t->cur.cap) + (t->cur.cap > KH_L ? (t->cur.cap >> KH_LB) : 1) * 96 + (n / KH_PART_TILE + 4096) * 16 + (size_t(2) << 20));
    if (ps != KH_OK) return ps; }
  const uint64_t* q;
  kh_status st = stage_in<uint64_t>(t, keys, n, where, &q);
  if (st != KH_OK) return st;
  if (t->kind == KHK_RH && n <= KH_SMALL_N && t->lsize > 0 && !g_disable_small) {      // backward-shift deletes in place
    uint64_t ne = 0; uint32_t done = 0;
    st = small_batch(t, reinterpret_cast<const char*>(q), 8, nullptr, 0, 0, (uint32_t)n, KH_SMALL_ERASE, &ne, &done);
    if (st != KH_OK) return st;
    t->lsize -= ne;
    *n_erased = ne;
    return KH_OK;
  }
  if (inplace_ok(t, n)) {       // backward-shift deletes in place, one lane per region of the table
    KhInplaceParams src;
    memset(&src, 0, sizeof(src));
    src.in_k = q; src.n = n;
    IpResult* res = nullptr;
    st = inplace_passes<KH_IP_ERASE>(t, src, n, &res);
    if (st != KH_OK) return st;
    HIPCHK(hipMemcpyAsync(t->hpin, &res->done, 8, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    t->lsize -= t->hpin[0];
    *n_erased = t->hpin[0];
    return KH_OK;
  }
  // ---- Robin Hood batch erase as streaming passes: the erase keys are partitioned by chunk like an insert batch (8-byte
  // records), and ONE launch re-lays the table out chunk by chunk, dropping every element whose key is among its chunk's erase keys (LDS
  // fold) -- no probe at random into HBM.  Speculative like the other one-launch forms: a chunk whose elements + erase keys exceed the
  // staging area, a carry chain or a poll time-out send the batch down the mark + re-layout path below.
  const uint32_t PBe = t->cur.cap > KH_L ? log2u(t->cur.cap >> KH_LB) : 0u;
  if (t->kind == KHK_RH && !g_disable_fused_rebuild && !getenv("KH_DISABLE_STREAM_ERASE") && t->cur.cap >= 2 * (uint64_t)KH_L && PBe <= 22 &&
      t->lsize > 0 && t->lsize <= threshold(t->cur.cap, 0.9f) && n <= 0xFFFFFFF0ull) {
    const size_t keep_blk = t->blk, keep_off = t->off;
    // hashed erase keys fill the chunks evenly: histogram-free partition into fixed slots (two passes for more than 2^11 chunks); a
    // slot that overflows (keys given many times over) is caught by the flag and the mark path takes the batch
    const bool slack = PBe > 11 && !g_disable_slack;
    const uint64_t mrec = slack ? (slack_slot((double)n / (double)(uint64_t(1) << PBe)) << PBe) + KH_PART_TILE : n;
    char *a, *b;
    TAKE(a, char, mrec * sizeof(KhRec8)); TAKE(b, char, mrec * sizeof(KhRec8));
    Partitioned R;
    st = partition_batch(t, reinterpret_cast<const char*>(q), 8, nullptr, 0, 0u, n, 0, PBe, reinterpret_cast<ulonglong2*>(a), reinterpret_cast<ulonglong2*>(b), R, slack, 2, nullptr, slack);
    if (st != KH_OK) return st;
    t->part_overflow = R.overflow;
    KhSlots nw;
    st = fresh_slots(t, t->cur.cap, nw);
    if (st != KH_OK) return st;
    KhFusedParams F;
    memset(&F, 0, sizeof(F));
    F.src.rec[0] = R.rec; F.src.off[0] = R.part_off; F.src.n = 1; F.src.merged_off = R.part_off; F.src.slot[0] = R.slot; F.src.cur[0] = R.cursor; F.src.rec12 = 2;
    F.PB = PBe; F.mode = KH_DEDUP_ERASE;
    F.R.Old = t->cur; F.R.PB = PBe;
    FusedRun run;
    // (erase keys of a chunk fit one per lane -- a histogram-free partition's slot, or the mean of an exact one leaves room: the ordered-stream
    //  kernel, four workgroups per CU; else the staging form)
    const bool ordered = (R.slot != 0 ? R.slot <= KH_ES_MAXK : n / (uint64_t(1) << PBe) <= KH_ES_MAXK / 2) && !getenv("KH_DISABLE_ORDERED_ERASE");
    { kh_status fs = launch_fused(t, ordered ? 4 : 3, F, nw, PBe, "k_erase_fused", &run); t->part_overflow = nullptr; if (fs != KH_OK) return fs; }
    const uint32_t* ff = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t->hpin) + 64);
    bool bad = R.overflow && (uint32_t)t->hpin[30];
    for (int i = 0; i < KH_NFLAGS; ++i) bad = bad || ff[i] != 0;
    const uint64_t placed = t->hpin[0];
    if (!bad && placed <= t->lsize) {
      KhSlots old = t->cur;
      t->cur = nw;
      retire_slots(t, old);
      *n_erased = t->lsize - placed;
      t->lsize = placed;
      return KH_OK;
    }
    if (getenv("KH_DEBUG_FUSED"))
      fprintf(stderr, "[kmerhash_amd] fused erase rejected: placed %llu size %llu flags=%u %u %u %u %u\n", (unsigned long long)placed, (unsigned long long)t->lsize, ff[0], ff[1], ff[2], ff[3], ff[4]);
    retire_slots(t, nw);
    t->blk = keep_blk; t->off = keep_off;
  }
  unsigned long long* cnt;
  TAKE(cnt, unsigned long long, 1);
  HIPCHK(hipMemsetAsync(cnt, 0, 8, t->stream));
  { Launch L(t, "k_erase_mark");
    KH_SWITCH_KIND_HASH(t->kind, t->hash, hipLaunchKernelGGL((k_erase_mark<KIND, HASH>), dim3(grid_for(n, KH_Q_THREADS * KH_Q_ITEMS, 2048)), dim3(KH_Q_THREADS), 0, t->stream, t->cur, q, n, t->seed, cnt)); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(t->hpin, cnt, 8, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  const uint64_t ne = t->hpin[0];
  if (ne && t->kind == KHK_RH) {
    st = rebuild(t, t->cur.cap, nullptr, nullptr, nullptr, nullptr, 0, true, t->lsize - ne);
    if (st != KH_OK) {        // the table keeps its elements: take the marks back
      hipLaunchKernelGGL(k_clear_marks, dim3(grid_for(t->cur.cap, 256)), dim3(256), 0, t->stream, t->cur);
      hipStreamSynchronize(t->stream);
      return st;
    }
  }
  t->lsize -= ne;
  *n_erased = ne;
  return KH_OK;
}

bool valid(const kh_table* t) { return t != nullptr; }

}  // namespace

// ===================================================================================================
// C ABI
// ===================================================================================================
extern "C" {

const char* kh_version(void) { return KH_VERSION_STR; }

kh_status kh_release_cached_memory(int device) {
  if (device < 0 || device >= 16) return KH_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return KH_ERR_HIP;
  hipDeviceSynchronize();
  pool_trim(device);
  return KH_OK;
}

kh_status kh_create(kh_table** out, kh_kind kind, uint32_t key_bytes, uint32_t val_bytes, kh_hash hash, uint64_t seed,
                    uint64_t capacity, float min_lf, float max_lf, int device) {
  if (!out) return KH_ERR_INVALID;
  *out = nullptr;
  if (key_bytes != 8 || val_bytes != 4) return KH_ERR_UNSUPPORTED;
  if ((int)kind < 0 || (int)kind > 1 || (int)hash < 0 || (int)hash > 3) return KH_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KH_ERR_HIP;   // fail loudly: no CPU fallback exists
  if (device < 0 || device >= ndev) return KH_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return KH_ERR_HIP;
  kh_table* t = new kh_table();
  t->kind = (int)kind; t->hash = (int)hash; t->device = device; t->seed = KhSeed{seed, 0u}; t->stream = nullptr;
  t->min_lf = min_lf; t->max_lf = max_lf; t->lsize = 0;
  t->cur = kNoSlots; t->spare = t->cur;
  t->blk = 0; t->off = 0; t->hpin = nullptr; t->prof = false; t->part_overflow = nullptr; t->batch_nodup = false; t->batch_vf = 1.0;
  memset(&t->ins, 0, sizeof(t->ins));
  const uint64_t cap = next_pow2(capacity);
  if (alloc_slots(t, cap, t->cur) != KH_OK) { delete t; return KH_ERR_NOMEM; }
  t->hpin = pinned_get();
  if (!t->hpin) { free_slots(t, t->cur); delete t; return KH_ERR_NOMEM; }
  if (fill_empty(t, t->cur) != KH_OK || hipStreamSynchronize(t->stream) != hipSuccess) { free_slots(t, t->cur); pinned_put(t->hpin); delete t; return KH_ERR_HIP; }
  t->min_load = threshold(cap, min_lf);
  t->max_load = threshold(cap, max_lf);
  *out = t;
  return KH_OK;
}

kh_status kh_destroy(kh_table* t) {
  if (!t) return KH_OK;
  hipSetDevice(t->device);
  hipStreamSynchronize(t->stream);
  for (auto& r : t->recs) { event_put(t->device, r.a); event_put(t->device, r.b); }
  free_slots(t, t->cur); free_slots(t, t->spare);
  for (auto& b : t->blocks) pool_free(t->device, b.p);
  pinned_put(t->hpin);
  delete t;
  return KH_OK;
}

kh_status kh_set_stream(kh_table* t, void* s) {
  if (!valid(t)) return KH_ERR_INVALID;
  if (t->stream == static_cast<hipStream_t>(s)) return KH_OK;     // unchanged: nothing to order
  hipSetDevice(t->device);
  hipStreamSynchronize(t->stream);                                   // work already issued stays ordered before the new stream's
  t->stream = static_cast<hipStream_t>(s);
  return KH_OK;
}
const char* kh_last_error(const kh_table* t) { return t ? t->err.c_str() : "null table"; }

kh_status kh_size(const kh_table* t, uint64_t* out) { if (!t || !out) return KH_ERR_INVALID; *out = t->lsize; return KH_OK; }
kh_status kh_capacity(const kh_table* t, uint64_t* out) { if (!t || !out) return KH_ERR_INVALID; *out = t->cur.cap; return KH_OK; }
kh_status kh_get_load_thresholds(const kh_table* t, uint64_t* mn, uint64_t* mx) {
  if (!t) return KH_ERR_INVALID;
  if (mn) *mn = t->min_load;
  if (mx) *mx = t->max_load;
  return KH_OK;
}
kh_status kh_set_min_load_factor(kh_table* t, float f) { if (!t) return KH_ERR_INVALID; t->min_lf = f; t->min_load = threshold(t->cur.cap, f); return KH_OK; }
kh_status kh_set_max_load_factor(kh_table* t, float f) { if (!t) return KH_ERR_INVALID; t->max_lf = f; t->max_load = threshold(t->cur.cap, f); return KH_OK; }
kh_status kh_get_load_factors(const kh_table* t, float* mn, float* mx, float* cur) {
  if (!t) return KH_ERR_INVALID;
  if (mn) *mn = t->min_lf;
  if (mx) *mx = t->max_lf;
  if (cur) *cur = static_cast<float>(t->lsize) / static_cast<float>(t->cur.cap);
  return KH_OK;
}
kh_status kh_clear(kh_table* t) {
  if (!t) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  t->lsize = 0;
  { kh_status fs = fill_empty(t, t->cur); if (fs != KH_OK) return fs; }
  HIPCHK(hipStreamSynchronize(t->stream));
  return KH_OK;
}
kh_status kh_reserve(kh_table* t, uint64_t n) { if (!t) return KH_ERR_INVALID; HIPCHK(hipSetDevice(t->device)); return do_reserve(t, n); }
kh_status kh_rehash(kh_table* t, uint64_t b) { if (!t) return KH_ERR_INVALID; HIPCHK(hipSetDevice(t->device)); return do_rehash(t, b); }

kh_status kh_insert(kh_table* t, const void* keys, const void* vals, uint64_t n, kh_mem where, uint64_t* n_inserted) {
  if (!t) return KH_ERR_INVALID;
  return do_insert(t, keys, 8, vals, 4, n, where, INS_FIRST, n_inserted);
}
kh_status kh_insert_pairs(kh_table* t, const void* pairs16, uint64_t n, kh_mem where, uint64_t* n_inserted) {
  if (!t) return KH_ERR_INVALID;
  return do_insert(t, pairs16, 16, pairs16 ? static_cast<const char*>(pairs16) + 8 : nullptr, 16, n, where, INS_FIRST, n_inserted);
}
kh_status kh_insert_one(kh_table* t, uint64_t key, uint32_t val, uint64_t* n_inserted) {
  if (!t) return KH_ERR_INVALID;
  return do_insert(t, &key, 8, &val, 4, 1, KH_MEM_HOST, INS_FIRST, n_inserted, false);
}
kh_status kh_update(kh_table* t, const void* keys, const void* vals, uint64_t n, kh_mem where, uint64_t* n_inserted) {
  if (!t) return KH_ERR_INVALID;
  return do_insert(t, keys, 8, vals, 4, n, where, INS_UPDATE, n_inserted);
}
// ---- streamed insert: ONE insert(Iter,Iter) whose input arrives in pieces (the multi-GPU exchange feeds the pieces as
//      they land; every feed is radix-partitioned right away, asynchronously, so that work overlaps the next transfer)
kh_status kh_insert_begin(kh_table* t, uint64_t n_total, int reduce_plus) {
  return kh_insert_begin_ex(t, n_total, reduce_plus ? KH_INS_REDUCE_PLUS : 0u);
}
kh_status kh_insert_begin_ex(kh_table* t, uint64_t n_total, unsigned flags) {
  if (!t) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is already in progress");
  if (flags & ~(unsigned)(KH_INS_REDUCE_PLUS | KH_INS_REPEATABLE)) return fail(t, KH_ERR_INVALID, "unknown flag");
  HIPCHK(hipSetDevice(t->device));
  memset(&t->ins, 0, sizeof(t->ins));
  t->ins.mode = (flags & KH_INS_REDUCE_PLUS) ? INS_PLUS : INS_FIRST;
  t->ins.repeatable = (flags & KH_INS_REPEATABLE) != 0 && !g_disable_slack;
  t->ins.n_total = n_total;
  const uint64_t cu = capacity_after(t, t->cur.cap, t->lsize, n_total ? n_total : 1, n_total, n_total ? n_total - 1 : 0);
  { kh_status ps = arena_prepare(t, n_total * ((flags & KH_INS_REPEATABLE) ? 72 : 56) + ws_rebuild(cu) + cu * 2 + (cu > KH_L ? (cu >> KH_LB) : 1) * 64 * (KH_MAX_SRC + 1) +
                                    (n_total / KH_PART_TILE + 4096 * KH_MAX_SRC) * 16 + (size_t(8) << 20));
    if (ps != KH_OK) return ps; }
  const uint32_t PB = cu > KH_L ? log2u(cu >> KH_LB) : 0u;
  // the one-shot rule evaluation needs at most one pending doubling and 32-bit stream positions; otherwise the pieces are
  // only collected and inserted by the general entry point at the end
  t->ins.fallback = n_total == 0 || n_total > 0xFFFFFFF0ull || PB > 22 || t->lsize >= threshold(t->cur.cap << 1, t->max_lf);
  t->ins.cap_u = cu; t->ins.PB = PB;
  if (n_total) {
    if (t->ins.fallback) { TAKE(t->ins.stage_k, uint64_t, n_total); TAKE(t->ins.stage_v, uint32_t, n_total); }
    else if (!t->ins.repeatable) { TAKE(t->ins.tmp, ulonglong2, n_total); TAKE(t->ins.fin, ulonglong2, n_total); }
    // (repeatable: the first feed's sample decides the layout, the buffers are taken there)
  }
  t->ins.active = true;
  return KH_OK;
}

kh_status kh_insert_feed(kh_table* t, const void* keys, const void* vals, uint64_t n, kh_mem where) {
  if (!t) return KH_ERR_INVALID;
  if (!t->ins.active) return fail(t, KH_ERR_INVALID, "kh_insert_feed without kh_insert_begin");
  if (n == 0) return KH_OK;
  if (!keys) return fail(t, KH_ERR_INVALID, "null keys");
  if (t->ins.fed + n > t->ins.n_total) return fail(t, KH_ERR_INVALID, "more pairs fed than announced to kh_insert_begin");
  HIPCHK(hipSetDevice(t->device));
  const uint64_t* dk; const uint32_t* dv;
  kh_status st = stage_in<uint64_t>(t, keys, n, where, &dk);
  if (st == KH_OK) st = stage_in<uint32_t>(t, vals, n, where, &dv);
  if (st != KH_OK) return st;
  if (t->ins.fallback) {
    HIPCHK(hipMemcpyAsync(t->ins.stage_k + t->ins.fed, dk, n * 8, hipMemcpyDeviceToDevice, t->stream));
    if (dv) HIPCHK(hipMemcpyAsync(t->ins.stage_v + t->ins.fed, dv, n * 4, hipMemcpyDeviceToDevice, t->stream));
    else HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(t->ins.stage_v + t->ins.fed), t->ins.mode == INS_PLUS ? 1 : 0, n, t->stream));
  } else {
    if (t->ins.S.n == KH_MAX_SRC) return fail(t, KH_ERR_UNSUPPORTED, "more than 16 feeds in one streamed insert");
    if (t->ins.S.n == 0 && n >= KH_SAMPLE_N) {
      // the first piece speaks for the batch: no duplicate among 65536 of its keys -> the one-launch build runs without its
      // LDS fold and CHECKS instead (a duplicate met after all sends the batch down the general path, nothing is lost).  One
      // host synchronisation per streamed insert, before anything else is in flight on this stream.
      unsigned long long* sset; uint32_t* dups;
      TAKE(sset, unsigned long long, KH_SAMPLE_SET); TAKE(dups, uint32_t, 1);
      HIPCHK(hipMemsetAsync(sset, 0, sizeof(unsigned long long) * KH_SAMPLE_SET, t->stream));
      HIPCHK(hipMemsetAsync(dups, 0, 4, t->stream));
      { Launch L(t, "k_sample_dups");
        hipLaunchKernelGGL(k_sample_dups, dim3(KH_SAMPLE_N / 256), dim3(256), 0, t->stream, reinterpret_cast<const char*>(dk), 8u, n, sset, dups); }
      HIPCHK(hipMemcpyAsync(t->hpin + 31, dups, 4, hipMemcpyDeviceToHost, t->stream));
      HIPCHK(hipStreamSynchronize(t->stream));
      t->ins.nodup = (uint32_t)t->hpin[31] == 0u;
      // a repeatable insert of (nearly) duplicate-free pieces: histogram-free partition into slots all pieces share
      t->ins.slack = t->ins.repeatable && (uint32_t)t->hpin[31] == 0u && part_buffer_records(t->ins.n_total, t->ins.PB, true) != t->ins.n_total;
    }
    if (t->ins.S.n == 0 && t->ins.repeatable) {       // buffers of the chosen layout
      const uint64_t nt = t->ins.n_total;
      if (t->ins.slack) {
        const uint32_t nparts = 1u << t->ins.PB;
        t->batch_nodup = t->ins.nodup;
        t->ins.rec12 = nodup_build_applies(t, t->ins.cap_u, t->ins.PB, t->ins.mode);
        t->batch_nodup = false;
        SlackShared& sh = t->ins.sh;
        sh.slot = slack_slot((double)nt / (double)nparts);
        char* f; TAKE(f, char, (sh.slot * nparts + KH_PART_TILE) * (t->ins.rec12 ? sizeof(KhRec12) : sizeof(ulonglong2)));
        t->ins.fin = reinterpret_cast<ulonglong2*>(f);
        TAKE(sh.cur2, unsigned long long, nparts); TAKE(sh.starts, uint64_t, (size_t)nparts + 1); TAKE(sh.ovf, uint32_t, 1);
        HIPCHK(hipMemsetAsync(sh.ovf, 0, 4, t->stream));
        hipLaunchKernelGGL(k_init_cursors, dim3((nparts + 256) / 256), dim3(256), 0, t->stream, sh.cur2, sh.starts, (uint64_t)nparts, sh.slot);
      } else { TAKE(t->ins.tmp, ulonglong2, nt); TAKE(t->ins.fin, ulonglong2, nt); }
    }
    Partitioned R;
    if (t->ins.slack) {
      // this piece's level-1 buckets (its own slots), then its records into the shared final slots
      const uint32_t B1 = (t->ins.PB + 1) / 2, nb1 = 1u << B1;
      const uint64_t slot1 = slack_slot((double)n / (double)nb1);
      char* tm; TAKE(tm, char, (slot1 * nb1 + KH_PART_TILE) * (t->ins.rec12 ? sizeof(KhRec12) : sizeof(ulonglong2)));
      st = partition_batch(t, reinterpret_cast<const char*>(dk), 8, reinterpret_cast<const char*>(dv), 4, t->ins.mode == INS_PLUS ? 1u : 0u,
                           n, t->ins.fed, t->ins.PB, reinterpret_cast<ulonglong2*>(tm), t->ins.fin, R, true, t->ins.rec12, &t->ins.sh);
      if (st != KH_OK) return st;
      t->ins.S.n = 1;      // one source whatever the number of pieces
    } else {
      st = partition_batch(t, reinterpret_cast<const char*>(dk), 8, reinterpret_cast<const char*>(dv), 4, t->ins.mode == INS_PLUS ? 1u : 0u,
                           n, t->ins.fed, t->ins.PB, t->ins.tmp, t->ins.fin + t->ins.fed, R);
      if (st != KH_OK) return st;
      t->ins.S.rec[t->ins.S.n] = R.rec; t->ins.S.off[t->ins.S.n] = R.part_off; ++t->ins.S.n;
    }
  }
  t->ins.fed += n;
  if (where == KH_MEM_HOST) HIPCHK(hipStreamSynchronize(t->stream));   // host buffers may be reused as soon as the feed returns
  return KH_OK;
}

kh_status kh_insert_end(kh_table* t, uint64_t* n_inserted) {
  if (n_inserted) *n_inserted = 0;
  if (!t) return KH_ERR_INVALID;
  if (!t->ins.active) return fail(t, KH_ERR_INVALID, "kh_insert_end without kh_insert_begin");
  t->ins.active = false;
  if (t->ins.fed != t->ins.n_total) return fail(t, KH_ERR_INVALID, "fewer pairs fed than announced to kh_insert_begin");
  HIPCHK(hipSetDevice(t->device));
  uint64_t nn = 0;
  kh_status st = KH_OK;
  const uint64_t n = t->ins.n_total;
  if (n) {
    if (t->ins.fallback) {
      st = insert_device(t, reinterpret_cast<const char*>(t->ins.stage_k), 8, reinterpret_cast<const char*>(t->ins.stage_v), 4, n, t->ins.mode, &nn);
    } else if (t->ins.slack) {
      const uint32_t nparts = 1u << t->ins.PB;
      const uint64_t list_cap = t->ins.sh.slot * nparts + KH_PART_TILE;
      KhSrcSet S;
      memset(&S, 0, sizeof(S));
      S.rec[0] = t->ins.fin; S.off[0] = t->ins.sh.starts; S.n = 1; S.merged_off = t->ins.sh.starts;
      S.slot[0] = t->ins.sh.slot; S.cur[0] = t->ins.sh.cur2; S.rec12 = t->ins.rec12 ? 1u : 0u;
      char* spare; TAKE(spare, char, list_cap * 16);
      t->part_overflow = t->ins.sh.ovf; t->batch_nodup = t->ins.nodup;
      st = insert_finish(t, S, n, t->ins.PB, t->ins.cap_u, t->ins.mode, 0, reinterpret_cast<ulonglong2*>(spare), &nn, list_cap);
      t->part_overflow = nullptr; t->batch_nodup = false;
      if (st == KH_RETRY_EXACT) {      // a slot overflowed / a duplicate met without stream positions: nothing was inserted
        if (n_inserted) *n_inserted = 0;
        HIPCHK(hipStreamSynchronize(t->stream));
        t->err = "the speculative partition of this repeatable streamed insert did not hold: feed the same pieces again (kh_insert_begin without KH_INS_REPEATABLE)";
        return KH_ERR_RETRY;
      }
    } else {
      KhSrcSet S = t->ins.S;
      if (S.n == 1) S.merged_off = S.off[0];
      else {
        uint64_t* mo;
        const uint64_t nq1 = (uint64_t(1) << t->ins.PB) + 1;
        TAKE(mo, uint64_t, nq1);
        hipLaunchKernelGGL(k_merge_offsets, dim3((uint32_t)((nq1 + 255) / 256)), dim3(256), 0, t->stream, S, nq1, mo);
        S.merged_off = mo;
      }
      t->batch_nodup = t->ins.nodup;
      st = insert_finish(t, S, n, t->ins.PB, t->ins.cap_u, t->ins.mode, 0, t->ins.tmp, &nn);
      t->batch_nodup = false;
    }
  }
  if (st == KH_OK) st = do_reserve(t, t->lsize);
  if (st == KH_OK) HIPCHK(hipStreamSynchronize(t->stream));
  if (n_inserted) *n_inserted = nn;
  return st;
}

kh_status kh_insert_abort(kh_table* t) {
  if (!t) return KH_ERR_INVALID;
  if (!t->ins.active) return KH_OK;
  // the feeds only wrote workspace (partition records); the table itself has not been touched.  Kernels queued by the feeds may
  // still read the caller's device buffers: wait for them, like kh_insert_end would have
  hipSetDevice(t->device);
  hipError_t e = hipStreamSynchronize(t->stream);
  t->ins.active = false;
  t->part_overflow = nullptr; t->batch_nodup = false;
  if (e != hipSuccess) return fail(t, KH_ERR_HIP, std::string("kh_insert_abort: ") + hipGetErrorString(e));
  return KH_OK;
}

kh_status kh_insert_reduce_plus(kh_table* t, const void* keys, const void* vals, uint64_t n, kh_mem where, uint64_t* n_inserted) {
  if (!t) return KH_ERR_INVALID;
  return do_insert(t, keys, 8, vals, 4, n, where, INS_PLUS, n_inserted);
}

kh_status kh_count(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint8_t* out01) {
  if (!t) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  if (n == 0) return KH_OK;
  if (!keys || !out01) return fail(t, KH_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, n * 9 + (size_t(1) << 20)); if (ps != KH_OK) return ps; }
  const uint64_t* q;
  kh_status st = stage_in<uint64_t>(t, keys, n, where, &q);
  if (st != KH_OK) return st;
  uint8_t* d = out01;
  if (where == KH_MEM_HOST) TAKE(d, uint8_t, n);
  st = launch_find(t, KH_FIND_COUNT, q, n, nullptr, d, nullptr, nullptr, nullptr);
  if (st != KH_OK) return st;
  if (where == KH_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(out01, d, n, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
  }
  arena_consolidate(t);
  return KH_OK;
}

kh_status kh_find(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint32_t* out_vals, uint8_t* out_found, uint64_t* n_found) {
  if (!t) return KH_ERR_INVALID;
  return do_find(t, keys, n, where, out_vals, out_found, nullptr, nullptr, nullptr, false, n_found);
}
kh_status kh_find_compact(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint64_t* out_keys, uint32_t* out_vals, uint64_t* n_found) {
  if (!t) return KH_ERR_INVALID;
  if (n && (!out_keys || !out_vals)) return fail(t, KH_ERR_INVALID, "null output");
  return do_find(t, keys, n, where, nullptr, nullptr, out_keys, out_vals, nullptr, true, n_found);
}
kh_status kh_find_compact_pairs(kh_table* t, const void* keys, uint64_t n, kh_mem where, void* out_pairs16, uint64_t* n_found) {
  if (!t) return KH_ERR_INVALID;
  if (n && !out_pairs16) return fail(t, KH_ERR_INVALID, "null output");
  return do_find(t, keys, n, where, nullptr, nullptr, nullptr, nullptr, out_pairs16, true, n_found);
}

kh_status kh_erase(kh_table* t, const void* keys, uint64_t n, kh_mem where, uint64_t* n_erased) {
  if (!t) return KH_ERR_INVALID;
  uint64_t ne = 0;
  kh_status st = erase_core(t, keys, n, where, &ne);
  if (n_erased) *n_erased = ne;
  if (st != KH_OK) { arena_consolidate(t); return st; }
  if (t->lsize < t->min_load) {
    if (t->kind == KHK_RH) st = do_reserve(t, t->lsize);   // hashmap_robinhood.hpp:1437: reserve() only grows
    else st = do_rehash(t, static_cast<uint64_t>(static_cast<float>(t->lsize) / t->max_lf));   // hashmap_linearprobe.hpp:1048
  }
  arena_consolidate(t);
  return st;
}
kh_status kh_erase_one(kh_table* t, uint64_t key, uint64_t* n_erased) {
  if (!t) return KH_ERR_INVALID;
  uint64_t ne = 0;
  kh_status st = erase_core(t, &key, 1, KH_MEM_HOST, &ne);
  if (n_erased) *n_erased = ne;
  if (st == KH_OK && t->lsize < t->min_load) st = do_rehash(t, t->cur.cap >> 1);   // :1425 / :1036
  arena_consolidate(t);
  return st;
}

}  // extern "C"
namespace {
// SoA view of the current table in the workspace (any of keys / vals / info / flags may be null)
kh_status unpack(kh_table* t, uint64_t* k, uint32_t* v, uint8_t* info, uint8_t* flags) {
  const uint64_t cap = t->cur.cap;
  Launch L(t, "k_unpack_slots");
  if (t->kind == KHK_RH) hipLaunchKernelGGL((k_unpack_slots<KHK_RH>), dim3(grid_for(cap, 256)), dim3(256), 0, t->stream, t->cur.s, cap, k, v, info, flags);
  else hipLaunchKernelGGL((k_unpack_slots<KHK_LP>), dim3(grid_for(cap, 256)), dim3(256), 0, t->stream, t->cur.s, cap, k, v, info, flags);
  HIPCHK(hipGetLastError());
  return KH_OK;
}
}  // namespace
extern "C" {
kh_status kh_to_vector(kh_table* t, uint64_t* keys_host, uint32_t* vals_host, uint64_t* n_out) {
  if (!t) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, t->cur.cap * 26 + (size_t(1) << 20)); if (ps != KH_OK) return ps; }
  const uint64_t cap = t->cur.cap;
  uint8_t* flags; uint64_t *sk, *ck; uint32_t *sv, *cv;
  TAKE(flags, uint8_t, cap); TAKE(sk, uint64_t, cap); TAKE(sv, uint32_t, cap); TAKE(ck, uint64_t, cap); TAKE(cv, uint32_t, cap);
  kh_status st = unpack(t, sk, sv, nullptr, flags);
  if (st != KH_OK) return st;
  uint64_t m = 0;
  st = compact(t, flags, sk, sv, cap, ck, cv, nullptr, &m);
  if (st != KH_OK) return st;
  if (m && keys_host) HIPCHK(hipMemcpyAsync(keys_host, ck, m * 8, hipMemcpyDeviceToHost, t->stream));
  if (m && vals_host) HIPCHK(hipMemcpyAsync(vals_host, cv, m * 4, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  if (n_out) *n_out = m;
  arena_consolidate(t);
  return KH_OK;
}
kh_status kh_export_info(kh_table* t, uint8_t* out_host) {
  if (!t || !out_host) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, t->cur.cap + (size_t(1) << 20)); if (ps != KH_OK) return ps; }
  uint8_t* info;
  TAKE(info, uint8_t, t->cur.cap);
  kh_status st = unpack(t, nullptr, nullptr, info, nullptr);
  if (st != KH_OK) return st;
  HIPCHK(hipMemcpyAsync(out_host, info, t->cur.cap, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return KH_OK;
}
kh_status kh_export_slots(kh_table* t, uint64_t* keys_host, uint32_t* vals_host) {
  if (!t) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, t->cur.cap * 12 + (size_t(1) << 20)); if (ps != KH_OK) return ps; }
  uint64_t* sk; uint32_t* sv;
  TAKE(sk, uint64_t, t->cur.cap); TAKE(sv, uint32_t, t->cur.cap);
  kh_status st = unpack(t, sk, sv, nullptr, nullptr);
  if (st != KH_OK) return st;
  if (keys_host) HIPCHK(hipMemcpyAsync(keys_host, sk, t->cur.cap * 8, hipMemcpyDeviceToHost, t->stream));
  if (vals_host) HIPCHK(hipMemcpyAsync(vals_host, sv, t->cur.cap * 4, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return KH_OK;
}
kh_status kh_export_raw_slots(kh_table* t, void* out_host) {
  if (!t || !out_host) return KH_ERR_INVALID;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipMemcpyAsync(out_host, t->cur.s, t->cur.cap * sizeof(KhSlot), hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return KH_OK;
}
kh_status kh_displacement_histogram(kh_table* t, uint64_t out[128]) {
  if (!t || !out) return KH_ERR_INVALID;
  for (int i = 0; i < 128; ++i) out[i] = 0;
  if (t->kind != KHK_RH) return KH_OK;
  if (t->ins.active) return fail(t, KH_ERR_INVALID, "a streamed insert is in progress (kh_insert_end first)");
  HIPCHK(hipSetDevice(t->device));
  { kh_status ps = arena_prepare(t, size_t(1) << 20); if (ps != KH_OK) return ps; }
  unsigned long long* d;
  TAKE(d, unsigned long long, 128);
  HIPCHK(hipMemsetAsync(d, 0, 128 * 8, t->stream));
  { Launch L(t, "k_disp_hist");
    hipLaunchKernelGGL(k_disp_hist, dim3(grid_for(t->cur.cap, 256, 1024)), dim3(256), 0, t->stream, t->cur.s, t->cur.cap, d); }
  HIPCHK(hipMemcpyAsync(out, d, 128 * 8, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  return KH_OK;
}

}  // extern "C"
namespace {
inline bool xform_ok(kh_key_transform xf, uint32_t k) { return xf == KH_XF_IDENTITY || (xf == KH_XF_DNA_LEX_LESS && k >= 1 && k <= 32); }
inline KhSeed make_seed(uint64_t seed, kh_key_transform xf, uint32_t k) { return KhSeed{seed, xf == KH_XF_DNA_LEX_LESS ? k : 0u}; }
}
extern "C" {
kh_status kh_set_key_transform(kh_table* t, kh_key_transform xf, uint32_t k) {
  if (!t) return KH_ERR_INVALID;
  if (!xform_ok(xf, k)) return fail(t, KH_ERR_INVALID, "unknown key transform / k outside 1..32");
  if (t->lsize != 0 || t->ins.active) return fail(t, KH_ERR_INVALID, "the key transform can only be set on an empty table");
  t->seed.xk = xf == KH_XF_DNA_LEX_LESS ? k : 0u;
  return KH_OK;
}
kh_status kh_get_key_transform(const kh_table* t, kh_key_transform* xf, uint32_t* k) {
  if (!t) return KH_ERR_INVALID;
  if (xf) *xf = t->seed.xk ? KH_XF_DNA_LEX_LESS : KH_XF_IDENTITY;
  if (k) *k = t->seed.xk;
  return KH_OK;
}
kh_status kh_hash_batch(kh_hash hash, uint64_t seed, const void* keys, uint64_t n, kh_mem where, uint64_t* out, int device, void* stream_) {
  return kh_hash_batch_transformed(hash, seed, KH_XF_IDENTITY, 0, keys, n, where, out, device, stream_);
}
kh_status kh_hash_batch_transformed(kh_hash hash, uint64_t seed_, kh_key_transform xf, uint32_t k, const void* keys, uint64_t n, kh_mem where,
                                    uint64_t* out, int device, void* stream_) {
  kh_table* t = nullptr;
  if (n == 0) return KH_OK;
  if (!keys || !out || (int)hash < 0 || (int)hash > 3 || !xform_ok(xf, k)) return KH_ERR_INVALID;
  const KhSeed seed = make_seed(seed_, xf, k);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(device));
  const uint64_t* dk = static_cast<const uint64_t*>(keys);
  uint64_t* dout = out; uint64_t* tmp = nullptr;
  if (where == KH_MEM_HOST) {
    HIPCHK(pool_alloc(device, n * 16, reinterpret_cast<void**>(&tmp)));
    HIPCHK(hipMemcpyAsync(tmp, keys, n * 8, hipMemcpyHostToDevice, stream));
    dk = tmp; dout = tmp + n;
  }
  KH_SWITCH_HASH((int)hash, hipLaunchKernelGGL((k_hash_batch<HASH>), dim3(grid_for(n, 256)), dim3(256), 0, stream, dk, n, seed, dout));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && where == KH_MEM_HOST) e = hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess && where == KH_MEM_HOST) e = hipStreamSynchronize(stream);
  if (tmp) pool_free(device, tmp);
  return e == hipSuccess ? KH_OK : KH_ERR_HIP;
}

kh_status kh_shard_permute(kh_hash hash, uint64_t seed, uint32_t p, const uint64_t* keys, const uint32_t* vals, uint64_t n,
                           uint64_t* out_keys, uint32_t* out_vals, uint64_t* counts_host, int device, void* stream_) {
  return kh_shard_permute_transformed(hash, seed, KH_XF_IDENTITY, 0, p, keys, vals, n, out_keys, out_vals, counts_host, device, stream_);
}
kh_status kh_shard_permute_transformed(kh_hash hash, uint64_t seed_, kh_key_transform xf, uint32_t k, uint32_t p, const uint64_t* keys,
                                       const uint32_t* vals, uint64_t n, uint64_t* out_keys, uint32_t* out_vals, uint64_t* counts_host,
                                       int device, void* stream_) {
  kh_table* t = nullptr;
  if (p == 0 || p > KH_SHARD_MAXR || !counts_host || (int)hash < 0 || (int)hash > 3 || !xform_ok(xf, k)) return KH_ERR_INVALID;
  const KhSeed seed = make_seed(seed_, xf, k);
  for (uint32_t r = 0; r < p; ++r) counts_host[r] = 0;
  if (n == 0) return KH_OK;
  if (!keys || (out_keys && vals && !out_vals)) return KH_ERR_INVALID;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(device));
  const uint32_t ntiles = (uint32_t)((n + KH_SHARD_TILE - 1) / KH_SHARD_TILE);
  const uint32_t pmask = (p & (p - 1)) == 0 ? p - 1 : 0;   // power of two: & (p-1); else % p.  (p == 1: mask 0 -> % 1)
  uint32_t* tc = nullptr; uint64_t* toff = nullptr;
  const uint64_t m = (uint64_t)p * ntiles;
  HIPCHK(pool_alloc(device, m * 4, reinterpret_cast<void**>(&tc)));
  if (pool_alloc(device, (m + 1) * 8, reinterpret_cast<void**>(&toff)) != hipSuccess) { pool_free(device, tc); return KH_ERR_NOMEM; }
  KH_SWITCH_HASH((int)hash, hipLaunchKernelGGL((k_shard_count<HASH>), dim3(ntiles), dim3(KH_SHARD_THREADS), 0, stream, keys, n, seed, p, pmask, tc, ntiles));
  hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, stream, tc, m, toff);
  if (!out_keys) {
    // count only: the caller sizes the exchange before it permutes (pipelined multi-GPU insert)
  } else if (p <= 8) {
    KH_SWITCH_HASH((int)hash, hipLaunchKernelGGL((k_shard_scatter8<HASH>), dim3(ntiles), dim3(KH_SHARD_THREADS), 0, stream, keys, vals, n, seed, p, pmask, toff, ntiles, out_keys, out_vals));
  } else {
    KH_SWITCH_HASH((int)hash, hipLaunchKernelGGL((k_shard_scatter<HASH>), dim3(ntiles), dim3(KH_SHARD_THREADS), 0, stream, keys, vals, n, seed, p, pmask, toff, ntiles, out_keys, out_vals));
  }
  std::vector<uint64_t> ends(p + 1);
  hipError_t e = hipGetLastError();
  for (uint32_t r = 0; r <= p && e == hipSuccess; ++r)
    e = hipMemcpyAsync(&ends[r], toff + (uint64_t)r * ntiles, 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  pool_free(device, tc); pool_free(device, toff);
  if (e != hipSuccess) return KH_ERR_HIP;
  for (uint32_t r = 0; r < p; ++r) counts_host[r] = ends[r + 1] - ends[r];
  return KH_OK;
}

// ---- shard plan: ONE count sweep + scan over a whole batch that is going to be exchanged in pieces; the per-piece destination
//      counts fall out of the scanned offsets at the piece boundaries (tile-aligned), and every piece is then permuted by the
//      scatter kernel alone (no second count, no second scan, one host synchronisation instead of one per piece)
struct kh_shard_plan {
  int device, hash; KhSeed seed; uint32_t p, pmask, pieces, ntiles; uint64_t n;
  uint32_t* tc; uint64_t* toff; uint64_t* bnd_dev;
  std::vector<uint64_t> bnd;      // [p][pieces+1] scanned offsets at the piece boundaries
};
kh_status kh_shard_plan_create(kh_shard_plan** out, kh_hash hash, uint64_t seed_, kh_key_transform xf, uint32_t k, uint32_t p, const uint64_t* keys,
                               uint64_t n, uint32_t pieces, uint64_t* counts_host, uint64_t* bounds_host, int device, void* stream_) {
  kh_table* t = nullptr;
  if (!out) return KH_ERR_INVALID;
  *out = nullptr;
  if (p == 0 || p > 8 || pieces == 0 || pieces > 64 || !counts_host || !bounds_host || (int)hash < 0 || (int)hash > 3 || !xform_ok(xf, k) || (n && !keys)) return KH_ERR_INVALID;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<kh_shard_plan> P(new kh_shard_plan());
  P->device = device; P->hash = (int)hash; P->seed = make_seed(seed_, xf, k); P->p = p; P->pmask = (p & (p - 1)) == 0 ? p - 1 : 0;
  P->pieces = pieces; P->n = n; P->tc = nullptr; P->toff = nullptr; P->bnd_dev = nullptr;
  P->ntiles = (uint32_t)((n + KH_SHARD_TILE - 1) / KH_SHARD_TILE);
  for (uint32_t i = 0; i <= pieces; ++i) bounds_host[i] = std::min<uint64_t>(n, ((uint64_t)P->ntiles * i / pieces) * KH_SHARD_TILE);
  for (uint64_t j = 0; j < (uint64_t)pieces * p; ++j) counts_host[j] = 0;
  P->bnd.assign((size_t)p * (pieces + 1), 0);
  if (n) {
    const uint64_t m = (uint64_t)p * P->ntiles;
    HIPCHK(pool_alloc(device, m * 4, reinterpret_cast<void**>(&P->tc)));
    if (pool_alloc(device, (m + 1) * 8, reinterpret_cast<void**>(&P->toff)) != hipSuccess ||
        pool_alloc(device, P->bnd.size() * 8, reinterpret_cast<void**>(&P->bnd_dev)) != hipSuccess) {
      if (P->toff) pool_free(device, P->toff);
      pool_free(device, P->tc);
      return KH_ERR_NOMEM;
    }
    KH_SWITCH_HASH((int)hash, hipLaunchKernelGGL((k_shard_count<HASH>), dim3(P->ntiles), dim3(KH_SHARD_THREADS), 0, stream, keys, n, P->seed, p, P->pmask, P->tc, P->ntiles));
    hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, stream, P->tc, m, P->toff);
    hipLaunchKernelGGL(k_shard_piece_bounds, dim3(1 + (uint32_t)P->bnd.size() / 256), dim3(256), 0, stream, (const uint64_t*)P->toff, P->ntiles, p, pieces, P->bnd_dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(P->bnd.data(), P->bnd_dev, P->bnd.size() * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { pool_free(device, P->tc); pool_free(device, P->toff); pool_free(device, P->bnd_dev); return KH_ERR_HIP; }
    for (uint32_t i = 0; i < pieces; ++i)
      for (uint32_t r = 0; r < p; ++r) counts_host[(size_t)i * p + r] = P->bnd[(size_t)r * (pieces + 1) + i + 1] - P->bnd[(size_t)r * (pieces + 1) + i];
  }
  *out = P.release();
  return KH_OK;
}
kh_status kh_shard_plan_permute(kh_shard_plan* P, uint32_t piece, const uint64_t* keys, const uint32_t* vals, uint64_t* out_keys, uint32_t* out_vals,
                                void* stream_) {
  kh_table* t = nullptr;
  if (!P || piece >= P->pieces || (vals && !out_vals)) return KH_ERR_INVALID;
  if (P->n == 0) return KH_OK;
  if (!keys || !out_keys) return KH_ERR_INVALID;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(P->device));
  const uint32_t t0 = (uint32_t)((uint64_t)P->ntiles * piece / P->pieces), t1 = (uint32_t)((uint64_t)P->ntiles * (piece + 1) / P->pieces);
  if (t1 == t0) return KH_OK;
  const uint64_t b0 = (uint64_t)t0 * KH_SHARD_TILE, b1 = std::min<uint64_t>(P->n, (uint64_t)t1 * KH_SHARD_TILE);
  KhShardAdj adj;
  uint64_t base = 0;      // where rank r's pairs of THIS piece start in the piece's output
  for (uint32_t r = 0; r < P->p; ++r) {
    const uint64_t lo = P->bnd[(size_t)r * (P->pieces + 1) + piece], hi = P->bnd[(size_t)r * (P->pieces + 1) + piece + 1];
    adj.a[r] = (long long)base - (long long)lo;
    base += hi - lo;
  }
  KH_SWITCH_HASH(P->hash, hipLaunchKernelGGL((k_shard_scatter8<HASH>), dim3(t1 - t0), dim3(KH_SHARD_THREADS), 0, stream, keys + b0, vals ? vals + b0 : nullptr, b1 - b0,
                                             P->seed, P->p, P->pmask, (const uint64_t*)P->toff, P->ntiles, out_keys, out_vals, t0, adj));
  HIPCHK(hipGetLastError());
  return KH_OK;
}
// piece `piece` written to its place in the layout of the WHOLE batch grouped by rank (what kh_shard_permute of all n pairs gives):
// the scanned [rank][tile] offsets already are positions in that layout
kh_status kh_shard_plan_permute_global(kh_shard_plan* P, uint32_t piece, const uint64_t* keys, const uint32_t* vals, uint64_t* out_keys, uint32_t* out_vals,
                                       void* stream_) {
  kh_table* t = nullptr;
  if (!P || piece >= P->pieces || (vals && !out_vals)) return KH_ERR_INVALID;
  if (P->n == 0) return KH_OK;
  if (!keys || !out_keys) return KH_ERR_INVALID;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(P->device));
  const uint32_t t0 = (uint32_t)((uint64_t)P->ntiles * piece / P->pieces), t1 = (uint32_t)((uint64_t)P->ntiles * (piece + 1) / P->pieces);
  if (t1 == t0) return KH_OK;
  const uint64_t b0 = (uint64_t)t0 * KH_SHARD_TILE, b1 = std::min<uint64_t>(P->n, (uint64_t)t1 * KH_SHARD_TILE);
  KhShardAdj adj;      // (all zero)
  KH_SWITCH_HASH(P->hash, hipLaunchKernelGGL((k_shard_scatter8<HASH>), dim3(t1 - t0), dim3(KH_SHARD_THREADS), 0, stream, keys + b0, vals ? vals + b0 : nullptr, b1 - b0,
                                             P->seed, P->p, P->pmask, (const uint64_t*)P->toff, P->ntiles, out_keys, out_vals, t0, adj));
  HIPCHK(hipGetLastError());
  return KH_OK;
}
kh_status kh_shard_plan_offsets(const kh_shard_plan* P, uint64_t* out) {
  if (!P || !out) return KH_ERR_INVALID;
  for (size_t i = 0; i < P->bnd.size(); ++i) out[i] = P->bnd[i];
  if (P->n == 0) for (size_t i = 0; i < P->bnd.size(); ++i) out[i] = 0;
  return KH_OK;
}
void kh_shard_plan_destroy(kh_shard_plan* P) {
  if (!P) return;
  if (P->tc) pool_free(P->device, P->tc);
  if (P->toff) pool_free(P->device, P->toff);
  if (P->bnd_dev) pool_free(P->device, P->bnd_dev);
  delete P;
}

// ---- k-mer generation front end ----------------------------------------------------------------------
}  // extern "C"
namespace {
// shared body of kh_kmers_from_sequence / kh_kmers_from_fastq
kh_status kmers_impl(const void* seq, uint64_t n, uint32_t k, int canonical, kh_mem where, bool fastq,
                     uint64_t* out_kmers, uint64_t* n_out, int device, void* stream_) {
  kh_table* t = nullptr;
  if (n_out) *n_out = 0;
  if (k < 1 || k > 32 || !n_out) return KH_ERR_INVALID;
  if (n < k) return KH_OK;
  if (!seq || !out_kmers) return KH_ERR_INVALID;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIPCHK(hipSetDevice(device));
  const uint64_t ntl = (n + KH_CMP_TILE - 1) / KH_CMP_TILE;            // tiles of the FASTQ line kernels
  const uint64_t nkt = (n + KH_KM_TILE - 1) / KH_KM_TILE;              // tiles of the k-mer kernels
  // one pooled block: [text copy (host input)] [masked text (FASTQ)] [tile sums] [tile offsets] [compacted out (host output)]
  const size_t sz_seq = where == KH_MEM_HOST ? ((n + 255) & ~size_t(255)) : 0;
  const size_t sz_msk = fastq ? ((n + 255) & ~size_t(255)) : 0;
  const size_t sz_sum = ((ntl * 4 + 255) & ~size_t(255)), sz_off = (ntl + 1) * 8;
  const size_t sz_out = where == KH_MEM_HOST ? n * 8 : 0;
  char* blk = nullptr;
  HIPCHK(pool_alloc(device, sz_seq + sz_msk + sz_sum + sz_off + 256 + sz_out, reinterpret_cast<void**>(&blk)));
  const uint8_t* dseq = static_cast<const uint8_t*>(seq);
  char* p = blk;
  if (where == KH_MEM_HOST) { dseq = reinterpret_cast<uint8_t*>(p); p += sz_seq; }
  uint8_t* msk = reinterpret_cast<uint8_t*>(p); p += sz_msk;
  uint32_t* sums = reinterpret_cast<uint32_t*>(p); p += sz_sum;
  uint64_t* offs = reinterpret_cast<uint64_t*>(p); p += (sz_off + 255) & ~size_t(255);
  uint64_t* dout = where == KH_MEM_HOST ? reinterpret_cast<uint64_t*>(p) : out_kmers;
  hipError_t e = hipSuccess;
  if (where == KH_MEM_HOST) e = hipMemcpyAsync(const_cast<uint8_t*>(dseq), seq, n, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    if (fastq) {      // keep the sequence lines only (line number = newlines before the byte; sequence lines are 1 mod 4)
      hipLaunchKernelGGL(k_newline_tile_sums, dim3((uint32_t)ntl), dim3(256), 0, stream, dseq, n, sums);
      hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, stream, sums, ntl, offs);
      hipLaunchKernelGGL(k_fastq_mask, dim3((uint32_t)ntl), dim3(256), 0, stream, dseq, n, offs, msk);
      dseq = msk;
    }
    // two passes over the text: valid windows per tile, scan, then the windows themselves, compacted and in order
    hipLaunchKernelGGL(k_kmers_count, dim3((uint32_t)nkt), dim3(KH_KM_THREADS), 0, stream, dseq, n, k, sums);
    hipLaunchKernelGGL(k_scan_u32_to_u64, dim3(1), dim3(KH_SCAN_THREADS), 0, stream, sums, nkt, offs);
    if (canonical) hipLaunchKernelGGL((k_kmers_emit<true>), dim3((uint32_t)nkt), dim3(KH_KM_THREADS), 0, stream, dseq, n, k, (const uint64_t*)offs, dout);
    else hipLaunchKernelGGL((k_kmers_emit<false>), dim3((uint32_t)nkt), dim3(KH_KM_THREADS), 0, stream, dseq, n, k, (const uint64_t*)offs, dout);
    e = hipGetLastError();
  }
  uint64_t total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, offs + nkt, 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess && where == KH_MEM_HOST && total) {
    e = hipMemcpyAsync(out_kmers, dout, total * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  pool_free(device, blk);
  if (e != hipSuccess) return KH_ERR_HIP;
  *n_out = total;
  return KH_OK;
}
}  // namespace
extern "C" {
kh_status kh_kmers_from_sequence(const void* seq, uint64_t n, uint32_t k, int canonical, kh_mem where,
                                 uint64_t* out_kmers, uint64_t* n_out, int device, void* stream_) {
  return kmers_impl(seq, n, k, canonical, where, false, out_kmers, n_out, device, stream_);
}
kh_status kh_kmers_from_fastq(const void* text, uint64_t n, uint32_t k, int canonical, kh_mem where,
                              uint64_t* out_kmers, uint64_t* n_out, int device, void* stream_) {
  return kmers_impl(text, n, k, canonical, where, true, out_kmers, n_out, device, stream_);
}

// ---- HyperLogLog (hyperloglog64.hpp) --------------------------------------------------------------
}  // extern "C"
struct kh_hll {
  int device, hash; uint64_t seed; uint32_t precision, ignored; uint32_t* regs; hipStream_t stream;
};
extern "C" {
kh_status kh_hll_create(kh_hll** out, uint32_t precision, uint32_t ignore_msb, kh_hash hash, uint64_t seed, int device) {
  kh_table* t = nullptr;
  if (!out) return KH_ERR_INVALID;
  *out = nullptr;
  if (precision < 4 || precision > 18 || precision + ignore_msb >= 64 || (int)hash < 0 || (int)hash > 3) return KH_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return KH_ERR_HIP;
  HIPCHK(hipSetDevice(device));
  kh_hll* h = new kh_hll();
  h->device = device; h->hash = (int)hash; h->seed = seed; h->precision = precision; h->ignored = ignore_msb; h->stream = nullptr; h->regs = nullptr;
  if (pool_alloc(device, sizeof(uint32_t) << precision, reinterpret_cast<void**>(&h->regs)) != hipSuccess) { delete h; return KH_ERR_NOMEM; }
  if (hipMemset(h->regs, 0, sizeof(uint32_t) << precision) != hipSuccess) { pool_free(device, h->regs); delete h; return KH_ERR_HIP; }
  *out = h;
  return KH_OK;
}
kh_status kh_hll_destroy(kh_hll* h) {
  if (!h) return KH_OK;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  pool_free(h->device, h->regs);
  delete h;
  return KH_OK;
}
kh_status kh_hll_set_stream(kh_hll* h, void* s) { if (!h) return KH_ERR_INVALID; h->stream = static_cast<hipStream_t>(s); return KH_OK; }
static kh_status hll_update(kh_hll* h, const void* in, uint64_t n, kh_mem where, bool from_keys) {
  kh_table* t = nullptr;
  if (!h) return KH_ERR_INVALID;
  if (n == 0) return KH_OK;
  if (!in) return KH_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  const uint64_t* d = static_cast<const uint64_t*>(in);
  uint64_t* tmp = nullptr;
  if (where == KH_MEM_HOST) {
    HIPCHK(pool_alloc(h->device, n * 8, reinterpret_cast<void**>(&tmp)));
    hipError_t e = hipMemcpyAsync(tmp, in, n * 8, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { pool_free(h->device, tmp); return KH_ERR_HIP; }
    d = tmp;
  }
  const int use_lds = h->precision <= 13 ? 1 : 0;
  const size_t smem = use_lds ? (sizeof(uint32_t) << h->precision) : 0;
  const uint32_t grid = grid_for(n, 256, 1024);
  if (from_keys) {
    KH_SWITCH_HASH(h->hash, hipLaunchKernelGGL((k_hll_update<HASH, true>), dim3(grid), dim3(256), smem, h->stream, d, n, KhSeed{h->seed, 0u}, h->precision, h->ignored, h->regs, use_lds));
  } else {
    hipLaunchKernelGGL((k_hll_update<KHH_IDENTITY, false>), dim3(grid), dim3(256), smem, h->stream, d, n, KhSeed{h->seed, 0u}, h->precision, h->ignored, h->regs, use_lds);
  }
  hipError_t e = hipGetLastError();
  if (tmp) { if (e == hipSuccess) e = hipStreamSynchronize(h->stream); pool_free(h->device, tmp); }
  return e == hipSuccess ? KH_OK : KH_ERR_HIP;
}
kh_status kh_hll_update(kh_hll* h, const void* keys, uint64_t n, kh_mem where) { return hll_update(h, keys, n, where, true); }
kh_status kh_hll_update_via_hashval(kh_hll* h, const void* hashes, uint64_t n, kh_mem where) { return hll_update(h, hashes, n, where, false); }
kh_status kh_hll_merge(kh_hll* h, const kh_hll* other) {
  kh_table* t = nullptr;
  if (!h || !other || h->precision != other->precision || h->device != other->device) return KH_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  const uint32_t m = 1u << h->precision;
  hipLaunchKernelGGL(k_hll_merge, dim3((m + 255) / 256), dim3(256), 0, h->stream, h->regs, other->regs, m);
  HIPCHK(hipGetLastError());
  return KH_OK;
}
kh_status kh_hll_clear(kh_hll* h) {
  kh_table* t = nullptr;
  if (!h) return KH_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemsetAsync(h->regs, 0, sizeof(uint32_t) << h->precision, h->stream));
  return KH_OK;
}
kh_status kh_hll_registers(kh_hll* h, uint8_t* out_host) {
  kh_table* t = nullptr;
  if (!h || !out_host) return KH_ERR_INVALID;
  HIPCHK(hipSetDevice(h->device));
  const uint32_t m = 1u << h->precision;
  std::vector<uint32_t> r(m);
  HIPCHK(hipMemcpyAsync(r.data(), h->regs, sizeof(uint32_t) * m, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (uint32_t i = 0; i < m; ++i) out_host[i] = (uint8_t)r[i];
  return KH_OK;
}
// internal_estimate (hyperloglog64.hpp:201-236): same operation order in double, so equal registers give an equal double
kh_status kh_hll_estimate(kh_hll* h, double* out) {
  if (!h || !out) return KH_ERR_INVALID;
  const uint32_t m = 1u << h->precision;
  std::vector<uint8_t> regs(m);
  kh_status st = kh_hll_registers(h, regs.data());
  if (st != KH_OK) return st;
  return kh_hll_estimate_registers(regs.data(), h->precision, out);
}
kh_status kh_hll_estimate_registers(const uint8_t* regs, uint32_t precision, double* out) {
  if (!regs || !out || precision < 4 || precision > 18) return KH_ERR_INVALID;
  const uint32_t m = 1u << precision;
  struct { uint32_t precision; } hh = {precision}; auto* h = &hh;
  double amm;
  switch (h->precision) {
    case 4: amm = 0.673; break;
    case 5: amm = 0.697; break;
    case 6: amm = 0.709; break;
    default: amm = 0.7213 / (1.0 + (1.079 / static_cast<double>(m))); break;
  }
  amm *= static_cast<double>(0x1ULL << (h->precision << 1U));
  double sum = 0.0;
  uint32_t zeros = 0;
  for (uint32_t i = 0; i < m; ++i) { sum += 1.0 / static_cast<double>(1ULL << regs[i]); if (regs[i] == 0) ++zeros; }
  double est = amm / sum;
  if (est <= static_cast<double>(5ULL * (m >> 1ULL))) {
    if (zeros > 0) est = static_cast<double>(m) * std::log(static_cast<double>(m) / static_cast<double>(zeros));
  }
  *out = est;
  return KH_OK;
}

kh_status kh_profile_enable(kh_table* t, int on) { if (!t) return KH_ERR_INVALID; prof_collect(t); t->prof = on != 0; return KH_OK; }
kh_status kh_profile_reset(kh_table* t) { if (!t) return KH_ERR_INVALID; prof_collect(t); t->prof_acc.clear(); return KH_OK; }
kh_status kh_profile_query(kh_table* t, const char* prefix, double* total_ms, uint64_t* launches) {
  if (!t || !prefix) return KH_ERR_INVALID;
  prof_collect(t);
  double ms = 0; uint64_t n = 0;
  const size_t pl = strlen(prefix);
  for (auto& a : t->prof_acc) if (a.first.compare(0, pl, prefix) == 0) { ms += a.second.first; n += a.second.second; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return KH_OK;
}
kh_status kh_profile_dump(kh_table* t, char* buf, uint64_t cap) {
  if (!t || !buf || cap == 0) return KH_ERR_INVALID;
  prof_collect(t);
  std::string s;
  char line[256];
  for (auto& a : t->prof_acc) {
    snprintf(line, sizeof(line), "%s %llu %.6f\n", a.first.c_str(), (unsigned long long)a.second.second, a.second.first);
    s += line;
  }
  snprintf(buf, cap, "%s", s.c_str());
  return KH_OK;
}

}  // extern "C"

#ifdef KH_TRACE
extern "C" int kh_debug_trace(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kh_trace), sizeof(unsigned long long) * 512 * 12); }
#endif
