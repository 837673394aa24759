// kh_kernels.h -- gfx950 device code of libkmerhash_amd (included once by kmerhash_amd.hip).
//
// Design.  The table is ONE array of 16-byte slots in HBM, {u64 key, u32 value, u32 info} per bucket (KhSlot; the low
// byte of `info` is the reference's info byte, bit 8 the erase mark of a batch erase on its fall-back path), read and written with one
// dwordx4 access, and is processed in CHUNKS of KH_L = 2^KH_LB consecutive home buckets (32 KB of table).
//   * A large mutating batch never probes HBM at random: it is radix-partitioned by (bit-reversed) chunk id with coalesced streaming
//     passes (k_part_scatter: 16-, 12- or 8-byte records; histogram-free into fixed slots when a sample finds no duplicates), and every
//     chunk of the new table is laid out in its canonical Robin Hood order (elements sorted by home bucket, slot = max(home, previous
//     slot + 1)) by one workgroup that histograms the chunk's home buckets in LDS and runs a (max,+) scan over them; run-over between
//     chunks is a (max,+) carry with a one-deep look-back.  k_build_fused does all of that in ONE launch, from partition records
//     (SRC 0: bulk build into an empty table), from the table itself (SRC 1: rehash / reserve / the erase fall-back), from both
//     folded together in LDS (SRC 2: insert into a non-empty table) or from the table minus the chunk's erase keys (SRC 3: batch
//     erase).  The benchmark case of SRC 0 (distinct keys, one source of 12-byte records) has its own kernel, k_build_lean (four
//     workgroups per CU; sort without the carry-in, look-back collected late), and so has the batch erase: k_erase_stream scans the
//     chunk's slots IN SLOT ORDER (a Robin Hood table is sorted by home bucket already) -- no staging, counting or ranking; and the insert
//     into a loaded table whose capacity stays: k_insert_stream (the table's elements in slot order, the batch's records chained per bucket).
//     The general path is k_dedup -> k_chunk_count -> k_chunk_carry -> k_chunk_place.
//   * Batches of <= 16 keys are applied in place by one lane (k_small_batch), mid-size batches in place by one lane per region of
//     512 slots (k_ip_bin / k_ip_apply / k_ip_serial).
//   * Read-only batches (k_find: find, count) probe the table one 64-byte SECTOR (4 aligned slots) at a time, four queries per lane
//     in flight, and compact their hits inside the same launch (wave ballots + decoupled look-back over the tiles).
// Wave64 everywhere; no MFMA (integer/indexing path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kmerhash_amd/kh_hash.h"

#define KH_LB 11                 // log2(chunk slots)
#define KH_L (1u << KH_LB)       // chunk slots (home buckets per workgroup)
#define KH_HS 4096u              // LDS de-dup set entries per workgroup (16 B each)
#define KH_CHUNK_THREADS 512
#ifndef KH_PART_THREADS             // (overridable for experiments: -DKH_PART_THREADS=1024 -DKH_PART_MAXPER=2)
#define KH_PART_THREADS 512
#define KH_PART_MAXPER 4            // digits per lane in the scatter's scan: nb <= 2048 bins
#endif
#ifndef KH_PART_ITEMS
#define KH_PART_ITEMS 16
#endif
#define KH_PART_TILE (KH_PART_THREADS * KH_PART_ITEMS)   // 8192 records per partition tile: ONE reservation per (tile, digit)
#ifndef KH_PART_STAGE
#define KH_PART_STAGE 4096       // records staged in LDS at a time: the tile is streamed out in TILE/STAGE rounds
#endif
#define KH_NONE 0xFFFFFFFFFFFFFFFFull
#define KH_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

enum { KHK_RH = 0, KHK_LP = 1 };
enum { KH_FLAG_PROBE_OVERFLOW = 0, KH_FLAG_REGION_OVERFLOW = 1, KH_FLAG_COUNT_OVERFLOW = 2, KH_FLAG_INTERNAL = 3, KH_FLAG_FUSE_INVALID = 4, KH_NFLAGS = 8 };

// info-byte predicates of the two reference encodings
template <int KIND> __device__ __forceinline__ bool kh_is_empty(uint32_t b) { return KIND == KHK_RH ? (b == 0x00u) : (b == 0x40u); }
template <int KIND> __device__ __forceinline__ bool kh_is_occupied(uint32_t b) { return KIND == KHK_RH ? (b >= 0x80u) : (b < 0x40u); }

// One bucket of the table: 16 bytes, naturally aligned, read and written with ONE dwordx4 access.  A probe needs the info byte,
// the key and (on a hit) the value of the same bucket: co-locating them makes a find touch one 64-byte sector of HBM where
// the SoA layout of round 1 touched three (info / keys / values arrays: 3.9 memory requests per query measured).
//   info: low byte = the reference's info_type byte (RH: 0x00 empty, 0x80|distance; LP: 0x40 empty, 0x80 deleted, 0x00 normal);
//         bit 8 (KH_INFO_ERASE_MARK) = "dropped by the re-layout that follows" (Robin Hood batch erase), invisible to probes;
//         the other bits are zero.
struct __align__(16) KhSlot { uint64_t key; uint32_t val; uint32_t info; };
#define KH_INFO_ERASE_MARK 0x100u
struct KhSlots {
  KhSlot* s;
  uint64_t cap;   // power of two
};
__device__ __forceinline__ uint4 kh_slot_ld(const KhSlot* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void kh_slot_st(KhSlot* p, uint64_t key, uint32_t val, uint32_t info) {
  uint4 w; w.x = (uint32_t)key; w.y = (uint32_t)(key >> 32); w.z = val; w.w = info;
  *reinterpret_cast<uint4*>(p) = w;
}
__device__ __forceinline__ uint64_t kh_slot_key(const uint4& w) { return (uint64_t)w.x | ((uint64_t)w.y << 32); }
template <int KIND> __device__ __forceinline__ uint32_t kh_empty_info() { return KIND == KHK_RH ? 0x00u : 0x40u; }

// ---------------------------------------------------------------------------------------------
// direct probing (find_pos): hashmap_robinhood.hpp:1058-1095 / hashmap_linearprobe.hpp:693-748
// Every step is one 16-byte load; consecutive buckets share a 64-byte sector three times out of four.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ uint64_t kh_find_pos(const KhSlot* __restrict__ slots, uint64_t mask, uint64_t home, uint64_t key, uint32_t* val_out = nullptr,
                                                uint32_t xk = 0) {
  uint64_t i = home;
  if (KIND == KHK_RH) {
    // reprobe = 0x80 + distance; stop as soon as the resident entry is "richer" (or the slot is empty).
    for (uint32_t reprobe = 0x80u; reprobe < 0x100u; ++reprobe) {
      const uint4 w = kh_slot_ld(slots + i);
      const uint32_t b = w.w & 0xFFu;
      if (reprobe > b) return KH_NONE;
      if (reprobe == b && kh_keq(kh_slot_key(w), key, xk)) { if (val_out) *val_out = w.z; return i; }
      i = (i + 1) & mask;
    }
    return KH_NONE;
  } else {
    // two-segment scan of the reference == circular scan; stop at empty, skip deleted
    for (uint64_t step = 0; step <= mask; ++step) {
      const uint4 w = kh_slot_ld(slots + i);
      const uint32_t b = w.w & 0xFFu;
      if (b == 0x40u) return KH_NONE;
      if (b < 0x40u && kh_keq(kh_slot_key(w), key, xk)) { if (val_out) *val_out = w.z; return i; }
      i = (i + 1) & mask;
    }
    return KH_NONE;
  }
}

// ---------------------------------------------------------------------------------------------
// batched hashing  (Hash::operator()(Key const*, count, out))
// ---------------------------------------------------------------------------------------------
template <int HASH>
__global__ void k_hash_batch(const uint64_t* __restrict__ keys, uint64_t n, KhSeed seed, uint64_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = kh_hash64<HASH>(keys[i], seed);
}

// ---------------------------------------------------------------------------------------------
// count / find : a workgroup takes a TILE of KH_Q_TILE queries (dynamic ticket), every lane owns KH_Q_ITEMS of them
// (query tile_base + j * 256 + tid: a wave's loads and stores of one j are contiguous).  The home buckets of all of a lane's
// queries are requested before the first one is looked at (8 independent 16-byte loads in flight per lane), and the rare
// continued probes advance in rounds, again all unresolved queries of the lane at once.
// find(Iter,Iter) returns the hits only, in query order (hashmap_robinhood.hpp:1194-1268): the compaction happens in the SAME
// launch.  The hits of a tile are ranked with wave ballots; the tile's position in the output comes from a decoupled
// look-back over the tiles before it: every tile publishes its hit count as one self-contained 8-byte granule
// {state, value} (relaxed agent-scope store: the data is the word itself, MI355X guide G16 "R2 granule"), first as an
// AGGREGATE, then -- once it knows what precedes it -- as an inclusive PREFIX; a tile sums aggregates backwards until it meets
// a prefix (one wave inspects 64 predecessors per step).  Tiles are handed out by a ticket, so every predecessor has at least
// been started: the wait is bounded by their run time, not by the dispatcher's order.
// ---------------------------------------------------------------------------------------------
#define KH_Q_THREADS 256
#define KH_Q_ITEMS 4
#define KH_Q_NB 4                // batches of KH_Q_ITEMS queries per lane and tile (one look-back per tile)
#define KH_Q_TILE (KH_Q_THREADS * KH_Q_ITEMS * KH_Q_NB)
#define KH_LB_AGG (1ull << 62)
#define KH_LB_PRE (2ull << 62)
#define KH_LB_VAL ((1ull << 62) - 1ull)
enum { KH_FIND_PERQUERY = 0, KH_FIND_COMPACT = 1, KH_FIND_PAIRS = 2, KH_FIND_COUNT = 3 };

struct KhFindParams {
  KhSlots T; const uint64_t* q; uint64_t n; KhSeed seed;
  uint32_t* out_vals; uint8_t* out_found;          // PERQUERY: value (hits only) + 0/1 flag per query; COUNT: out_found only
  uint64_t* out_keys; uint8_t* out_pairs16;        // COMPACT: out_keys + out_vals; PAIRS: 16-byte (key, value, 0) records
  uint32_t* ticket;                                // zero at launch
  unsigned long long* tile_state;                  // [tiles] zero at launch (COMPACT / PAIRS)
  unsigned long long* n_found;                     // zero at launch: total hits
};

// probes the KH_Q_ITEMS queries of this lane; returns the bit mask of hits, vals[j] of the hits.
// A probe reads the table one 64-byte SECTOR (KH_Q_W = 4 aligned slots) at a time: the four 16-byte loads of a sector are
// in flight together, so a chain of d slots costs ceil((d + offset) / 4) dependent memory round trips instead of d -- the
// rounds a wave spends on the longest chain among its 256 queries (12-15 slots at load 0.8) drop from ~14 to ~4.
#define KH_Q_W 4
template <int KIND, int HASH, bool XF>
__device__ __forceinline__ uint32_t kh_probe_items(const KhSlots& T, const uint64_t (&key)[KH_Q_ITEMS], uint32_t valid, KhSeed seed_,
                                                   uint32_t (&val)[KH_Q_ITEMS], uint32_t* swapped = nullptr, uint64_t* at_out = nullptr) {
  // at_out (erase): slot index of hit j
  // swapped (key transform active): bit j set when the stored key of hit j is not the query's own bit pattern but its
  // equivalent under the transform (the reverse complement): find returns the STORED pair (hashmap_robinhood.hpp:1194-1268)
  // XF = false: the table has no key transform -- the compiler drops every trace of it from the hot kernel
  const KhSeed seed = KhSeed{seed_.s, XF ? seed_.xk : 0u};
  const uint64_t mask = T.cap - 1;
  const uint32_t xk = seed.xk;
  uint32_t hit = 0, swp = 0;
  if (T.cap < KH_Q_W) {                                // tables of 1 or 2 buckets: slot by slot
#pragma unroll
    for (int j = 0; j < KH_Q_ITEMS; ++j)
      if ((valid >> j) & 1u) {
        const uint64_t at = kh_find_pos<KIND>(T.s, mask, kh_hash64<HASH>(key[j], seed) & mask, key[j], &val[j], xk);
        if (at != KH_NONE) { hit |= 1u << j; if (xk && T.s[at].key != key[j]) swp |= 1u << j; if (at_out) at_out[j] = at; }
      }
    if (swapped) *swapped = swp;
    return hit;
  }
  uint64_t base[KH_Q_ITEMS];                           // first slot of the sector being looked at
  uint32_t first[KH_Q_ITEMS];                          // slots of that sector in front of the home bucket (first sector only)
  uint32_t dist[KH_Q_ITEMS];                           // probe distance of slot base + first
  uint4 w[KH_Q_ITEMS][KH_Q_W];
#pragma unroll
  for (int j = 0; j < KH_Q_ITEMS; ++j) {
    const uint64_t home = kh_hash64<HASH>(key[j], seed) & mask;
    base[j] = home & ~(uint64_t)(KH_Q_W - 1); first[j] = (uint32_t)(home & (KH_Q_W - 1)); dist[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < KH_Q_ITEMS; ++j)
    if ((valid >> j) & 1u) {
#pragma unroll
      for (int s = 0; s < KH_Q_W; ++s) w[j][s] = kh_slot_ld(T.s + base[j] + s);
    }
  uint32_t active = valid;
  const uint64_t max_steps = KIND == KHK_RH ? 128u : T.cap;
  while (active) {
#pragma unroll
    for (int j = 0; j < KH_Q_ITEMS; ++j) {
#pragma unroll
      for (int s = 0; s < KH_Q_W; ++s) {
        if (((active >> j) & 1u) && (uint32_t)s >= first[j]) {
          const uint32_t b = w[j][s].w & 0xFFu;
          if (KIND == KHK_RH) {
            const uint32_t reprobe = 0x80u + dist[j] + (uint32_t)s - first[j];
            if (reprobe > b || reprobe > 0xFFu) active &= ~(1u << j);                               // richer resident or empty: absent
            else if (reprobe == b && kh_keq(kh_slot_key(w[j][s]), key[j], xk)) { hit |= 1u << j; val[j] = w[j][s].z; active &= ~(1u << j); if (xk && kh_slot_key(w[j][s]) != key[j]) swp |= 1u << j; if (at_out) at_out[j] = base[j] + s; }
          } else {
            if (b == 0x40u) active &= ~(1u << j);
            else if (b < 0x40u && kh_keq(kh_slot_key(w[j][s]), key[j], xk)) { hit |= 1u << j; val[j] = w[j][s].z; active &= ~(1u << j); if (xk && kh_slot_key(w[j][s]) != key[j]) swp |= 1u << j; if (at_out) at_out[j] = base[j] + s; }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KH_Q_ITEMS; ++j) {
      if ((active >> j) & 1u) {
        dist[j] += KH_Q_W - first[j]; first[j] = 0;
        if ((uint64_t)dist[j] >= max_steps) { active &= ~(1u << j); continue; }
        base[j] = (base[j] + KH_Q_W) & mask;
#pragma unroll
        for (int s = 0; s < KH_Q_W; ++s) w[j][s] = kh_slot_ld(T.s + base[j] + s);
      }
    }
  }
  if (swapped) *swapped = swp;
  return hit;
}

template <int KIND, int HASH, int OUT, bool XF>
__global__ __launch_bounds__(KH_Q_THREADS) void k_find(KhFindParams P) {
  __shared__ uint32_t s_tile;
  __shared__ __align__(16) uint32_t s_wcnt[KH_Q_NB * KH_Q_ITEMS][KH_Q_THREADS / 64];
  __shared__ unsigned long long s_prefix;
  __shared__ uint32_t s_val[(OUT == KH_FIND_COMPACT || OUT == KH_FIND_PAIRS) ? KH_Q_TILE : 1];      // values of the tile's hits (16 KB)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t ntiles = (uint32_t)((P.n + KH_Q_TILE - 1) / KH_Q_TILE);
  uint32_t acc = 0;
  for (;;) {
    __syncthreads();                       // s_tile / s_wcnt / s_prefix of the previous tile have been read by every lane
    if (tid == 0) s_tile = atomicAdd(P.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= ntiles) {
      // per-query forms: the hit count was kept in registers over all the tiles of this workgroup -- one atomic per wave
      // at exit (one per wave and tile were 39 K returning same-address atomics for 10^7 queries: 0.3 ms of serialisation)
      if ((OUT == KH_FIND_PERQUERY || OUT == KH_FIND_COUNT) && P.n_found) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0 && acc) atomicAdd(P.n_found, (unsigned long long)acc);
      }
      return;
    }
    const uint64_t base = (uint64_t)tile * KH_Q_TILE;
    // the tile is probed in KH_Q_NB batches of KH_Q_ITEMS queries per lane; one look-back per tile ranks all of them.  The
    // batch loop stays rolled (the registers of a batch -- 4 sectors of 4 slots per lane -- are reused by the next one); the
    // values of the hits wait in LDS
    uint32_t hit = 0, swapped = 0;                           // bit b * KH_Q_ITEMS + j
#pragma unroll 1
    for (int b = 0; b < KH_Q_NB; ++b) {
      uint64_t key[KH_Q_ITEMS]; uint32_t v[KH_Q_ITEMS];
      uint32_t valid = 0;
#pragma unroll
      for (int j = 0; j < KH_Q_ITEMS; ++j) {
        const uint64_t i = base + (uint64_t)(b * KH_Q_ITEMS + j) * KH_Q_THREADS + tid;
        key[j] = 0; v[j] = 0;
        if (i < P.n) { key[j] = P.q[i]; valid |= 1u << j; }
      }
      if (!__any(valid != 0)) continue;                      // (the last tile may end early)
      uint32_t sw = 0;
      const uint32_t h = kh_probe_items<KIND, HASH, XF>(P.T, key, valid, P.seed, v, XF ? &sw : nullptr);
      hit |= h << (b * KH_Q_ITEMS);
      swapped |= sw << (b * KH_Q_ITEMS);
      if (OUT == KH_FIND_PERQUERY || OUT == KH_FIND_COUNT) {
#pragma unroll
        for (int j = 0; j < KH_Q_ITEMS; ++j) {
          const uint64_t i = base + (uint64_t)(b * KH_Q_ITEMS + j) * KH_Q_THREADS + tid;
          if ((valid >> j) & 1u) {
            const bool f = (h >> j) & 1u;
            P.out_found[i] = f ? 1 : 0;
            if (OUT == KH_FIND_PERQUERY && f && P.out_vals) P.out_vals[i] = v[j];
            acc += f ? 1u : 0u;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < KH_Q_ITEMS; ++j) s_val[(b * KH_Q_ITEMS + j) * KH_Q_THREADS + tid] = v[j];
      }
    }
    if (OUT == KH_FIND_PERQUERY || OUT == KH_FIND_COUNT) continue;
    // ---- hits of the tile per (batch, item) and wave (query order = (batch, item)-major, then lane); the rank of every hit is
    // derived from these counts again at the write-out (16 ranks kept in registers cost the fourth wave per SIMD)
    static_assert(KH_Q_THREADS / 64 == 4, "s_wcnt rows are read as one uint4");
#pragma unroll
    for (int j = 0; j < KH_Q_NB * KH_Q_ITEMS; ++j) {
      const unsigned long long m = __ballot((hit >> j) & 1u);
      if (lane == 0) s_wcnt[j][wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < KH_Q_NB * KH_Q_ITEMS; ++j) {
      const uint4 c = *reinterpret_cast<const uint4*>(&s_wcnt[j][0]);
      total += c.x + c.y + c.z + c.w;
    }
    // ---- decoupled look-back (wave 0): exclusive prefix of this tile over the tiles before it
    if (wid == 0) {
      if (lane == 0 && tile + 1 < ntiles)
        __hip_atomic_store(&P.tile_state[tile], (tile == 0 ? KH_LB_PRE : KH_LB_AGG) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long excl = 0;
      int64_t look = (int64_t)tile - 1;             // lane l inspects tile look - l
      unsigned long long st = 0;
      while (look >= 0) {
        const int64_t t = look - (int64_t)lane;
        // (re)load only what has not been seen published yet: 1280 resident workgroups x 64 lanes polling in a tight loop
        // would keep the L2 busier than the probes do (MI355X guide, polling-cost)
        if (t < 0) st = KH_LB_PRE;                  // lanes before tile 0 behave as a zero prefix
        else if (!(st >> 62)) st = __hip_atomic_load(&P.tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ready = __ballot((st >> 62) != 0ull);
        const unsigned long long pre = __ballot((st >> 62) == 2ull);
        const uint32_t first_nr = ~ready ? (uint32_t)__ffsll((long long)~ready) - 1u : 64u;
        const uint32_t first_pre = pre ? (uint32_t)__ffsll((long long)pre) - 1u : 64u;
        if (first_pre < first_nr) {                 // a prefix with every aggregate between it and this tile: done
          unsigned long long v = lane <= first_pre ? (st & KH_LB_VAL) : 0ull;
          for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          excl += v;
          break;
        }
        if (first_nr == 64u) {                      // 64 aggregates, no prefix among them: next window
          unsigned long long v = st & KH_LB_VAL;
          for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          excl += v;
          look -= 64;
          st = 0;
          continue;
        }
        __builtin_amdgcn_s_sleep(20);               // a predecessor is still probing: ~0.5 us, then look again
      }
      if (lane == 0) {
        if (tile != 0 && tile + 1 < ntiles)
          __hip_atomic_store(&P.tile_state[tile], KH_LB_PRE | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile + 1 == ntiles) *P.n_found = excl + total;
        s_prefix = excl;
      }
    }
    __syncthreads();
    const unsigned long long obase = s_prefix;
    // the keys of the hits are read again (the batch's registers were reused; the tile's 32 KB of queries are L2-hot): four
    // loads in flight per lane, clamped indices instead of a branch per item (which makes every load wait for the one before)
    uint32_t run = 0;                               // hits of the tile in the (batch, item) rows before j
#pragma unroll
    for (int g = 0; g < KH_Q_NB * KH_Q_ITEMS; g += 4) {
      uint64_t kk[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint64_t i = base + (uint64_t)(g + jj) * KH_Q_THREADS + tid;
        kk[jj] = P.q[i < P.n ? i : P.n - 1];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = g + jj;
        const unsigned long long m = __ballot((hit >> j) & 1u);
        const uint4 c = *reinterpret_cast<const uint4*>(&s_wcnt[j][0]);
        const uint32_t before = run + (wid > 0 ? c.x : 0u) + (wid > 1 ? c.y : 0u) + (wid > 2 ? c.z : 0u) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        run += c.x + c.y + c.z + c.w;
        if ((hit >> j) & 1u) {
          uint64_t key = kk[jj];
          if (XF && ((swapped >> j) & 1u)) key = kh_revcomp(key, P.seed.xk);        // the table holds the other strand's bit pattern
          const unsigned long long o = obase + before;
          if (OUT == KH_FIND_PAIRS) {
            uint4 w; w.x = (uint32_t)key; w.y = (uint32_t)(key >> 32); w.z = s_val[j * KH_Q_THREADS + tid]; w.w = 0;
            *reinterpret_cast<uint4*>(P.out_pairs16 + o * 16) = w;
          } else {
            P.out_keys[o] = key;
            P.out_vals[o] = s_val[j * KH_Q_THREADS + tid];
          }
        }
      }
    }
  }
}

// RH: hits are MARKED in the slot's info word (bit 8: invisible to probes) and the table is then re-laid-out without the
//     marked elements (== backward-shift deletion, hashmap_robinhood.hpp:1294-1356).
// LP: tombstone in place (info = 0x80), hashmap_linearprobe.hpp:911-978.
// A key listed twice in the batch erases once: the atomic decides who counts it.
template <int KIND, int HASH>
__global__ __launch_bounds__(KH_Q_THREADS) void k_erase_mark(KhSlots T, const uint64_t* __restrict__ q, uint64_t n, KhSeed seed,
                                                          unsigned long long* __restrict__ n_erased) {
  // probing as in k_find: KH_Q_ITEMS queries per lane, one 64-byte sector per round (0.63 -> 0.3x ms per 10^7 keys against
  // the slot-by-slot loop)
  const uint64_t stride = (uint64_t)gridDim.x * KH_Q_THREADS * KH_Q_ITEMS;
  uint32_t mine = 0;
  for (uint64_t base = (uint64_t)blockIdx.x * KH_Q_THREADS * KH_Q_ITEMS; base < n; base += stride) {
    uint64_t key[KH_Q_ITEMS], at[KH_Q_ITEMS]; uint32_t val[KH_Q_ITEMS];
    uint32_t valid = 0;
#pragma unroll
    for (int j = 0; j < KH_Q_ITEMS; ++j) {
      const uint64_t i = base + (uint64_t)j * KH_Q_THREADS + threadIdx.x;
      key[j] = 0; at[j] = 0; val[j] = 0;
      if (i < n) { key[j] = q[i]; valid |= 1u << j; }
    }
    const uint32_t hit = kh_probe_items<KIND, HASH, true>(T, key, valid, seed, val, nullptr, at);
#pragma unroll
    for (int j = 0; j < KH_Q_ITEMS; ++j) {
      if (!((hit >> j) & 1u)) continue;
      if (KIND == KHK_RH) {
        const uint32_t old = atomicOr(&T.s[at[j]].info, KH_INFO_ERASE_MARK);
        if (!(old & KH_INFO_ERASE_MARK)) ++mine;
      } else {
        const uint32_t old = atomicOr(&T.s[at[j]].info, 0x80u);
        if ((old & 0xFFu) < 0x40u) ++mine;
      }
    }
  }
  // one atomic per workgroup (same-address atomics serialise in the L2 at ~12 ns each: 16 K waves would spend 0.2 ms there)
  __shared__ uint32_t s_mine[KH_Q_THREADS / 64];
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0) s_mine[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < KH_Q_THREADS / 64; ++w) tot += s_mine[w];
    if (tot) atomicAdd(n_erased, (unsigned long long)tot);
  }
}
// a batch erase whose re-layout could not run (no memory for the new buffer) takes its marks back
__global__ void k_clear_marks(KhSlots T) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.cap; i += stride)
    if (T.s[i].info & KH_INFO_ERASE_MARK) T.s[i].info &= 0xFFu;
}
// every slot of a table empty (kh_create, clear(); re-layouts write each slot of their destination themselves)
template <int KIND>
__global__ void k_fill_empty(KhSlots T) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.cap; i += stride) kh_slot_st(T.s + i, 0, 0, kh_empty_info<KIND>());
}
// test hook (KH_DEBUG_POISON): a destination buffer starts as garbage that no probe would accept as empty
__global__ void k_poison(KhSlots T) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.cap; i += stride) kh_slot_st(T.s + i, 0xEEEEEEEEEEEEEEEEull, 0xEEEEEEEEu, 0xEEu);
}

// ---------------------------------------------------------------------------------------------
// stream compaction of find hits (find(Iter,Iter) returns only the hits, in query order)
// ---------------------------------------------------------------------------------------------
#define KH_CMP_TILE 2048
__global__ void k_flag_tile_sums(const uint8_t* __restrict__ flags, uint64_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t wsum[4];
  const uint64_t base = (uint64_t)blockIdx.x * KH_CMP_TILE + (uint64_t)threadIdx.x * 8;
  uint32_t c = 0;
  if (base + 8 <= n) {
    const uint64_t fw = *reinterpret_cast<const uint64_t*>(flags + base);
#pragma unroll
    for (int j = 0; j < 8; ++j) c += ((fw >> (8 * j)) & 0xFFu) ? 1u : 0u;
  } else {
    for (int j = 0; j < 8; ++j) if (base + j < n && flags[base + j]) ++c;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-workgroup exclusive scan of u32 counts into u64 offsets (out has n+1 entries; out[n] = total).
// A sweep covers 16 chunks of 2048 counts: lane t owns counts [4t, 4t+4) of every chunk (one 16-byte load each, all 16 in
// flight at once), the 16 chunk scans run in registers with ONE barrier between them and the write-out.  (The earlier
// version swept 16384 counts at a time with three barriers and a dependent load per sweep: 77 us for the 65537 counts of
// a 10^8-key partition.)
#define KH_SCAN_THREADS 512
#define KH_SCAN_CHUNKS 16
#define KH_SCAN_CHUNK (KH_SCAN_THREADS * 4)
__global__ __launch_bounds__(KH_SCAN_THREADS) void k_scan_u32_to_u64(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ out) {
  __shared__ uint64_t wsum[KH_SCAN_CHUNKS][KH_SCAN_THREADS / 64];
  __shared__ uint64_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t base = 0; base < n; base += (uint64_t)KH_SCAN_CHUNKS * KH_SCAN_CHUNK) {
    uint32_t v[KH_SCAN_CHUNKS][4];
#pragma unroll
    for (int c = 0; c < KH_SCAN_CHUNKS; ++c) {
      const uint64_t i0 = base + (uint64_t)c * KH_SCAN_CHUNK + 4 * tid;
      if (aligned && i0 + 4 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(in + i0);
        v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][j] = (i0 + j < n) ? in[i0 + j] : 0u;
      }
    }
    uint64_t incl[KH_SCAN_CHUNKS];
#pragma unroll
    for (int c = 0; c < KH_SCAN_CHUNKS; ++c) {
      uint64_t x = (uint64_t)v[c][0] + v[c][1] + v[c][2] + v[c][3];
      for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(x, off, 64);
        if (lane >= (uint32_t)off) x += o;
      }
      incl[c] = x;
      if (lane == 63) wsum[c][wid] = x;
    }
    __syncthreads();
    uint64_t running = carry_s;
#pragma unroll
    for (int c = 0; c < KH_SCAN_CHUNKS; ++c) {
      uint64_t wpre = 0, tot = 0;
#pragma unroll
      for (uint32_t w = 0; w < KH_SCAN_THREADS / 64; ++w) { const uint64_t x = wsum[c][w]; if (w < wid) wpre += x; tot += x; }
      uint64_t ex = running + wpre + incl[c] - ((uint64_t)v[c][0] + v[c][1] + v[c][2] + v[c][3]);
      const uint64_t i0 = base + (uint64_t)c * KH_SCAN_CHUNK + 4 * tid;
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (i0 + j < n) out[i0 + j] = ex; ex += v[c][j]; }
      running += tot;
    }
    __syncthreads();
    if (tid == 0) carry_s = running;
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

__global__ __launch_bounds__(256) void k_compact_hits(const uint8_t* __restrict__ flags, const uint64_t* __restrict__ q, const uint32_t* __restrict__ vals,
                               uint64_t n, const uint64_t* __restrict__ tile_off,
                               uint64_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals, uint8_t* __restrict__ out_pairs16) {
  // one 256-thread workgroup per KH_CMP_TILE queries; each lane owns 8 consecutive queries (one 8-byte flag word),
  // so the hit order inside the tile is the query order; hits are staged in LDS and streamed out coalesced.
  __shared__ uint64_t lk[KH_CMP_TILE];
  __shared__ uint32_t lv[KH_CMP_TILE];
  __shared__ uint32_t wtot[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t tbase = (uint64_t)blockIdx.x * KH_CMP_TILE;
  const uint64_t base = tbase + (uint64_t)tid * 8;
  uint64_t fw = 0;
  if (base + 8 <= n) fw = *reinterpret_cast<const uint64_t*>(flags + base);     // flags buffer is 256-byte aligned, base % 8 == 0
  else for (int j = 0; j < 8; ++j) if (base + j < n && flags[base + j]) fw |= 1ull << (8 * j);
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) c += ((fw >> (8 * j)) & 0xFFu) ? 1u : 0u;
  uint32_t incl = c;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= (uint32_t)off) incl += o;
  }
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  uint32_t wpre = 0, total = 0;
  for (uint32_t w = 0; w < 4; ++w) { if (w < wid) wpre += wtot[w]; total += wtot[w]; }
  uint32_t x = wpre + incl - c;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if ((fw >> (8 * j)) & 0xFFu) { lk[x] = q[base + j]; lv[x] = vals ? vals[base + j] : 0u; ++x; }
  }
  __syncthreads();
  const uint64_t obase = tile_off[blockIdx.x];
  for (uint32_t s = tid; s < total; s += 256) {
    if (out_pairs16) {
      uint4 w;
      w.x = (uint32_t)lk[s]; w.y = (uint32_t)(lk[s] >> 32); w.z = lv[s]; w.w = 0;
      *reinterpret_cast<uint4*>(out_pairs16 + (obase + s) * 16) = w;
    } else {
      out_keys[obase + s] = lk[s];
      if (out_vals) out_vals[obase + s] = lv[s];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// radix partition of a batch by (bit-reversed) chunk id.  Records travel as SoA (key, val, idx).
// ---------------------------------------------------------------------------------------------
struct KhTile { uint64_t beg; uint32_t len; uint32_t seg; };
// 12-byte record (key, value) without the stream position: a batch whose sample showed no duplicate key, going into an empty
// table, needs no first-occurrence order (every key is its own first occurrence; any duplicate the build meets sends the
// batch back through the 16-byte path).  One dwordx3 access each: a quarter less partition traffic.
struct KhRec12 { uint32_t klo, khi, val; };

struct KhPartParams {
  const char* kbase; uint32_t kstride;     // input keys (stride 8 = SoA, 16 = pair array)
  const char* vbase; uint32_t vstride;     // first pass: input values (null: every record carries vconst); iv = position<<32 | value
  uint32_t vconst;
  uint64_t idx_base;                       // first pass: stream position of record 0 (streamed inserts: records fed before)
  const ulonglong2* rec_in;                // later passes: the input records (key, iv); kbase/vbase unused
  uint64_t n;                              // number of input records
  const KhTile* tiles;                     // null: arithmetic tiles of KH_PART_TILE over [0,n), seg 0
  const uint32_t* ntiles_dev;              // with tiles: actual tile count
  uint32_t ntiles;                         // without tiles: tile count
  KhSeed seed;
  uint32_t PB;                             // total partition bits
  uint32_t shift;                          // digit = (q >> shift) & (nb-1)
  uint32_t nb;                             // bins in this pass (power of two, <= 2048)
  uint32_t* counts;                        // [nseg*nb] histogram (hist kernel)
  unsigned long long* cursor;              // [nseg*nb] running output offsets (scatter kernel)
  ulonglong2* orec;                        // output records: 16-byte (key, iv) pairs, one dwordx4 access each
  // histogram-free mode (hashed keys spread evenly): output partition j owns the fixed slot [j * slot, (j + 1) * slot) of
  // orec and its cursor starts at j * slot; no histogram pass, no offset scan.  A partition that outgrows its slot raises
  // *overflow (its surplus records land in a dump area behind the output) and the host repeats the batch with exact offsets.  0 = exact offsets.
  uint64_t slot;
  uint64_t dump;                           // first record of the dump area in orec (KH_PART_TILE records)
  uint32_t* overflow;
  int xcd_swizzle;                         // eight consecutive tiles per XCD (adjacent output runs meet in one L2; speed only)
  // second pass over the fixed slots of a histogram-free first pass, tiles by arithmetic (tiles == null): input segment s is the slot
  // [s * slot_in, cur_in[s]) of rec_in, cut into tps tiles of KH_PART_TILE records -- tile t = (segment t / tps, piece t % tps).  A
  // workgroup then knows where its records lie without reading a tile list: it requests them together with the segment's cursor
  // (ONE HBM round trip before the first record arrives instead of three: tile count, tile, records) and drops what lies behind the fill.
  uint64_t slot_in; const unsigned long long* cur_in; uint32_t tps;
  // (k_part_scatter<HASH, true>: rec_in / orec hold KhRec12 records instead -- histogram-free mode only)
};

// partition id of a hash: chunk id at the partitioning capacity, bit-reversed so that the
// partitions belonging to one chunk of ANY smaller power-of-two capacity are contiguous.
__device__ __forceinline__ uint32_t kh_part_q(uint64_t h, uint32_t PB) {
  if (PB == 0) return 0;
  uint32_t p = (uint32_t)(h >> KH_LB) & ((1u << PB) - 1u);
  return __brev(p) >> (32 - PB);
}

__device__ __forceinline__ KhTile kh_get_tile(const KhPartParams& P, uint32_t t) {
  if (P.tiles) return P.tiles[t];
  KhTile d;
  d.beg = (uint64_t)t * KH_PART_TILE;
  uint64_t rem = P.n - d.beg;
  d.len = rem < KH_PART_TILE ? (uint32_t)rem : KH_PART_TILE;
  d.seg = 0;
  return d;
}

template <int HASH, bool REC8 = false>      // REC8: rec_in holds 8-byte keys (second level of a counting insert's exact partition)
__global__ __launch_bounds__(KH_PART_THREADS) void k_part_hist(KhPartParams P) {
  extern __shared__ __align__(16) uint32_t kh_dyn_smem[];
  uint32_t* hist = kh_dyn_smem;
  const uint32_t tid = threadIdx.x, nb = P.nb;
  const uint32_t ntiles = P.tiles ? *P.ntiles_dev : P.ntiles;
  for (uint32_t i = tid; i < nb; i += KH_PART_THREADS) hist[i] = 0;
  __syncthreads();
  const uint32_t tpb = (ntiles + gridDim.x - 1) / gridDim.x;
  const uint32_t t0 = blockIdx.x * tpb;
  const uint32_t t1 = (t0 + tpb < ntiles) ? t0 + tpb : ntiles;
  uint32_t cur_seg = 0xFFFFFFFFu;
  for (uint32_t t = t0; t < t1; ++t) {
    KhTile d = kh_get_tile(P, t);
    if (d.seg != cur_seg) {
      if (cur_seg != 0xFFFFFFFFu) {
        __syncthreads();
        for (uint32_t i = tid; i < nb; i += KH_PART_THREADS) {
          uint32_t c = hist[i];
          if (c) { atomicAdd(&P.counts[(uint64_t)cur_seg * nb + i], c); hist[i] = 0; }
        }
        __syncthreads();
      }
      cur_seg = d.seg;
    }
    for (uint32_t i = tid; i < d.len; i += KH_PART_THREADS) {
      uint64_t key = P.rec_in ? (REC8 ? reinterpret_cast<const uint64_t*>(P.rec_in)[d.beg + i] : P.rec_in[d.beg + i].x)
                              : *reinterpret_cast<const uint64_t*>(P.kbase + (d.beg + i) * P.kstride);
      uint32_t q = kh_part_q(kh_hash64<HASH>(key, P.seed), P.PB);
      atomicAdd(&hist[(q >> P.shift) & (nb - 1)], 1u);
    }
  }
  __syncthreads();
  if (cur_seg != 0xFFFFFFFFu)
    for (uint32_t i = tid; i < nb; i += KH_PART_THREADS) {
      uint32_t c = hist[i];
      if (c) atomicAdd(&P.counts[(uint64_t)cur_seg * nb + i], c);
    }
}

// One tile per workgroup.  Records are first counted per digit in LDS (rank = returning LDS atomic), the
// tile's slice of every digit's output range is reserved with one global atomic per (tile, digit), then the
// records are staged in LDS in digit order and streamed out, so that consecutive lanes write consecutive
// addresses (a direct scatter costs 4.8x the algorithmic write traffic in partial-sector writes: profiles/
// round-1 PMC notes).
__device__ __forceinline__ void kh_rec_set(ulonglong2& r, uint64_t key, unsigned long long iv) { r = make_ulonglong2(key, iv); }
__device__ __forceinline__ void kh_rec_set(KhRec12& r, uint64_t key, unsigned long long iv) { r.klo = (uint32_t)key; r.khi = (uint32_t)(key >> 32); r.val = (uint32_t)iv; }
__device__ __forceinline__ void kh_rec_get(const ulonglong2& r, uint64_t& key, unsigned long long& iv) { key = r.x; iv = r.y; }
__device__ __forceinline__ void kh_rec_get(const KhRec12& r, uint64_t& key, unsigned long long& iv) { key = r.klo | ((uint64_t)r.khi << 32); iv = r.val; }
// record kinds: 0 = 16 bytes (key, position << 32 | value); 1 = 12 bytes (key, value): duplicate-free batch into an empty table;
// 2 = 8 bytes (key): a counting insert whose values are the constant 1 (Reducer = std::plus needs neither value nor position) --
// half the partition traffic of the k-mer counter's batches
struct KhRec8 { uint64_t key; };
__device__ __forceinline__ void kh_rec_set(KhRec8& r, uint64_t key, unsigned long long) { r.key = key; }
__device__ __forceinline__ void kh_rec_get(const KhRec8& r, uint64_t& key, unsigned long long& iv) { key = r.key; iv = 0; }
template <int RK> struct KhRecOf { typedef ulonglong2 type; };
template <> struct KhRecOf<1> { typedef KhRec12 type; };
template <> struct KhRecOf<2> { typedef KhRec8 type; };
#ifdef KH_PART_OCC3
// experiment: three workgroups per CU for the 12- and 8-byte record kinds (80 VGPRs, <= 53 KB of LDS: 3072 / 4096 records staged)
#define KH_PART_LB(RK) __launch_bounds__(KH_PART_THREADS, (RK) ? 6 : 4)      // (HIP: second argument = waves per SIMD)
#define KH_PART_STAGE_OF(RK) ((RK) == 1 ? 3072u : (uint32_t)KH_PART_STAGE)
#else
#define KH_PART_LB(RK) __launch_bounds__(KH_PART_THREADS)
#define KH_PART_STAGE_OF(RK) ((uint32_t)KH_PART_STAGE)
#endif
template <int HASH, int RK>
__global__ KH_PART_LB(RK) void k_part_scatter(KhPartParams P) {
  constexpr bool R12 = RK != 0;                   // (no stream positions in the records)
  constexpr uint32_t STAGE = KH_PART_STAGE_OF(RK);
  typedef typename KhRecOf<RK>::type Rec;
  extern __shared__ __align__(16) uint32_t kh_dyn_smem[];
  __shared__ Rec lrec[STAGE];
  __shared__ uint16_t ld[STAGE];
  __shared__ uint32_t wtot[KH_PART_THREADS / 64];
  const uint32_t nb = P.nb;
  uint32_t* hist = kh_dyn_smem;                 // [nb] counts, then reused as running fill
  uint32_t* loff = kh_dyn_smem + nb;            // [nb] exclusive offsets inside the tile
  unsigned long long* gbase = reinterpret_cast<unsigned long long*>(kh_dyn_smem + 2 * ((nb + 1u) & ~1u));   // [nb]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t ntiles = P.tiles ? *P.ntiles_dev : P.ntiles;
  if (blockIdx.x >= ntiles) return;
  uint32_t tile = blockIdx.x;
  if (P.xcd_swizzle && (tile | 63u) < ntiles) {      // (whole groups of 64 tiles only) eight CONSECUTIVE tiles per XCD: b, b + 8, ... share one
    const uint32_t x = tile & 7u, j = (tile >> 3) & 7u;
    tile = (tile & ~63u) + x * 8u + j;
  }
  KhTile d;
  uint32_t last;                                    // index the requests are clamped to
  unsigned long long fill_end = 0;
  if (P.slot_in) {                                  // (arithmetic tiles over fixed slots: the tile's length arrives with its records)
    const uint32_t seg = tile / P.tps, k = tile - seg * P.tps;
    fill_end = P.cur_in[seg];
    d.seg = seg; d.beg = (uint64_t)seg * P.slot_in + (uint64_t)k * KH_PART_TILE; d.len = 0;
    const uint64_t room = P.slot_in - (uint64_t)k * KH_PART_TILE;      // (> 0: tps = ceil(slot_in / KH_PART_TILE) at most)
    last = (room < KH_PART_TILE ? (uint32_t)room : KH_PART_TILE) - 1u;
  } else {
    d = kh_get_tile(P, tile);
    if (d.len == 0) return;
    last = d.len - 1u;                              // (a tile holds at least one record)
  }
  for (uint32_t i = tid; i < nb; i += KH_PART_THREADS) hist[i] = 0;
  uint64_t key[KH_PART_ITEMS];
  unsigned long long iv[KH_PART_ITEMS];
  uint32_t dr[KH_PART_ITEMS];                   // digit << 16 | rank inside the digit (rank < 8192)
  // every lane requests all its records before it looks at the first one (indices past the tile's end are clamped, not predicated:
  // a branch per item makes the compiler wait for each load in turn -- 16 dependent HBM round trips per tile)
  if (P.rec_in) {
    const Rec* src = reinterpret_cast<const Rec*>(P.rec_in) + d.beg;
    Rec rr[KH_PART_ITEMS];
#pragma unroll
    for (int j = 0; j < KH_PART_ITEMS; ++j) { const uint32_t i = tid + j * KH_PART_THREADS; rr[j] = src[i < last ? i : last]; }
#pragma unroll
    for (int j = 0; j < KH_PART_ITEMS; ++j) kh_rec_get(rr[j], key[j], iv[j]);
  } else {
    const char* kb = P.kbase + d.beg * P.kstride;
#pragma unroll
    for (int j = 0; j < KH_PART_ITEMS; ++j) {
      const uint32_t i = tid + j * KH_PART_THREADS, ic = i < last ? i : last;
      key[j] = *reinterpret_cast<const uint64_t*>(kb + (uint64_t)ic * P.kstride);
    }
    if (P.vbase) {
      const char* vb = P.vbase + d.beg * P.vstride;
#pragma unroll
      for (int j = 0; j < KH_PART_ITEMS; ++j) {
        const uint32_t i = tid + j * KH_PART_THREADS, ic = i < last ? i : last;
        iv[j] = *reinterpret_cast<const uint32_t*>(vb + (uint64_t)ic * P.vstride);
      }
    } else {
#pragma unroll
      for (int j = 0; j < KH_PART_ITEMS; ++j) iv[j] = P.vconst;
    }
    if (!R12) {
      const uint64_t pos0 = P.idx_base + d.beg;
#pragma unroll
      for (int j = 0; j < KH_PART_ITEMS; ++j) iv[j] |= (unsigned long long)(pos0 + tid + j * KH_PART_THREADS) << 32;
    }
  }
  if (P.slot_in) {
    const uint64_t sbeg = (uint64_t)d.seg * P.slot_in;
    uint64_t fill = fill_end - sbeg;
    if (fill > P.slot_in) fill = P.slot_in;                                // (the first pass sent what ran over to its dump area, flagged)
    const uint64_t off = d.beg - sbeg;
    // (the host cuts the slot into tiles up to mean + 9 sigma of a segment's fill, not up to the slot's end: a fill beyond that is flagged too)
    if (off == (uint64_t)(P.tps - 1u) * KH_PART_TILE && fill > (uint64_t)P.tps * KH_PART_TILE && tid == 0) *P.overflow = 1u;
    d.len = fill > off ? (fill - off < KH_PART_TILE ? (uint32_t)(fill - off) : KH_PART_TILE) : 0u;
    if (d.len == 0) return;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KH_PART_ITEMS; ++j) {
    const uint32_t i = tid + j * KH_PART_THREADS;
    const uint32_t q = kh_part_q(kh_hash64<HASH>(key[j], P.seed), P.PB);
    const uint32_t dg = (q >> P.shift) & (nb - 1);
    dr[j] = dg << 16;
    if (i < d.len) dr[j] |= atomicAdd(&hist[dg], 1u);
  }
  __syncthreads();
  // exclusive scan of the digit counts (each thread owns nb/512 consecutive bins) + global reservation
  const uint32_t per = (nb + KH_PART_THREADS - 1) / KH_PART_THREADS;
  uint32_t mine = 0;
  for (uint32_t k = 0; k < per; ++k) { uint32_t b = tid * per + k; if (b < nb) mine += hist[b]; }
  uint32_t incl = mine;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = __shfl_up(incl, off, 64);
    if (lane >= (uint32_t)off) incl += o;
  }
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  uint32_t run = incl - mine;
  for (uint32_t w = 0; w < wid; ++w) run += wtot[w];
  // the reservations are device-scope returning atomics (6 M of them per launch at 4096-record tiles cost 0.16 of the
  // 0.83 ms: hence 8192-record tiles); issue them, stage the first round in LDS meanwhile, collect the results afterwards
  unsigned long long gres[KH_PART_MAXPER];
#pragma unroll
  for (uint32_t k = 0; k < KH_PART_MAXPER; ++k) {
    gres[k] = 0;
    const uint32_t b = tid * per + k;
    if (k < per && b < nb) {
      const uint32_t c = hist[b];
      loff[b] = run;
      run += c;
      if (c) {
        unsigned long long g = atomicAdd(&P.cursor[(uint64_t)d.seg * nb + b], (unsigned long long)c);
        // histogram-free mode: a reservation that runs over the partition's slot is redirected, whole, to the dump area behind
        // the output (one tile's worth of records) and the batch is flagged for a second try with exact offsets
        if (P.slot && g + c > ((uint64_t)d.seg * nb + b + 1) * P.slot) { *P.overflow = 1u; g = P.dump + loff[b]; }
        gres[k] = g;
      }
    }
  }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < d.len; r0 += STAGE) {
    if (r0) __syncthreads();                    // the previous round has been streamed out
#pragma unroll
    for (int j = 0; j < KH_PART_ITEMS; ++j) {
      uint32_t i = tid + j * KH_PART_THREADS;
      if (i < d.len) {
        const uint32_t dg = dr[j] >> 16;
        const uint32_t s = loff[dg] + (dr[j] & 0xFFFFu) - r0;     // position in the tile's digit order, relative to this round
        if (s < STAGE) { kh_rec_set(lrec[s], key[j], iv[j]); ld[s] = (uint16_t)dg; }
      }
    }
    if (r0 == 0) {
#pragma unroll
      for (uint32_t k = 0; k < KH_PART_MAXPER; ++k) {
        const uint32_t b = tid * per + k;
        if (k < per && b < nb) gbase[b] = gres[k];
      }
    }
    __syncthreads();
    const uint32_t rl = d.len - r0 < STAGE ? d.len - r0 : STAGE;
    for (uint32_t s = tid; s < rl; s += KH_PART_THREADS) {
      const uint32_t dd = ld[s];
      const uint64_t pos = gbase[dd] + (r0 + s - loff[dd]);
      reinterpret_cast<Rec*>(P.orec)[pos] = lrec[s];
    }
  }
}

// Both partition levels from ONE sweep over the keys: a 65536-bin histogram of the full partition id lives in LDS as
// 16-bit fields (128 KB, one 1024-lane workgroup per CU).  A field that reaches 0x8000 hands 0x8000 counts to the
// global counter (exactly one lane observes the crossing), so no field can overflow whatever the key distribution.
// The second-level histogram pass over the 16-byte records (0.32 ms at 1e8 keys) is not needed any more.
#define KH_FULLHIST_THREADS 1024
// More than 2^16 partitions (tables beyond 2^27 buckets): the partition ids are cut into 2^slice_bits slices of 2^16; one
// launch per slice sweeps all keys and counts the ones whose id falls into its slice (two sweeps for 1.25e8 keys per GPU cost
// 0.6 ms; the two-level fallback histogram costs 1.9 ms there).
template <int HASH>
__global__ __launch_bounds__(KH_FULLHIST_THREADS) void k_part_hist_full(const char* __restrict__ kbase, uint32_t kstride, uint64_t n, KhSeed seed,
                                                                         uint32_t PB, uint32_t slice, uint32_t slice_bits,
                                                                         uint32_t* __restrict__ counts /* [2^PB], zeroed */) {
  extern __shared__ __align__(16) uint32_t kh_dyn_smem[];     // 2^(PB - slice_bits) / 2 words
  const uint32_t lb = PB - slice_bits;                        // bits of the id inside a slice
  const uint32_t words = lb ? (1u << (lb - 1)) : 1u;
  const uint32_t lmask = (1u << lb) - 1u;
  for (uint32_t i = threadIdx.x; i < words; i += KH_FULLHIST_THREADS) kh_dyn_smem[i] = 0;
  __syncthreads();
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * per;
  const uint64_t b1 = b0 + per < n ? b0 + per : n;
  // eight keys per lane and trip: their loads are in flight together (one dependent HBM round trip per trip, not per key)
  for (uint64_t i0 = b0 + threadIdx.x; i0 < b1; i0 += 8 * KH_FULLHIST_THREADS) {
    uint64_t key[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t i = i0 + (uint64_t)j * KH_FULLHIST_THREADS;
      key[j] = i < b1 ? *reinterpret_cast<const uint64_t*>(kbase + i * kstride) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (i0 + (uint64_t)j * KH_FULLHIST_THREADS >= b1) break;
      const uint32_t qg = kh_part_q(kh_hash64<HASH>(key[j], seed), PB);
      if (slice_bits && (qg >> lb) != slice) continue;
      const uint32_t q = qg & lmask;
      const uint32_t sh = 16u * (q & 1u);
      const uint32_t old = atomicAdd(&kh_dyn_smem[q >> 1], 1u << sh);
      if ((((old >> sh) & 0xFFFFu) + 1u) == 0x8000u) {
        atomicSub(&kh_dyn_smem[q >> 1], 0x8000u << sh);
        atomicAdd(&counts[qg], 0x8000u);
      }
    }
  }
  counts += (size_t)slice << lb;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < words; i += KH_FULLHIST_THREADS) {
    const uint32_t w = kh_dyn_smem[i];
    if (w & 0xFFFFu) atomicAdd(&counts[2 * i], w & 0xFFFFu);
    if (w >> 16) atomicAdd(&counts[2 * i + 1], w >> 16);
  }
}
// segoff[s] = part_off[s * nb2] (s <= nb1): the pass-1 bucket boundaries are every nb2-th partition boundary
__global__ void k_seg_offsets(const uint64_t* __restrict__ part_off, uint32_t nb1, uint32_t nb2, uint64_t* __restrict__ segoff,
                              unsigned long long* __restrict__ cur1) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s <= nb1) { const uint64_t v = part_off[(uint64_t)s * nb2]; segoff[s] = v; if (s < nb1) cur1[s] = v; }
}

// Small batches (<= 32768 pairs, <= 32 partitions): the whole partition in ONE launch.  Workgroup q sweeps the input (it stays in
// L2) and keeps the records of partition q in its own slot of n records (any skew fits); the cursor / start arrays are what the
// histogram-free layout uses (partition q = orec[q * n, cursor[q])).  Replaces histogram + scan + scatter (and their memsets) where
// the launches, not the bytes, are the cost: the in-place path for batches of middle size.
__device__ __forceinline__ uint32_t kh_wave_append(bool want, uint32_t* counter);      // (defined with the de-dup helpers below)
template <int HASH>
__global__ __launch_bounds__(512) void k_part_direct(const char* __restrict__ kbase, uint32_t kstride, const char* __restrict__ vbase, uint32_t vstride,
                                                     uint32_t vconst, uint64_t n, KhSeed seed, uint32_t PB, ulonglong2* __restrict__ orec,
                                                     unsigned long long* __restrict__ cursor, uint64_t* __restrict__ starts) {
  __shared__ uint32_t cnt;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (uint64_t i0 = 0; i0 < n; i0 += 512) {
    const uint64_t i = i0 + tid;
    bool take = false;
    uint64_t key = 0;
    if (i < n) {
      key = *reinterpret_cast<const uint64_t*>(kbase + i * kstride);
      take = kh_part_q(kh_hash64<HASH>(key, seed), PB) == q;
    }
    const uint32_t pos = kh_wave_append(take, &cnt);
    if (take) {
      const unsigned long long iv = ((unsigned long long)i << 32) | (vbase ? *reinterpret_cast<const uint32_t*>(vbase + i * vstride) : vconst);
      orec[(uint64_t)q * n + pos] = make_ulonglong2(key, iv);
    }
  }
  __syncthreads();
  if (tid == 0) { cursor[q] = (uint64_t)q * n + cnt; starts[q] = (uint64_t)q * n; if (q == 0) starts[gridDim.x] = (uint64_t)gridDim.x * n; }
}

// Is the batch heavy in duplicates?  (Only then are the partitions of hashed keys uneven enough to outgrow the fixed slots of
// the histogram-free partition.)  KH_SAMPLE_N keys at a regular stride go into an open-addressing set in global memory; a key met
// again counts as a duplicate.  benchmark_hashtables' input (x5.5 multiplicity, 1.8e7 distinct of 1e8) gives ~120 hits, distinct
// k-mers none; a few keys with 10^4 copies among 10^8 slip through -- the slot overflow flag catches those.
#define KH_SAMPLE_N 65536u
#define KH_SAMPLE_SET (1u << 18)
__global__ void k_sample_dups(const char* __restrict__ kbase, uint32_t kstride, uint64_t n, unsigned long long* __restrict__ set /* zeroed */,
                              uint32_t* __restrict__ dups) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KH_SAMPLE_N) return;
  const uint64_t pos = (uint64_t)i * (n / KH_SAMPLE_N);
  const unsigned long long key = *reinterpret_cast<const uint64_t*>(kbase + pos * kstride) ^ 0x8000000000000001ull;     // (0 marks an empty entry)
  const unsigned long long tag = key ? key : 1ull;
  uint32_t slot = (uint32_t)kh_fmix64(tag) & (KH_SAMPLE_SET - 1);
  for (;;) {
    const unsigned long long cur = atomicCAS(&set[slot], 0ull, tag);
    if (cur == 0ull) break;
    if (cur == tag) { atomicAdd(dups, 1u); break; }
    slot = (slot + 1) & (KH_SAMPLE_SET - 1);
  }
}

// histogram-free partition: cursor[j] = j * slot (and, if asked, the same values as partition start offsets)
// both levels of a histogram-free partition in one launch: level-1 cursors, level-2 cursors + slot starts, the overflow flag
__global__ void k_init_cursors2(unsigned long long* __restrict__ cur1, uint64_t n1, uint64_t slot1, unsigned long long* __restrict__ cur2,
                                uint64_t* __restrict__ starts, uint64_t n2, uint64_t slot2, uint32_t* __restrict__ ovf) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) cur1[i] = i * slot1;
  if (i < n2) cur2[i] = i * slot2;
  if (i <= n2) starts[i] = i * slot2;
  if (i == 0) *ovf = 0u;
}
__global__ void k_init_cursors(unsigned long long* __restrict__ cursor, uint64_t* __restrict__ starts, uint64_t n, uint64_t slot) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cursor[i] = i * slot;
  if (starts && i <= n) starts[i] = i * slot;
}

// tiles of KH_PART_TILE records that never straddle a segment (second partition pass).  slot != 0: segment s is the fixed
// slot [s * slot, cursor[s]) that the first, histogram-free pass filled (clamped to the slot).
__global__ void k_make_tiles(const uint64_t* __restrict__ segoff, uint32_t nseg, KhTile* __restrict__ tiles, uint32_t* __restrict__ ntiles_out,
                             const unsigned long long* __restrict__ cursor = nullptr, uint64_t slot = 0) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nseg; base += 1024) {
    uint32_t s = base + tid;
    uint64_t beg = 0, len = 0;
    if (s < nseg) {
      if (slot) { beg = (uint64_t)s * slot; len = cursor[s] - beg; if (len > slot) len = slot; }
      else { beg = segoff[s]; len = segoff[s + 1] - beg; }
    }
    uint32_t nt = (uint32_t)((len + KH_PART_TILE - 1) / KH_PART_TILE);
    uint32_t incl = nt;
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    uint32_t wpre = 0;
    for (uint32_t w = 0; w < wid; ++w) wpre += wtot[w];
    uint32_t carry = carry_s;
    uint32_t first = carry + wpre + incl - nt;
    for (uint32_t k = 0; k < nt; ++k) {
      KhTile d;
      d.beg = beg + (uint64_t)k * KH_PART_TILE;
      uint64_t rem = len - (uint64_t)k * KH_PART_TILE;
      d.len = rem < KH_PART_TILE ? (uint32_t)rem : KH_PART_TILE;
      d.seg = s;
      tiles[first + k] = d;
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + wpre + incl;
    __syncthreads();
  }
  if (tid == 0) *ntiles_out = carry_s;
}


// (max,+) composite: f(x) = max(A, x + n); combine(first, then) = then o first
struct KhMP { long long A; long long n; };
#define KH_MP_NEG (-(1ll << 60))
__device__ __forceinline__ KhMP kh_mp_combine(KhMP first, KhMP then) {
  KhMP r;
  long long a = first.A + then.n;
  r.A = then.A > a ? then.A : a;
  r.n = first.n + then.n;
  return r;
}
// exclusive scan of per-thread composites over the workgroup (thread order); also returns the total
__device__ __forceinline__ KhMP kh_block_scan_mp(KhMP v, KhMP* s_wtot, KhMP* total) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  KhMP incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    KhMP o;
    o.A = __shfl_up(incl.A, off, 64);
    o.n = __shfl_up(incl.n, off, 64);
    if (lane >= (uint32_t)off) incl = kh_mp_combine(o, incl);
  }
  KhMP excl;
  excl.A = __shfl_up(incl.A, 1, 64);
  excl.n = __shfl_up(incl.n, 1, 64);
  if (lane == 0) { excl.A = KH_MP_NEG; excl.n = 0; }
  __syncthreads();
  if (lane == 63) s_wtot[wid] = incl;
  __syncthreads();
  KhMP wpre; wpre.A = KH_MP_NEG; wpre.n = 0;
  for (uint32_t w = 0; w < wid; ++w) wpre = kh_mp_combine(wpre, s_wtot[w]);
  if (total) {
    KhMP t = wpre;
    for (uint32_t w = wid; w < nw; ++w) t = kh_mp_combine(t, s_wtot[w]);
    *total = t;
  }
  return kh_mp_combine(wpre, excl);
}


// 32-bit form for one chunk (positions < 2^13, counts < 2^12): half the cross-lane traffic of the 64-bit scan.  s_wtot must not be
// in use by an earlier call of the same workgroup (no barrier in front of its stores).
struct KhMP32 { int A; int n; };
#define KH_MP32_NEG (-(1 << 28))
__device__ __forceinline__ KhMP32 kh_mp_combine(KhMP32 first, KhMP32 then) {
  KhMP32 r;
  const int a = first.A + then.n;
  r.A = then.A > a ? then.A : a;
  r.n = first.n + then.n;
  return r;
}
// One step of a wavefront scan through the data-parallel-primitive path of the vector ALU (no LDS crossbar: a __shfl_up is a
// ds_bpermute_b32, 14 of them in a row made the scan 14 % of the bulk build's chunk time): every lane takes the composite of the lane
// the DPP control names -- row_shr:1/2/4/8 inside its row of 16, then lane 15 / lane 31 of the rows before it -- or the identity where
// there is none (old value, bound_ctrl off), and puts it in front of its own.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ KhMP32 kh_mp32_dpp_step(KhMP32 incl) {
  KhMP32 o;
  o.A = __builtin_amdgcn_update_dpp(KH_MP32_NEG, incl.A, CTRL, ROW_MASK, 0xF, false);
  o.n = __builtin_amdgcn_update_dpp(0, incl.n, CTRL, ROW_MASK, 0xF, false);
  return kh_mp_combine(o, incl);          // (identity in front: unchanged)
}
__device__ __forceinline__ KhMP32 kh_block_scan_mp32(KhMP32 v, KhMP32* s_wtot, KhMP32* total) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  KhMP32 incl = v;
  incl = kh_mp32_dpp_step<0x111, 0xF>(incl);      // row_shr:1
  incl = kh_mp32_dpp_step<0x112, 0xF>(incl);      // row_shr:2
  incl = kh_mp32_dpp_step<0x114, 0xF>(incl);      // row_shr:4
  incl = kh_mp32_dpp_step<0x118, 0xF>(incl);      // row_shr:8
  incl = kh_mp32_dpp_step<0x142, 0xA>(incl);      // row_bcast:15 into rows 1 and 3
  incl = kh_mp32_dpp_step<0x143, 0xC>(incl);      // row_bcast:31 into rows 2 and 3
  KhMP32 excl;
  excl.A = __builtin_amdgcn_update_dpp(KH_MP32_NEG, incl.A, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane 0: the identity)
  excl.n = __builtin_amdgcn_update_dpp(0, incl.n, 0x138, 0xF, 0xF, false);
  if (lane == 63) s_wtot[wid] = incl;
  __syncthreads();
  KhMP32 wpre; wpre.A = KH_MP32_NEG; wpre.n = 0;
  KhMP32 t = wpre;
  for (uint32_t w = 0; w < nw; ++w) {
    const KhMP32 x = s_wtot[w];
    if (w < wid) wpre = kh_mp_combine(wpre, x);
    t = kh_mp_combine(t, x);
  }
  *total = t;
  return kh_mp_combine(wpre, excl);
}

// fill an LDS array (16-byte aligned, a multiple of 16 bytes) with one 32-bit pattern, one ds_write_b128 per lane and trip
__device__ __forceinline__ void kh_lds_fill16(void* p, uint32_t bytes, uint32_t word) {
  uint4* q = reinterpret_cast<uint4*>(p);
  for (uint32_t i = threadIdx.x; i < bytes / 16u; i += blockDim.x) q[i] = make_uint4(word, word, word, word);
}
// wave-aggregated append: lanes with `want` get consecutive positions from *counter (one LDS atomic per wave)
__device__ __forceinline__ uint32_t kh_wave_append(bool want, uint32_t* counter) {
  const unsigned long long m = __ballot(want);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (m) {
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader, 64);
  }
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ uint32_t kh_wave_sum(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ uint32_t kh_wave_max(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}

#define KH_UPD_APPLIED 0x80000000u
enum { KH_DEDUP_FIRST = 0, KH_DEDUP_LAST = 1, KH_DEDUP_PLUS = 2, KH_DEDUP_ERASE = 3 };     // (ERASE: k_build_fused<.., 3>, no fold)
#define KH_DD_M 2048u            // records staged per de-dup round

// Folds the duplicates among the ns records staged in LDS (lk = keys, liv = idx<<32|val) into one representative per
// distinct key: the record that claims the key's entry of the 32-bit index set (one ds_cmpst_b32); every other
// occurrence merges its (idx|val) word into the representative's with one 64-bit LDS atomic -- min = first value wins
// (the index is the high word), max = last value wins, add = std::plus on the value.  Returns the bit mask of this
// lane's records (x = it * KH_CHUNK_THREADS + tid) that are representatives.  ns <= KH_DD_M < KH_HS entries: the probe
// always finds an empty entry.  Caller: set[] zeroed and records staged before (barrier), barrier after.
__device__ __forceinline__ uint32_t kh_dd_fold(unsigned long long* lk, unsigned long long* liv, uint32_t* set, uint32_t ns, int mode, uint32_t xk = 0) {
  const uint32_t tid = threadIdx.x;
  uint32_t rep_mask = 0;
  uint32_t rep_of[KH_DD_M / KH_CHUNK_THREADS];
  for (uint32_t x0 = 0, it = 0; x0 < ns; x0 += KH_CHUNK_THREADS, ++it) {
    const uint32_t x = x0 + tid;
    if (it < KH_DD_M / KH_CHUNK_THREADS) rep_of[it] = 0xFFFFFFFFu;
    if (x < ns) {
      const unsigned long long key = lk[x];
      // (key transform: a k-mer and its reverse complement are one key -- they must meet in the same set entry)
      uint32_t slot = (uint32_t)kh_fmix64(kh_xf(key, xk) + 0x9E3779B97F4A7C15ull) & (KH_HS - 1);
      for (;;) {
        // one LDS round trip per probe: the CAS itself tells whether the entry was free (most are: load <= 0.37)
        const uint32_t cur = atomicCAS(&set[slot], 0u, x + 1u);
        if (cur == 0) { rep_mask |= 1u << it; break; }
        const uint32_t rep = cur - 1u;
        if (kh_keq(lk[rep], key, xk)) {
          const unsigned long long iv = liv[x];
          if (mode == KH_DEDUP_FIRST) atomicMin(&liv[rep], iv);
          else if (mode == KH_DEDUP_LAST) atomicMax(&liv[rep], iv);
          else atomicAdd(&liv[rep], iv & 0xFFFFFFFFull);
          if (it < KH_DD_M / KH_CHUNK_THREADS) rep_of[it] = rep;
          break;
        }
        slot = (slot + 1) & (KH_HS - 1);
      }
    }
  }
  if (xk && mode == KH_DEDUP_FIRST) {
    // first value wins, and so does the first KEY: the element stored is the bit pattern of the earliest occurrence (the
    // reference inserts one by one and never replaces a key).  The winner of a group is the record whose (position | value)
    // word survived the min; where that is not the representative itself it hands its key over.  (lk[] of a representative is
    // only compared under the transform, which the hand-over does not change.)
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const uint32_t x = it * KH_CHUNK_THREADS + tid;
      if (x < ns && rep_of[it] != 0xFFFFFFFFu && liv[rep_of[it]] == liv[x]) lk[rep_of[it]] = lk[x];
    }
  }
  return rep_mask;
}

// The records of one partition may come from several feeds of a streamed insert (kh_insert_begin/feed/end): every feed
// was partitioned on its own, so partition q is the concatenation of q's slice of every source, in feed order.
#define KH_MAX_SRC 16
struct KhSrcSet {
  const ulonglong2* rec[KH_MAX_SRC];     // partitioned records of source s
  const uint64_t* off[KH_MAX_SRC];       // [nparts+1] partition offsets inside rec[s]
  uint64_t slot[KH_MAX_SRC];             // != 0: histogram-free source: partition q = rec[s][q * slot, cur[s][q]) (clamped to the slot)
  const unsigned long long* cur[KH_MAX_SRC];
  uint32_t n;                            // number of sources (>= 1)
  uint32_t rec12;                        // 1 (one histogram-free source only): rec[0] holds KhRec12 records (key, value; no stream position)
                                         // 2 (one source, counting insert, general path only): rec[0] holds 8-byte keys, every value is 1
  const uint64_t* merged_off;            // [nparts+1] sum over the sources of off[s][q]: where partition q's OUTPUT list starts
};
// per-workgroup view of partition q.  One source (every plain insert): the slice is addressed directly through a
// wave-uniform pointer, no LDS and no barrier.  Several sources: s_ptr[s] = rec[s] + (first record of the slice) - (records
// of the partition before it), s_cum[s] = records before it, so that record i of the partition is s_ptr[s][i] for the s
// with s_cum[s] <= i < s_cum[s+1]; the pointer table lives in LDS (a per-lane index into the kernel arguments would cost
// a dependent global load per record).
struct KhSrcView {
  const ulonglong2* one;                 // != nullptr: single source, slice start
  const KhRec12* one12;                  // != nullptr: single source of 12-byte records, slice start
  const uint64_t* one8;                  // != nullptr: single source of 8-byte keys (value 1 each), slice start
  uint32_t m;                            // records of the partition
  uint32_t n;                            // sources
};
template <bool ALLOW12 = false>
__device__ __forceinline__ KhSrcView kh_src_setup(const KhSrcSet& S, uint32_t q, const ulonglong2** s_ptr, uint32_t* s_cum) {
  KhSrcView V;
  V.n = S.n;
  V.one12 = nullptr; V.one8 = nullptr;
  if (S.n == 1) {
    if (S.slot[0]) {
      const uint64_t b = (uint64_t)q * S.slot[0];
      const uint64_t c = S.cur[0][q] - b;
      V.one = S.rec[0] + b;
      if (ALLOW12 && S.rec12) { V.one = nullptr; V.one12 = reinterpret_cast<const KhRec12*>(S.rec[0]) + b; }
      V.m = (uint32_t)(c < S.slot[0] ? c : S.slot[0]);
      return V;
    }
    const uint64_t b = S.off[0][q];
    V.one = S.rec[0] + b;
    V.m = (uint32_t)(S.off[0][q + 1] - b);
    return V;
  }
  if (threadIdx.x < 64) {
    const uint32_t s = threadIdx.x;
    uint64_t b = 0; uint32_t cnt = 0;
    if (s < S.n) { b = S.off[s][q]; cnt = (uint32_t)(S.off[s][q + 1] - b); }
    uint32_t inc = cnt;
    for (int o = 1; o < KH_MAX_SRC; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)s >= o) inc += t; }
    if (s < S.n) { s_cum[s] = inc - cnt; s_ptr[s] = S.rec[s] + b - (inc - cnt); }
    if (s == S.n - 1) s_cum[S.n] = inc;
  }
  __syncthreads();
  V.one = nullptr;
  V.m = s_cum[S.n];
  return V;
}
// one source of 8-byte keys (counting insert, k_dedup only)
__device__ __forceinline__ KhSrcView kh_src_setup8(const KhSrcSet& S, uint32_t q) {
  KhSrcView V;
  V.n = 1; V.one = nullptr; V.one12 = nullptr;
  if (S.slot[0]) {
    const uint64_t b = (uint64_t)q * S.slot[0], c = S.cur[0][q] - b;
    V.one8 = reinterpret_cast<const uint64_t*>(S.rec[0]) + b;
    V.m = (uint32_t)(c < S.slot[0] ? c : S.slot[0]);
  } else {
    const uint64_t b = S.off[0][q];
    V.one8 = reinterpret_cast<const uint64_t*>(S.rec[0]) + b;
    V.m = (uint32_t)(S.off[0][q + 1] - b);
  }
  return V;
}
// several sources: where record i of the partition lives (the loads themselves are issued by the caller, all of a lane's together)
__device__ __forceinline__ const ulonglong2* kh_src_addr(const KhSrcView& V, const ulonglong2* const* s_ptr, const uint32_t* s_cum, uint32_t i) {
  uint32_t s = 0;
  while (s + 1 < V.n && i >= s_cum[s + 1]) ++s;
  return s_ptr[s] + i;
}
template <bool ALLOW12 = false>
__device__ __forceinline__ ulonglong2 kh_src_load(const KhSrcView& V, const ulonglong2* const* s_ptr, const uint32_t* s_cum, uint32_t i) {
  if (V.one) return V.one[i];
  if (ALLOW12 && V.one12) { const KhRec12 r = V.one12[i]; return make_ulonglong2(r.klo | ((uint64_t)r.khi << 32), (unsigned long long)r.val); }
  uint32_t s = 0;
  while (s + 1 < V.n && i >= s_cum[s + 1]) ++s;
  return s_ptr[s][i];
}

// ---------------------------------------------------------------------------------------------
// K1: per-partition first-wins de-duplication in LDS + membership test against the current table.
// Emits the batch's DISTINCT NEW keys (with the value of their first occurrence).
// ---------------------------------------------------------------------------------------------
struct KhDedupParams {
  KhSrcSet src;                                                  // partitioned records (key, idx<<32|val) of every feed
  uint64_t* nk; uint32_t* nv;                                    // outputs, written at src.merged_off[q] + j
  uint32_t* cnt_new;                                             // [nparts]
  // KH_DEDUP_PLUS into a non-empty table: the keys the table already holds are NOT increased by this kernel (the host may still
  // have to discard the attempt: a histogram-free partition that overflowed, a failing re-layout).  Their (slot index, sum) pairs
  // are listed from the END of the partition's output region downwards (nk[end - 1 - j] = slot, nv[end - 1 - j] = sum; new keys
  // + existing keys <= records of the partition, so the two lists never meet); k_apply_plus adds them once the attempt stands
  uint32_t* cnt_upd;                                             // [nparts]; bit 31 (KH_UPD_APPLIED): this partition's sums are in the table already
  // != 0: nothing can discard this attempt before the table is re-laid out (exact partition offsets, no repeatable streamed insert):
  // a partition that needs ONE class (nearly all do) adds its sums right here, as part of its membership probes, and sets
  // KH_UPD_APPLIED; the list is written all the same -- it is what takes the sums back if the re-layout fails
  int plus_immediate;
  unsigned long long* max_idx_plus1;                             // max (first-occurrence index + 1) over new keys
  KhSlots T; KhSeed seed;
  // speculative fusion of the chunk-count step (empty table, one partition == one chunk of capacity count_cap):
  // home-bucket counts and the chunk's (max,+) summary are produced here and k_chunk_count is skipped when the
  // capacity decided after this kernel equals count_cap
  uint64_t count_cap; uint32_t PB; uint16_t* homecnt; long long* sumA; long long* sumN;
  int table_empty;                                               // size() == 0: skip the membership probes
  // > 1: the xcd_group consecutive partitions q that probe the SAME chunk of the (smaller) current table are given to workgroups b,
  // b + 8, b + 16, ... -- one XCD under the observed round-robin placement (speed only): their membership probes then share
  // the sectors of that chunk in one L2 instead of fetching them once per XCD
  uint32_t xcd_group;
  int mode;                                                      // KH_DEDUP_FIRST : insert (first value wins, emit keys the table lacks)
                                                                 // KH_DEDUP_LAST  : kh_update assign pass (last value wins, written in place)
                                                                 // KH_DEDUP_PLUS  : reducer std::plus: values of equal keys are summed; keys the
                                                                 //                  table holds are increased in place, the others are emitted
  uint32_t* flags;
};

// LDS budget 52 KB (3 workgroups per CU): the partition's records are staged in LDS (16 B each) and the hash set
// holds 32-bit record indices, so a set entry is claimed with one 32-bit CAS and duplicates fold into the claimed
// record's (idx|val) word with one 64-bit min/max/add.
template <int KIND, int HASH, bool REC8 = false>      // REC8: the one source holds 8-byte keys, every value 1 (counting insert)
__global__ __launch_bounds__(KH_CHUNK_THREADS) void k_dedup(KhDedupParams P) {
  __shared__ unsigned long long lk[KH_DD_M];
  __shared__ unsigned long long liv[KH_DD_M];
  __shared__ uint32_t set[KH_HS];               // 0 = empty, else staged record index + 1 (the key's representative)
  __shared__ uint32_t n_staged, out_count, upd_count, overflow, max_idx;
  __shared__ uint32_t cnt16[KH_L / 2];          // fused chunk count: two 16-bit home counters per word
  __shared__ KhMP s_wtot[KH_CHUNK_THREADS / 64];
  __shared__ const ulonglong2* s_ptr[KH_MAX_SRC];
  __shared__ uint32_t s_cum[KH_MAX_SRC + 1];
  const uint32_t tid = threadIdx.x;
  uint32_t q = blockIdx.x;
  if (P.xcd_group > 1) {       // (host: power of two, gridDim.x a multiple of 8 * xcd_group)
    const uint32_t G = P.xcd_group, x = q & 7u, row = q >> 3;
    q = (((row / G) << 3) + x) * G + (row & (G - 1u));
  }
  // one source in fixed slots (histogram-free partition): the slot lies where it lies whatever its fill, so the first tile's records are
  // requested together with the slot's cursor (indices clamped to the slot; the fill masks them below) -- one HBM round trip in front of the
  // first fold instead of two
  ulonglong2 pre[KH_DD_M / KH_CHUNK_THREADS];
  const bool have_pre = P.src.n == 1 && P.src.slot[0] != 0 && (REC8 ? P.src.rec12 == 2 : P.src.rec12 == 0);
  if (have_pre) {
    const uint64_t b0 = (uint64_t)q * P.src.slot[0];
    const uint32_t lastc = (uint32_t)P.src.slot[0] - 1u;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const uint32_t i = it * KH_CHUNK_THREADS + tid;
      if (REC8) pre[it] = make_ulonglong2(reinterpret_cast<const uint64_t*>(P.src.rec[0])[b0 + (i < lastc ? i : lastc)], 1ull);
      else pre[it] = P.src.rec[0][b0 + (i < lastc ? i : lastc)];
    }
  }
  const KhSrcView V = REC8 ? kh_src_setup8(P.src, q) : kh_src_setup(P.src, q, s_ptr, s_cum);
  const uint32_t m = V.m;
  const uint64_t beg = P.src.merged_off[q];            // output list of this partition
  const bool plus_live = P.mode == KH_DEDUP_PLUS && !P.table_empty;
  const uint64_t end = plus_live ? P.src.merged_off[q + 1] : 0;      // (the list of existing keys grows down from here)
  const uint64_t mask = P.T.cap - 1;
  const bool fuse = P.count_cap != 0;
  const uint64_t cmask = P.count_cap - 1;
  const uint32_t Lc = P.count_cap > KH_L ? KH_L : (uint32_t)P.count_cap;
  const uint32_t chunk = P.PB ? (__brev(q) >> (32 - P.PB)) : 0u;      // partition id = bit-reversed chunk id
  const uint64_t Sc = (uint64_t)chunk * Lc;
  if (m == 0 && !fuse) { if (tid == 0) { P.cnt_new[q] = 0; if (plus_live) P.cnt_upd[q] = 0; } return; }
  // The records are streamed through the staging area in tiles: the distinct keys found so far stay at its front
  // (D of them), the rest is refilled from the stream, folded, and the representatives are compacted to the front
  // again.  A partition of ANY size and multiplicity (1e7 copies of one k-mer are one key of one partition) therefore
  // needs one pass over its records as long as its DISTINCT keys fit (<= KH_DD_M - 512); only then are the keys split
  // into R classes by a hash independent of the table's and the stream is swept once per class.
  uint32_t R = 1;
  unsigned long long* tmp = reinterpret_cast<unsigned long long*>(set);      // compaction scratch: KH_HS * 4 B = KH_DD_M * 8 B
  static_assert(KH_HS * 4 == KH_DD_M * 8, "set[] doubles as the compaction scratch");
  bool done = false;
  while (!done) {
    if (tid == 0) { out_count = 0; upd_count = 0; overflow = 0; max_idx = 0; }
    if (fuse) for (uint32_t i = tid; i < KH_L / 2; i += KH_CHUNK_THREADS) cnt16[i] = 0;
    for (uint32_t r = 0; r < R; ++r) {
      uint32_t D = 0, pos = 0, ns = 0, rep_mask = 0;
      __syncthreads();
      for (;;) {
        // ---- refill: stream sub-tiles of KH_CHUNK_THREADS records while a whole sub-tile still fits
        if (R == 1) {
          // one class (nearly always): the next records go to the staging area as they come, up to four per
          // lane requested together (one HBM round trip for a partition of ~1500 records, not one per sub-tile of 512; clamped
          // indices, no branch per record)
          const uint32_t room = ((KH_DD_M - D) / KH_CHUNK_THREADS) * KH_CHUNK_THREADS;
          const uint32_t take_n = m - pos < room ? m - pos : room;
          if (take_n) {                                    // (an empty partition has nothing to read: m - 1 would wrap)
            ulonglong2 rr[KH_DD_M / KH_CHUNK_THREADS];
            const uint32_t last = m - 1u;
            if (have_pre && pos == 0) {       // (requested at the top, with the cursor)
#pragma unroll
              for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) rr[it] = pre[it];
            } else if (REC8) {       // counting insert: 8-byte keys, every value 1
#pragma unroll
              for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = pos + it * KH_CHUNK_THREADS + tid; rr[it] = make_ulonglong2(V.one8[i < last ? i : last], 1ull); }
            } else if (V.one) {
#pragma unroll
              for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = pos + it * KH_CHUNK_THREADS + tid; rr[it] = V.one[i < last ? i : last]; }
            } else {       // several feeds: find every record's source first, then request them together
              const ulonglong2* pp[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
              for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = pos + it * KH_CHUNK_THREADS + tid; pp[it] = kh_src_addr(V, s_ptr, s_cum, i < last ? i : last); }
#pragma unroll
              for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) rr[it] = *pp[it];
            }
#pragma unroll
            for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
              const uint32_t j = it * KH_CHUNK_THREADS + tid;
              if (j < take_n) { lk[D + j] = rr[it].x; liv[D + j] = rr[it].y; }
            }
          }
          pos += take_n;
          ns = D + take_n;
          __syncthreads();
        } else {
        if (tid == 0) n_staged = 0;
        __syncthreads();
        do {
          const uint32_t i = pos + tid;
          bool take = false;
          unsigned long long key = 0, iv = 0;
          if (i < m) {
            const ulonglong2 rr = REC8 ? make_ulonglong2(V.one8[i], 1ull) : kh_src_load(V, s_ptr, s_cum, i);
            key = rr.x; iv = rr.y;
            take = R == 1 || (uint32_t)((kh_fmix64(key + 0x9E3779B97F4A7C15ull) >> 32) % R) == r;
          }
          const uint32_t x = D + kh_wave_append(take, &n_staged);
          if (take) { lk[x] = key; liv[x] = iv; }
          pos += KH_CHUNK_THREADS;
          __syncthreads();
          ns = D + n_staged;
          __syncthreads();        // every lane has read the count before the next sub-tile's appends move it
        } while (pos < m && ns + KH_CHUNK_THREADS <= KH_DD_M);
        }
        // ---- fold duplicates into their representative; rep_mask: which of this lane's records are representatives
        for (uint32_t s = tid; s < KH_HS; s += KH_CHUNK_THREADS) set[s] = 0;
        __syncthreads();
        rep_mask = kh_dd_fold(lk, liv, set, ns, P.mode, P.seed.xk);
        __syncthreads();
        if (pos >= m) break;                       // stream exhausted: the representatives are this class's distinct keys
        // ---- compact the representatives to the front (set[] is free again: scratch), then continue with the stream
        if (tid == 0) n_staged = 0;
        __syncthreads();
        uint32_t nx[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
          const bool rep = (rep_mask >> it) & 1u;
          nx[it] = kh_wave_append(rep, &n_staged);
          if (rep) tmp[nx[it]] = lk[it * KH_CHUNK_THREADS + tid];
        }
        __syncthreads();
        D = n_staged;
        for (uint32_t j = tid; j < D; j += KH_CHUNK_THREADS) { const unsigned long long k = tmp[j]; lk[j] = k; }
        __syncthreads();
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it)
          if ((rep_mask >> it) & 1u) tmp[nx[it]] = liv[it * KH_CHUNK_THREADS + tid];
        __syncthreads();
        for (uint32_t j = tid; j < D; j += KH_CHUNK_THREADS) { const unsigned long long v = tmp[j]; liv[j] = v; }
        __syncthreads();
        if (D + KH_CHUNK_THREADS > KH_DD_M) { if (tid == 0) overflow = 1; __syncthreads(); break; }   // too many distinct keys for one class
      }
      if (overflow) break;
      // the distinct keys of this class: test membership in the current table, emit the new ones
      uint32_t my_max = 0;
      for (uint32_t x0 = 0, it = 0; x0 < ns; x0 += KH_CHUNK_THREADS, ++it) {
        const uint32_t x = x0 + tid;
        bool emit = false, upd = false;
        unsigned long long key = 0, iv = 0;
        uint64_t at = KH_NONE;
        if (x < ns && ((rep_mask >> it) & 1u)) {
          key = lk[x]; iv = liv[x];
          uint32_t cur_val = 0;
          if (!P.table_empty) {
            const uint64_t h = kh_hash64<HASH>(key, P.seed);
            at = kh_find_pos<KIND>(P.T.s, mask, h & mask, key, &cur_val, P.seed.xk);
          }
          if (P.mode == KH_DEDUP_LAST) { if (at != KH_NONE) P.T.s[at].val = (uint32_t)iv; }   // kh_update's assign pass: store the LAST value
          else if (P.mode == KH_DEDUP_PLUS && at != KH_NONE) {
            upd = true;                                                    // (slot, sum) listed for k_apply_plus ...
            if (P.plus_immediate && R == 1) P.T.s[at].val = cur_val + (uint32_t)iv;      // ... or added at once (one lane per distinct key: no race)
          }
          else emit = at == KH_NONE;
        }
        if (plus_live) {       // (wave-uniform)
          const uint32_t upos = kh_wave_append(upd, &upd_count);
          if (upd) { P.nk[end - 1 - upos] = at; P.nv[end - 1 - upos] = (uint32_t)iv; }
        }
        const uint32_t pos = kh_wave_append(emit, &out_count);
        if (emit) {
          P.nk[beg + pos] = key;
          P.nv[beg + pos] = (uint32_t)iv;
          const uint32_t ix = P.mode == KH_DEDUP_FIRST ? (uint32_t)(iv >> 32) + 1u : 0u;
          my_max = ix > my_max ? ix : my_max;
          if (fuse) {
            const uint32_t b = (uint32_t)((kh_hash64<HASH>(key, P.seed) & cmask) - Sc);
            atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1)));
          }
        }
      }
      my_max = kh_wave_max(my_max);
      if ((tid & 63) == 0 && my_max) atomicMax(&max_idx, my_max);
      __syncthreads();
    }
    if (overflow) { R *= 2; __syncthreads(); if (R > 2 * m + 2) { if (tid == 0) atomicOr(&P.flags[KH_FLAG_INTERNAL], 1u); break; } }
    else done = true;
  }
  __syncthreads();
  if (tid == 0) {
    P.cnt_new[q] = out_count;
    if (plus_live) P.cnt_upd[q] = upd_count | ((P.plus_immediate && R == 1) ? KH_UPD_APPLIED : 0u);
    if (out_count) atomicMax(P.max_idx_plus1, (unsigned long long)max_idx);
  }
  if (fuse) {      // what k_chunk_count would produce for this chunk
    __syncthreads();
    if (tid == 0 && out_count > 0xFFFFu) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);   // 16-bit fields could have wrapped
    KhMP v; v.A = KH_MP_NEG; v.n = 0;
#pragma unroll
    for (uint32_t j = 0; j < KH_L / KH_CHUNK_THREADS; ++j) {
      const uint32_t b = tid * (KH_L / KH_CHUNK_THREADS) + j;
      if (b < Lc) {
        const uint32_t cb = (cnt16[b >> 1] >> (16 * (b & 1))) & 0xFFFFu;
        P.homecnt[Sc + b] = (uint16_t)cb;
        KhMP h; h.A = (long long)b + cb; h.n = cb;
        v = kh_mp_combine(v, h);
      }
    }
    KhMP total;
    kh_block_scan_mp(v, s_wtot, &total);
    if (tid == 0) { P.sumA[chunk] = (long long)Sc + total.A; P.sumN[chunk] = total.n; }
  }
}

// the deferred half of a KH_DEDUP_PLUS pass over a non-empty table: every key the table already held gets the sum k_dedup listed for
// it (sign = +1), or loses it again (sign = -1: the re-layout that followed failed and the table must read as before).  Every slot
// appears in at most one list entry (a key belongs to one partition and one class): no race.
// sign +1 skips the partitions whose sums k_dedup added itself (KH_UPD_APPLIED); sign -1 takes back every listed sum.
__global__ void k_apply_plus(KhSlot* __restrict__ slots, const uint64_t* __restrict__ merged_off, const uint32_t* __restrict__ cnt_upd,
                             const uint64_t* __restrict__ nk, const uint32_t* __restrict__ nv, uint32_t nparts, int sign) {
  for (uint32_t q = blockIdx.x; q < nparts; q += gridDim.x) {
    const uint64_t end = merged_off[q + 1];
    const uint32_t cw = cnt_upd[q];
    if (sign > 0 && (cw & KH_UPD_APPLIED)) continue;
    const uint32_t c = cw & ~KH_UPD_APPLIED;
    for (uint32_t j = threadIdx.x; j < c; j += blockDim.x) {
      const uint64_t at = nk[end - 1 - j];
      const uint32_t d = nv[end - 1 - j];
      slots[at].val += sign > 0 ? d : (0u - d);
    }
  }
}

// merged_off[q] = sum over the sources of off[s][q] (q <= nparts): start of partition q in the merged (output) order
__global__ void k_merge_offsets(KhSrcSet S, uint64_t nq_plus1, uint64_t* __restrict__ merged) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq_plus1) { uint64_t v = 0; for (uint32_t s = 0; s < S.n; ++s) v += S.off[s][q]; merged[q] = v; }
}

// gather the per-partition outputs of k_dedup into one contiguous list (partition order)
__global__ void k_gather_new(const uint64_t* __restrict__ part_off, const uint64_t* __restrict__ noff,
                             const uint64_t* __restrict__ nk, const uint32_t* __restrict__ nv,
                             uint64_t* __restrict__ ck, uint32_t* __restrict__ cv) {
  const uint32_t q = blockIdx.x;
  const uint64_t src = part_off[q], dst = noff[q];
  const uint32_t c = (uint32_t)(noff[q + 1] - dst);
  for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) { ck[dst + i] = nk[src + i]; cv[dst + i] = nv[src + i]; }
}

// ---------------------------------------------------------------------------------------------
// chunk rebuild: K2a count homes + chunk summary, K2b carry scan over chunks, K3 placement
// ---------------------------------------------------------------------------------------------
struct KhRebuildParams {
  KhSlots Old;                       // source table
  int drop_marked;                   // RH erase: elements whose info word carries KH_INFO_ERASE_MARK are left out
  KhSlots New;                       // destination table (info pre-set to "empty")
  const uint64_t* ck; const uint32_t* cv;   // new distinct elements, grouped by partition
  const uint64_t* noff;              // [nparts+1] start of every partition's list in ck/cv (null: no new elements)
  const uint32_t* ncnt;              // per-partition list length; null: lists are dense (length = noff[q+1]-noff[q])
  uint32_t PB;                       // partition bits the new elements were grouped with
  KhSeed seed;
  uint16_t* homecnt;                 // [New.cap] elements per home bucket
  long long* sumA; long long* sumN;  // per-chunk (max,+) summary, absolute positions
  const long long* xcarry;           // per-chunk carry-in (absolute first free position)
  uint32_t* flags;
};

__device__ __forceinline__ uint32_t kh_log2u(uint64_t x) { return 63u - (uint32_t)__clzll((long long)x); }

// Calls f(key, val, home_new) for every live element of the OLD table whose new home lies in new
// chunk c.  Elements with old home in old chunk o sit in [S_o, first empty slot at/after S_o + L).
template <int KIND, int HASH, typename F>
__device__ __forceinline__ void kh_for_each_old(const KhRebuildParams& P, uint32_t c, uint32_t* s_emin, F f) {
  const uint32_t tid = threadIdx.x;
  if (P.Old.cap == 0) return;            // the host passes cap 0 for an empty source table
  const uint64_t cap_o = P.Old.cap, mask_o = cap_o - 1, mask_n = P.New.cap - 1;
  const uint32_t nch_o = cap_o > KH_L ? (uint32_t)(cap_o >> KH_LB) : 1u;
  const uint32_t nch_n = P.New.cap > KH_L ? (uint32_t)(P.New.cap >> KH_LB) : 1u;
  const uint32_t Lo = cap_o > KH_L ? KH_L : (uint32_t)cap_o;
  const uint32_t spill_max = (cap_o - Lo) < KH_L ? (uint32_t)(cap_o - Lo) : KH_L;
  uint32_t o = nch_o >= nch_n ? c : (c & (nch_o - 1));
  const uint32_t ostep = nch_o >= nch_n ? nch_n : nch_o;   // second form: exactly one iteration
  for (; o < nch_o; o += ostep) {
    const uint64_t S = (uint64_t)o * Lo;
    // how far behind the chunk its elements can sit.  Robin Hood: the probe distance is at most 127.  Linear probing: up to
    // the first empty slot -- searched window by window, because a run of occupied slots can be longer than a chunk at high
    // load (load 0.9, 10^7 elements: runs of ~3000 slots) and an element may have been pushed to its very end.
    uint64_t e_total = 0;
    const uint64_t beyond = cap_o - Lo;
    if (KIND == KHK_RH) e_total = beyond < 128u ? beyond : 128u;
    else {
      for (uint64_t base = 0; base < beyond; base += KH_L) {
        const uint32_t window = beyond - base < KH_L ? (uint32_t)(beyond - base) : KH_L;
        __syncthreads();
        if (tid == 0) *s_emin = window;
        __syncthreads();
        for (uint32_t t = tid; t < window; t += KH_CHUNK_THREADS) {
          if (kh_is_empty<KIND>(P.Old.s[(S + Lo + base + t) & mask_o].info & 0xFFu)) { atomicMin(s_emin, t); break; }
        }
        __syncthreads();
        const uint32_t e = *s_emin;
        e_total = base + e;
        if (e < window) break;
      }
    }
    (void)spill_max;
    const uint64_t len = (uint64_t)Lo + e_total;
    for (uint64_t t = tid; t < len; t += KH_CHUNK_THREADS) {
      const uint4 w = kh_slot_ld(P.Old.s + ((S + t) & mask_o));
      if (!kh_is_occupied<KIND>(w.w & 0xFFu)) continue;
      if (P.drop_marked && (w.w & KH_INFO_ERASE_MARK)) continue;
      const uint64_t key = kh_slot_key(w);
      const uint64_t h = kh_hash64<HASH>(key, P.seed);
      if ((uint32_t)((h & mask_o) >> KH_LB) != o) continue;          // belongs to a neighbouring old chunk
      if ((uint32_t)((h & mask_n) >> KH_LB) != c) continue;          // goes to another new chunk (growing)
      f(key, w.z, h & mask_n);
    }
    if (nch_o < nch_n) break;
  }
  __syncthreads();
}

// Calls f(key, val, home_new) for every new element whose home lies in new chunk c.  The elements were grouped
// by the bit-reversed chunk id of the partitioning capacity, so a chunk of any smaller-or-equal capacity owns a
// contiguous run of partitions [q0, q0 + 2^(PB-k)).
template <int HASH, typename F>
__device__ __forceinline__ void kh_for_each_new(const KhRebuildParams& P, uint32_t c, F f) {
  if (!P.noff) return;
  const uint64_t mask_n = P.New.cap - 1;
  const uint32_t nch_n = P.New.cap > KH_L ? (uint32_t)(P.New.cap >> KH_LB) : 1u;
  const uint32_t k = kh_log2u(nch_n);
  const uint32_t span_bits = P.PB - k;                    // PB >= k by construction
  const uint32_t q0 = k ? ((__brev(c) >> (32 - k)) << span_bits) : 0u;
  const uint32_t q1 = q0 + (1u << span_bits);
  if (!P.ncnt) {                                           // dense: one contiguous range
    const uint64_t b = P.noff[q0], e = P.noff[q1];
    for (uint64_t i = b + threadIdx.x; i < e; i += KH_CHUNK_THREADS) {
      const uint64_t key = P.ck[i];
      f(key, P.cv[i], kh_hash64<HASH>(key, P.seed) & mask_n);
    }
  } else {                                                 // a few per-partition lists (span is small)
    for (uint32_t q = q0; q < q1; ++q) {
      const uint64_t b = P.noff[q];
      const uint32_t n = P.ncnt[q];
      for (uint32_t i = threadIdx.x; i < n; i += KH_CHUNK_THREADS) {
        const uint64_t key = P.ck[b + i];
        f(key, P.cv[b + i], kh_hash64<HASH>(key, P.seed) & mask_n);
      }
    }
  }
}

#define KH_HOMES_PER_THREAD (KH_L / KH_CHUNK_THREADS)

template <int KIND, int HASH>
__global__ __launch_bounds__(KH_CHUNK_THREADS) void k_chunk_count(KhRebuildParams P) {
  __shared__ uint32_t cnt[KH_L];
  __shared__ uint32_t s_emin;
  __shared__ KhMP s_wtot[KH_CHUNK_THREADS / 64];
  const uint32_t tid = threadIdx.x, c = blockIdx.x;
  const uint32_t Ln = P.New.cap > KH_L ? KH_L : (uint32_t)P.New.cap;
  const uint64_t Sc = (uint64_t)c * Ln;
  for (uint32_t i = tid; i < KH_L; i += KH_CHUNK_THREADS) cnt[i] = 0;
  __syncthreads();
  kh_for_each_old<KIND, HASH>(P, c, &s_emin, [&](uint64_t, uint32_t, uint64_t hn) { atomicAdd(&cnt[hn - Sc], 1u); });
  kh_for_each_new<HASH>(P, c, [&](uint64_t, uint32_t, uint64_t hn) { atomicAdd(&cnt[hn - Sc], 1u); });
  __syncthreads();
  // per-thread composite over its consecutive homes, relative to the chunk start
  KhMP v; v.A = KH_MP_NEG; v.n = 0;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    if (b < Ln) {
      uint32_t cb = cnt[b];
      if (cb > 0xFFFFu) { atomicOr(&P.flags[KH_FLAG_COUNT_OVERFLOW], 1u); cb = 0xFFFFu; }
      P.homecnt[Sc + b] = (uint16_t)cb;
      KhMP h; h.A = (long long)b + cb; h.n = cb;
      v = kh_mp_combine(v, h);
    }
  }
  KhMP total;
  kh_block_scan_mp(v, s_wtot, &total);
  if (tid == 0) { P.sumA[c] = (long long)Sc + total.A; P.sumN[c] = total.n; }
}

// K2b: one workgroup scans the chunk summaries.  x[c] = absolute first free position entering chunk c.
// The table is circular: the run-over of the last chunk enters chunk 0, so the prefix composites are
// applied to x0 = max(0, F_all(0) - cap) (a fixed point as long as one slot of the table stays free).
#define KH_CARRY_ITEMS 8
__global__ void k_chunk_carry(const long long* __restrict__ sumA, const long long* __restrict__ sumN, uint32_t nch, long long cap,
                              long long* __restrict__ xcarry, KhMP* __restrict__ prefix_tmp) {
  __shared__ KhMP s_wtot[16];
  __shared__ KhMP s_carry;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) { s_carry.A = KH_MP_NEG; s_carry.n = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < nch; base += 1024 * KH_CARRY_ITEMS) {
    const uint32_t c0 = base + tid * KH_CARRY_ITEMS;
    KhMP it[KH_CARRY_ITEMS];
    KhMP v; v.A = KH_MP_NEG; v.n = 0;
#pragma unroll
    for (int j = 0; j < KH_CARRY_ITEMS; ++j) {
      it[j].A = KH_MP_NEG; it[j].n = 0;
      if (c0 + j < nch) { it[j].A = sumA[c0 + j]; it[j].n = sumN[c0 + j]; }
      v = kh_mp_combine(v, it[j]);
    }
    KhMP total;
    KhMP excl = kh_block_scan_mp(v, s_wtot, &total);
    const KhMP carry = s_carry;
    KhMP run = kh_mp_combine(carry, excl);
#pragma unroll
    for (int j = 0; j < KH_CARRY_ITEMS; ++j) {
      if (c0 + j < nch) prefix_tmp[c0 + j] = run;
      run = kh_mp_combine(run, it[j]);
    }
    __syncthreads();
    if (tid == 0) s_carry = kh_mp_combine(carry, total);
    __syncthreads();
  }
  KhMP all = s_carry;
  long long end0 = all.A > all.n ? all.A : all.n;   // F_all(0)
  long long x0 = end0 - cap;
  if (x0 < 0) x0 = 0;
  for (uint32_t c = tid; c < nch; c += 1024) {
    KhMP p = prefix_tmp[c];
    long long x = x0 + p.n;
    xcarry[c] = p.A > x ? p.A : x;
  }
}

// K3.  The chunk's slice of the new table is assembled in LDS and streamed out: the workgroup owns the slots
// [S_c + carry_in, S_c + max(L, end)) -- its own L home buckets minus what the previous chunks ran over into, plus
// its own run-over -- so the slices of all workgroups tile the (circular) table exactly once and every info byte,
// occupied or empty, is written by exactly one workgroup with coalesced stores.
#define KH_SPILL 256     // run-over slots staged in LDS; anything further out (only LP clusters) is stored directly
template <int KIND, int HASH>
__global__ __launch_bounds__(KH_CHUNK_THREADS) void k_chunk_place(KhRebuildParams P) {
  __shared__ uint64_t skeys[KH_L + KH_SPILL];
  __shared__ uint32_t svals[KH_L + KH_SPILL];
  __shared__ uint32_t sinfo_w[(KH_L + KH_SPILL) / 4];
  __shared__ uint32_t fill[KH_L];
  __shared__ uint32_t start[KH_L];
  __shared__ uint32_t s_emin;
  __shared__ long long s_pend;
  __shared__ KhMP s_wtot[KH_CHUNK_THREADS / 64];
  uint8_t* sinfo = reinterpret_cast<uint8_t*>(sinfo_w);
  const uint32_t tid = threadIdx.x, c = blockIdx.x;
  const uint32_t Ln = P.New.cap > KH_L ? KH_L : (uint32_t)P.New.cap;
  const uint64_t Sc = (uint64_t)c * Ln, mask_n = P.New.cap - 1;
  const uint32_t empty4 = KIND == KHK_RH ? 0u : 0x40404040u;
  for (uint32_t i = tid; i < KH_L + KH_SPILL; i += KH_CHUNK_THREADS) { skeys[i] = 0; svals[i] = 0; }
  for (uint32_t i = tid; i < (KH_L + KH_SPILL) / 4; i += KH_CHUNK_THREADS) sinfo_w[i] = empty4;
  uint32_t cb[KH_HOMES_PER_THREAD];
  KhMP v; v.A = KH_MP_NEG; v.n = 0;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    cb[j] = b < Ln ? P.homecnt[Sc + b] : 0u;
    if (b < Ln) { KhMP h; h.A = (long long)b + cb[j]; h.n = cb[j]; v = kh_mp_combine(v, h); }
    fill[tid * KH_HOMES_PER_THREAD + j] = 0;
  }
  KhMP excl = kh_block_scan_mp(v, s_wtot, nullptr);
  const long long xr = P.xcarry[c] - (long long)Sc;          // carry-in relative to the chunk start (<= 0: none)
  long long p = excl.A > xr + excl.n ? excl.A : xr + excl.n;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    if (b < Ln) {
      long long st = p > (long long)b ? p : (long long)b;
      start[b] = (uint32_t)st;
      p = st + cb[j];
    }
  }
  if (tid == KH_CHUNK_THREADS - 1) s_pend = p;     // homes are owned in thread order: the last lane holds the end
  __syncthreads();
  auto place = [&](uint64_t key, uint32_t val, uint64_t hn) {
    uint32_t b = (uint32_t)(hn - Sc);
    uint32_t r = atomicAdd(&fill[b], 1u);
    uint32_t prel = start[b] + r;
    uint32_t dist = prel - b;
    uint8_t ib = 0x00;
    if (KIND == KHK_RH) {
      if (dist > 127u) { atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u); dist = 127u; }
      ib = (uint8_t)(0x80u | dist);
    }
    if (prel < KH_L + KH_SPILL) { skeys[prel] = key; svals[prel] = val; sinfo[prel] = ib; }
    else kh_slot_st(P.New.s + ((Sc + prel) & mask_n), key, val, ib);
  };
  kh_for_each_old<KIND, HASH>(P, c, &s_emin, place);
  kh_for_each_new<HASH>(P, c, place);
  __syncthreads();
  // stream the slice out
  long long pend = s_pend;
  if (pend < (long long)Ln) pend = Ln;
  // threads with b >= Ln own nothing; when Ln < KH_L the last lane may not be the owner of the last home
  const uint32_t lo = xr > 0 ? (uint32_t)xr : 0u;
  uint32_t hi = pend < (long long)(KH_L + KH_SPILL) ? (uint32_t)pend : (KH_L + KH_SPILL);
  for (uint32_t s0 = lo + tid; s0 < hi; s0 += KH_CHUNK_THREADS) {
    kh_slot_st(P.New.s + ((Sc + s0) & mask_n), skeys[s0], svals[s0], sinfo[s0]);
  }
}

// ---------------------------------------------------------------------------------------------
// Fused bulk build (empty table, one partition == one chunk of the predicted capacity): de-dup, home count, carry
// and placement in ONE kernel -- the distinct elements never leave LDS.  The only inter-workgroup dependency is the
// run-over of the previous chunk.  A chunk with n_c + KH_XB <= L elements publishes its run-over before it knows its
// own carry-in (a carry-in of at most KH_XB slots cannot reach its end), so the look-back is one chunk deep.  Chunk c
// is handled by workgroup blockIdx c: the dispatcher is observed to start workgroups in index order, so the predecessor
// is running or done and the poll lasts microseconds -- but nothing relies on it: the poll is BOUNDED, a timeout raises a
// flag and the host repeats the batch on the general path (a ticket counter would make the order a guarantee, but
// 65536 returning atomics on one word cost 2.3 ms here: measured 3.7 ms with ticket vs 1.4 ms without).  Chunk 0
// depends on the LAST chunk (circular table); it only publishes, parks its distinct keys in global memory and is
// placed by k_chunk_place after the kernel.  The published word is one naturally aligned 8-byte granule {valid, run-over, count} that carries its own data: it is
// written and polled with RELAXED agent-scope atomics (L1-bypassing `sc1` accesses; MI355X guide G16 'R2 granule').
// A release store here would write back the XCD's whole dirty L2 (65536 times, with 1.7 GB of table stores in
// flight): measured 14 ms instead of 1.6.  Anything outside these assumptions
// (a partition larger than the LDS staging area, a carry-in above KH_XB, a poll that times out) raises a flag and the
// host repeats the batch on the general path (k_dedup / k_chunk_count / k_chunk_carry / k_chunk_place).
// ---------------------------------------------------------------------------------------------
#define KH_XB 127u               // bound on a carry-in that may be ignored when publishing early
#define KH_FSPILL 128u           // run-over slots staged in LDS by the fused kernel
struct KhFusedParams {
  KhSrcSet src; uint32_t PB;
  KhSlots New; KhSeed seed; int mode;                  // KH_DEDUP_FIRST or KH_DEDUP_PLUS
  unsigned long long* pub;                               // [nch] zero-initialised: bit63 valid | run-over << 32 | count
  uint64_t* ck0; uint32_t* cv0; uint16_t* homecnt0;      // chunk 0 parked here: distinct keys/values (KH_DD_M), home counts (KH_L)
  uint32_t* maxidx;                                      // [nch] per chunk: max(first-occurrence index + 1) (k_fused_totals reduces)
  // early give-up on duplicate-heavy batches: after 64 chunks the distinct/record ratio predicts the final size; if even
  // 1.15x of it fits the next smaller capacity the speculation is hopeless, the remaining workgroups return at once
  uint64_t n_total, half_max_load;
  uint64_t base_size;                                    // SRC == 2: elements already in the table (all of them stay)
  unsigned long long* est;                               // [0] distinct so far << 32 | records so far (ONE word: the pair must be
                                                         //     read consistently), [1] abort
  uint32_t* flags;
  long long poll_limit;                                  // cycles the look-back may wait for its predecessor (~4 ms; a test hook shortens it)
  int nodup;                                             // SRC == 0: a sample of the batch found no duplicate -- skip the LDS hash-set fold and
                                                         //   CHECK instead that no home bucket received two equal keys (any duplicate: general path)
  KhRebuildParams R;                                     // SRC == 1 only: source table, erase mask, new distinct elements
};

// SRC == 1 (Robin Hood): the chunk's elements come from the CURRENT table (same capacity or half of the new one) plus the
// batch's new distinct keys, instead of from partition records -- the one-launch form of k_chunk_count + k_chunk_carry +
// k_chunk_place for erase, rehash/reserve and inserts into a non-empty table.  An element with home bucket in old chunk o
// sits in a slot of [S_o, S_o + L + 128) (probe distance <= 127) and carries its home in the info byte (slot - distance):
// no hash is evaluated unless the capacity doubles (one more hash bit is needed then).  Staged as
// lk[x] = key, liv[x] = (home - chunk start) << 32 | value; the elements are distinct, so there is nothing to fold.
// Returns the number of staged elements (may exceed the staging area: the caller gives up then).
// MERGE (SRC == 2): the staged table elements will be folded together with the batch's records of this chunk, so they
// carry iv = 0 << 32 | value (stream position field 0: an element of the table beats every record of the batch, whose
// position fields are shifted up by one) and the new distinct lists are not read.
// Linear probing (KIND == KHK_LP): the info byte carries no distance, so the home is hash & mask, and an element of old
// chunk o can sit anywhere up to the first EMPTY slot behind the chunk (searched window by window through *scratch);
// tombstones are dropped, as a rehash of the reference drops them.
template <int KIND, int HASH, bool MERGE>
__device__ __forceinline__ uint32_t kh_stage_from_table(const KhRebuildParams& R, uint32_t c, uint64_t Sc, unsigned long long* lk,
                                                         unsigned long long* liv, uint32_t* n_staged, uint32_t* scratch) {
  const uint32_t tid = threadIdx.x;
  const uint64_t cap_o = R.Old.cap, mask_o = cap_o - 1, mask_n = R.New.cap - 1;
  if (tid == 0) *n_staged = 0;
  __syncthreads();
  if (cap_o) {
    const uint32_t nch_o = (uint32_t)(cap_o >> KH_LB);       // host: cap_o >= 2 * KH_L and New.cap in {cap_o, 2 * cap_o}
    const uint32_t o = c & (nch_o - 1);
    const bool same = R.New.cap == cap_o;
    const uint64_t S = (uint64_t)o * KH_L;
    uint64_t len = KH_L + 128u;                               // Robin Hood: probe distance <= 127
    if (KIND == KHK_LP) {
      const uint64_t beyond = cap_o - KH_L;
      uint64_t e_total = beyond;
      for (uint64_t base = 0; base < beyond; base += KH_L) {
        const uint32_t window = beyond - base < KH_L ? (uint32_t)(beyond - base) : KH_L;
        __syncthreads();
        if (tid == 0) *scratch = window;
        __syncthreads();
        for (uint32_t t = tid; t < window; t += KH_CHUNK_THREADS) {
          if ((R.Old.s[(S + KH_L + base + t) & mask_o].info & 0xFFu) == 0x40u) { atomicMin(scratch, t); break; }
        }
        __syncthreads();
        const uint32_t e = *scratch;
        e_total = base + e;
        if (e < window) break;
      }
      __syncthreads();
      if (tid == 0) *scratch = 0;
      len = KH_L + e_total;
    }
    if (KIND == KHK_RH) {
      // the 2176 candidate slots are 4.25 per lane: all of them are requested at once, one 16-byte load each (key, value and
      // info byte arrive together: one HBM round trip per chunk)
      constexpr uint32_t NS = (KH_L + 128u + KH_CHUNK_THREADS - 1) / KH_CHUNK_THREADS;
      uint4 w[NS];
#pragma unroll
      for (uint32_t it = 0; it < NS; ++it) {
        const uint32_t t = it * KH_CHUNK_THREADS + tid;
        w[it] = kh_slot_ld(R.Old.s + ((S + (t < KH_L + 128u ? t : KH_L + 127u)) & mask_o));      // (clamped, not predicated: loads stay in flight together)
      }
#pragma unroll
      for (uint32_t it = 0; it < NS; ++it) {
        const uint64_t sl = (S + it * KH_CHUNK_THREADS + tid) & mask_o;
        const uint32_t inf = it * KH_CHUNK_THREADS + tid < KH_L + 128u ? (w[it].w & 0xFFu) : 0u;
        bool take = false;
        uint32_t hrel = 0;
        const uint64_t key = kh_slot_key(w[it]);
        if (inf >= 0x80u && !(R.drop_marked && (w[it].w & KH_INFO_ERASE_MARK))) {
          const uint64_t home_o = (sl - (inf & 0x7Fu)) & mask_o;
          if ((uint32_t)(home_o >> KH_LB) == o) {
            if (same) { take = true; hrel = (uint32_t)(home_o - S); }
            else {
              const uint64_t hn = kh_hash64<HASH>(key, R.seed) & mask_n;
              take = (uint32_t)(hn >> KH_LB) == c;
              hrel = (uint32_t)(hn - Sc);
            }
          }
        }
        const uint32_t x = kh_wave_append(take, n_staged);
        if (take && x < KH_DD_M) { lk[x] = key; liv[x] = MERGE ? (unsigned long long)w[it].z : (((unsigned long long)hrel << 32) | w[it].z); }
      }
    } else
    for (uint64_t t0 = 0; t0 < len; t0 += KH_CHUNK_THREADS) {
      const uint64_t t = t0 + tid;
      bool take = false;
      uint64_t key = 0; uint32_t val = 0, hrel = 0;
      if (t < len) {
        const uint4 w = kh_slot_ld(R.Old.s + ((S + t) & mask_o));
        if ((w.w & 0xFFu) < 0x40u) {
          key = kh_slot_key(w);
          const uint64_t h = kh_hash64<HASH>(key, R.seed);
          if ((uint32_t)((h & mask_o) >> KH_LB) == o && (uint32_t)((h & mask_n) >> KH_LB) == c) {
            take = true; val = w.z; hrel = (uint32_t)((h & mask_n) - Sc);
          }
        }
      }
      const uint32_t x = kh_wave_append(take, n_staged);
      if (take && x < KH_DD_M) { lk[x] = key; liv[x] = MERGE ? (unsigned long long)val : (((unsigned long long)hrel << 32) | val); }
    }
  }
  if (!MERGE)
    kh_for_each_new<HASH>(R, c, [&](uint64_t key, uint32_t val, uint64_t hn) {
      const uint32_t x = kh_wave_append(true, n_staged);
      if (x < KH_DD_M) { lk[x] = key; liv[x] = ((unsigned long long)(uint32_t)(hn - Sc) << 32) | val; }
    });
  __syncthreads();
  return *n_staged;
}

#ifdef KH_TRACE
__device__ unsigned long long kh_trace[512 * 12];
#ifndef KH_TRACE_SRC
#define KH_TRACE_SRC 0          // which source mode of k_build_fused the stamps follow
#endif
#define KH_STAMP(i) do { if (SRC == KH_TRACE_SRC && threadIdx.x == 0 && blockIdx.x >= 20000 && blockIdx.x < 20512) kh_trace[(blockIdx.x - 20000) * 12 + (i)] = clock64(); } while (0)
#else
#define KH_STAMP(i)
#endif
template <int KIND, int HASH, int SRC>
__global__ __launch_bounds__(KH_CHUNK_THREADS) void k_build_fused(KhFusedParams P) {
  __shared__ unsigned long long lk[KH_DD_M];
  __shared__ unsigned long long liv[KH_DD_M];
  __shared__ uint32_t set[KH_HS];               // de-dup index set, later cnt/fill = set[0..L) and start = set[L..2L)
  // the chunk image is kept as 16-bit entries (11-bit index into the staged records | 5-bit distance code; 0xFFFF = empty slot): 4.3 KB instead of the
  // 28 KB of a (key, value, info) image, which keeps the kernel at 53.7 KB of LDS = 3 workgroups per CU
  __shared__ __align__(8) uint16_t simg[KH_L + KH_FSPILL];
  __shared__ KhMP32 s_wtot[KH_CHUNK_THREADS / 64];
  __shared__ uint32_t s_chunk, s_x, s_max, s_abort;
  __shared__ long long s_pend;
  static_assert(KH_HS >= 2 * KH_L, "set[] is reused as cnt/fill + start");
  uint32_t* cnt = set;
  uint32_t* start = set + KH_L;
  const uint32_t tid = threadIdx.x;
  const uint64_t cap = P.New.cap, mask_n = cap - 1;
  const uint32_t nch = (uint32_t)(cap >> KH_LB);          // host guarantees cap >= 2 * KH_L
  KH_STAMP(0);
  if (tid == 0) {
    s_chunk = blockIdx.x;
    s_max = 0;
    s_abort = (SRC == 0 || SRC == 2) ? (uint32_t)__hip_atomic_load(&P.est[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  }
  __syncthreads();
  // the source table of a streamed insert lives in simg[] until the records are staged: every byte of LDS counts here
  // (53.7 KB = three workgroups per CU; 200 bytes more and only two fit, which costs 45% of the kernel's throughput)
  const ulonglong2** s_ptr = reinterpret_cast<const ulonglong2**>(simg);
  uint32_t* s_cum = reinterpret_cast<uint32_t*>(simg) + 2 * KH_MAX_SRC;
  static_assert((KH_L + KH_FSPILL) * 2 >= KH_MAX_SRC * 8 + (KH_MAX_SRC + 1) * 4, "source table fits the image array");
  const uint32_t c = s_chunk;
  const uint64_t Sc = (uint64_t)c * KH_L;
  const unsigned long long VALID = 1ull << 63;
  uint32_t m, rep_mask;
  uint32_t vote_old = 0, vote_rec = 0;      // early give-up vote: elements that were in the table, records of the batch
  // (record indices travel as 11-bit fields next to a 5-bit distance code, 0xFFFF = empty slot: index 2047 stays unused)
  if (SRC == 0) {
    const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
    const KhSrcView V = kh_src_setup<true>(P.src, q, s_ptr, s_cum);
    m = V.m;
    KH_STAMP(1);
    vote_rec = m;
    const bool aborted = s_abort != 0;
    if (m >= KH_DD_M || aborted) {     // does not fit the staging area / speculation given up: general path
      if (tid == 0) {
        // (an aborted launch was flagged once by the workgroup that gave up: 65 K atomics on one word would cost 2 ms)
        if (!aborted) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
        __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    // ---- de-dup (as k_dedup, single round)
    const bool nodup = P.nodup && c != 0;            // (chunk 0 is parked, not placed here: it keeps the fold)
    if (!nodup) for (uint32_t s = tid; s < KH_HS; s += KH_CHUNK_THREADS) set[s] = 0;
    if (m) {   // all (<= 4) records of a lane are requested before the first one is stored: one HBM round trip, not four.  (Clamped
      // indices, no branch per record: with one the compiler waits for every load in turn.)
      ulonglong2 rr[KH_DD_M / KH_CHUNK_THREADS];
      const uint32_t last = m - 1u;
      if (V.one12) {
        KhRec12 r3[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; r3[it] = V.one12[i < last ? i : last]; }
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) rr[it] = make_ulonglong2(r3[it].klo | ((uint64_t)r3[it].khi << 32), (unsigned long long)r3[it].val);
      } else if (V.one) {
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; rr[it] = V.one[i < last ? i : last]; }
      } else {
        const ulonglong2* pp[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; pp[it] = kh_src_addr(V, s_ptr, s_cum, i < last ? i : last); }
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) rr[it] = *pp[it];
      }
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
        const uint32_t i = it * KH_CHUNK_THREADS + tid;
        if (i < m) { lk[i] = rr[it].x; liv[i] = rr[it].y; }
      }
    }
    if (V.n > 1) __syncthreads();       // the source table (in simg[]) has been read by every lane
    for (uint32_t i = tid; i < KH_L + KH_FSPILL; i += KH_CHUNK_THREADS) simg[i] = 0xFFFFu;
    __syncthreads();
    if (nodup) {                        // every record stands for itself; equal keys are looked for after the placement
      rep_mask = 0;
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) if (it * KH_CHUNK_THREADS + tid < m) rep_mask |= 1u << it;
    } else rep_mask = kh_dd_fold(lk, liv, set, m, P.mode, P.seed.xk);
    __syncthreads();
    KH_STAMP(2);
  } else if (SRC == 2) {
    // ---- insert into a non-empty table: the chunk's current elements + the batch's records of this chunk, folded together
    const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
    const KhSrcView V = kh_src_setup(P.src, q, s_ptr, s_cum);
    for (uint32_t s = tid; s < KH_HS; s += KH_CHUNK_THREADS) set[s] = 0;
    if (s_abort != 0) {                // speculation given up by the vote of the first chunks (flagged there)
      if (tid == 0) __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    const uint32_t n_old = kh_stage_from_table<KIND, HASH, true>(P.R, c, Sc, lk, liv, &s_x, &s_max);     // (ends with a barrier)
    KH_STAMP(1);
    m = n_old + V.m;
    vote_old = n_old; vote_rec = V.m;
    if (n_old >= KH_DD_M || m >= KH_DD_M) {
      if (tid == 0) {
        atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
        __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (V.m) {   // (all of a lane's records requested before the first one is stored: clamped indices, as above)
      ulonglong2 rr[KH_DD_M / KH_CHUNK_THREADS];
      const uint32_t last = V.m - 1u;
      if (V.one) {
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; rr[it] = V.one[i < last ? i : last]; }
      } else {
        const ulonglong2* pp[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; pp[it] = kh_src_addr(V, s_ptr, s_cum, i < last ? i : last); }
#pragma unroll
        for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) rr[it] = *pp[it];
      }
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
        const uint32_t i = it * KH_CHUNK_THREADS + tid;
        if (i < V.m) { lk[n_old + i] = rr[it].x; liv[n_old + i] = rr[it].y + (1ull << 32); }       // position + 1: 0 is "already in the table"
      }
    }
    if (V.n > 1) __syncthreads();
    for (uint32_t i = tid; i < KH_L + KH_FSPILL; i += KH_CHUNK_THREADS) simg[i] = 0xFFFFu;
    __syncthreads();
    rep_mask = kh_dd_fold(lk, liv, set, m, P.mode, P.seed.xk);
    __syncthreads();
    KH_STAMP(2);
  } else if (SRC == 3) {
    // ---- batch erase without random access into HBM: the chunk's elements (home from the info byte: no hash) are staged next to the
    // batch's erase keys of this chunk (8-byte records, partitioned by chunk like an insert batch); an element whose key is among
    // them is dropped -- what erase_no_resize's backward shift leaves (hashmap_robinhood.hpp:1294-1356)
    const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
    for (uint32_t s = tid; s < KH_L; s += KH_CHUNK_THREADS) set[s] = 0;      // (list heads per home bucket, below)
    // the erase keys are requested BEFORE the table's slots (clamped indices, no branch per key: all loads of a lane in flight
    // together), so that the chunk pays one HBM round trip for both, not two in a row -- and with a histogram-free partition (a slot
    // whose place does not depend on its fill) together with the slot's cursor as well: indices clamped to the slot, whatever lies
    // behind the fill is dropped below
    uint64_t kk[KH_DD_M / KH_CHUNK_THREADS];
    KhSrcView V;
    V.n = 1; V.one = nullptr; V.one12 = nullptr;
    if (P.src.slot[0]) {
      const uint64_t b = (uint64_t)q * P.src.slot[0];
      V.one8 = reinterpret_cast<const uint64_t*>(P.src.rec[0]) + b;
      const uint32_t last = (uint32_t)P.src.slot[0] - 1u;
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; kk[it] = V.one8[i < last ? i : last]; }
      const uint64_t e = P.src.cur[0][q] - b;
      V.m = (uint32_t)(e < P.src.slot[0] ? e : P.src.slot[0]);
    } else {
      V = kh_src_setup8(P.src, q);
      const uint32_t last = V.m ? V.m - 1u : 0u;
      const uint64_t* src8 = V.m ? V.one8 : reinterpret_cast<const uint64_t*>(P.R.Old.s);      // (an empty partition: any valid address)
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; kk[it] = src8[i < last ? i : last]; }
    }
    const uint32_t n_old = kh_stage_from_table<KIND, HASH, false>(P.R, c, Sc, lk, liv, &s_x, &s_max);     // (ends with a barrier)
    m = n_old + V.m;
    if (n_old >= KH_DD_M || m >= KH_DD_M) {      // denser than the staging area: the caller marks and re-lays out instead
      if (tid == 0) {
        atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
        __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const uint32_t i = it * KH_CHUNK_THREADS + tid;
      if (i < V.m) lk[n_old + i] = kk[it];
    }
    for (uint32_t i = tid; i < KH_L + KH_FSPILL; i += KH_CHUNK_THREADS) simg[i] = 0xFFFFu;
    __syncthreads();
    // an erase key can only meet elements of its own home bucket: the erase keys (a few per cent of the chunk's records) are chained per
    // home bucket -- set[b] = head of bucket b's list (index + 1, 0 = none), the link in the key's unused liv[] entry -- and every element
    // of the table looks at the head of ITS bucket (home from the info byte, no hash): nearly always 0
    for (uint32_t i = tid; i < V.m; i += KH_CHUNK_THREADS) {
      const uint32_t b = (uint32_t)((kh_hash64<HASH>(lk[n_old + i], P.seed) & mask_n) - Sc) & (KH_L - 1u);
      liv[n_old + i] = (unsigned long long)atomicExch(&set[b], n_old + i + 1u);
    }
    __syncthreads();
    uint32_t keep = 0;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const uint32_t x = it * KH_CHUNK_THREADS + tid;
      if (x < n_old) {
        const unsigned long long key = lk[x];
        uint32_t e = set[(uint32_t)(liv[x] >> 32) & (KH_L - 1u)];
        bool hit = false;
        while (e) {
          if (kh_keq(lk[e - 1u], key, P.seed.xk)) { hit = true; break; }
          e = (uint32_t)liv[e - 1u];
        }
        if (!hit) keep |= 1u << it;
      }
    }
    __syncthreads();
    rep_mask = keep;
  } else {
    m = kh_stage_from_table<KIND, HASH, false>(P.R, c, Sc, lk, liv, &s_x, &s_max);
    if (m >= KH_DD_M) {                // denser than the staging area: general path
      if (tid == 0) {
        atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
        __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    for (uint32_t i = tid; i < KH_L + KH_FSPILL; i += KH_CHUNK_THREADS) simg[i] = 0xFFFFu;
    rep_mask = 0;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) if (it * KH_CHUNK_THREADS + tid < m) rep_mask |= 1u << it;
    __syncthreads();
  }
  // ---- home counts of the distinct keys (set[] is dead from here on)
  for (uint32_t i = tid; i < 2 * KH_L; i += KH_CHUNK_THREADS) set[i] = 0;
  __syncthreads();
  uint32_t hb[KH_DD_M / KH_CHUNK_THREADS];
  uint32_t my_max = 0;
#pragma unroll
  for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
    hb[it] = 0;
    const uint32_t x = it * KH_CHUNK_THREADS + tid;
    if ((rep_mask >> it) & 1u) {
      hb[it] = (SRC == 0 || SRC == 2) ? (uint32_t)((kh_hash64<HASH>(lk[x], P.seed) & mask_n) - Sc) : ((uint32_t)(liv[x] >> 32) & 0x3FFFFFFFu);
      atomicAdd(&cnt[hb[it]], 1u);
      // position + 1 of the first occurrence of a NEW key (SRC 2: the field is already shifted, 0 = the key was in the table)
      if (SRC == 0 && P.mode == KH_DEDUP_FIRST) { const uint32_t ix = (uint32_t)(liv[x] >> 32) + 1u; my_max = ix > my_max ? ix : my_max; }
      if (SRC == 2 && P.mode == KH_DEDUP_FIRST) { const uint32_t ix = (uint32_t)(liv[x] >> 32); my_max = ix > my_max ? ix : my_max; }
    }
  }
  my_max = kh_wave_max(my_max);
  if ((tid & 63) == 0 && my_max) atomicMax(&s_max, my_max);
  __syncthreads();
  KH_STAMP(3);
  uint32_t cb[KH_HOMES_PER_THREAD];
  KhMP32 v; v.A = KH_MP32_NEG; v.n = 0;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    cb[j] = cnt[b];
    KhMP32 h; h.A = (int)(b + cb[j]); h.n = (int)cb[j];
    v = kh_mp_combine(v, h);
  }
  KhMP32 total;
  const KhMP32 excl = kh_block_scan_mp32(v, s_wtot, &total);
  const uint32_t n_c = (uint32_t)total.n;
  const long long spill0 = total.A > (long long)KH_L ? total.A - (long long)KH_L : 0;
  const bool early = n_c + KH_XB <= KH_L;
  KH_STAMP(4);
  if (c == 0) {     // circular table: chunk 0 follows the last chunk -> publish, park, and leave the placement to the tail launch
    if (tid == 0) {
      if (!early) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      // (12-byte records carry no stream position: a fold that merged equal keys could not tell which came first)
      if (SRC == 0 && P.src.rec12 && n_c != m) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[0], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      P.maxidx[0] = s_max;
      s_x = 0;
    }
#pragma unroll
    for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) P.homecnt0[tid * KH_HOMES_PER_THREAD + j] = (uint16_t)cb[j];
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const bool rep = (rep_mask >> it) & 1u;
      const uint32_t pos = kh_wave_append(rep, &s_x);
      if (rep) { const uint32_t x = it * KH_CHUNK_THREADS + tid; P.ck0[pos] = lk[x]; P.cv0[pos] = (uint32_t)liv[x]; }
    }
    return;
  }
  // ---- publish / look back
  if (tid == 0) {
    if ((SRC == 0 || SRC == 2) && c < 64) {     // the first 64 chunks vote on the duplicate ratio (same-address atomics are kept off the other 65 K)
      const unsigned long long mine = ((unsigned long long)(n_c - vote_old) << 32) | vote_rec;     // new distinct keys, records
      const unsigned long long tot = atomicAdd(&P.est[0], mine) + mine;      // < 64 * 2048 records: the low word cannot carry
      const uint32_t sn = (uint32_t)(tot >> 32), sm = (uint32_t)tot;
      if ((c == 63 || (nch < 64 && c == nch - 1)) && sm > 0) {
        const double dhat = (double)P.base_size + (double)P.n_total * (double)sn / (double)sm * 1.15;
        if (dhat <= (double)P.half_max_load) {
          atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
          __hip_atomic_store(&P.est[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (early) __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t pc = c - 1;
    unsigned long long w = 0;
    const long long t0 = clock64();
    for (;;) {
      w = __hip_atomic_load(&P.pub[pc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w & VALID) break;
      if (clock64() - t0 > P.poll_limit) { atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u); w = VALID; break; }   // bounded: ~4 ms, then general path
      __builtin_amdgcn_s_sleep(4);
    }
    const uint32_t x = (uint32_t)((w >> 32) & 0x7FFFFFFFu);
    if (x > KH_XB) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);       // a carry chain: general path
    if (!early) {
      const long long e = (long long)x + n_c;
      const long long pe = total.A > e ? total.A : e;
      const long long sp = pe > (long long)KH_L ? pe - (long long)KH_L : 0;
      __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)sp << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_x = x;
    P.maxidx[c] = s_max;
  }
  __syncthreads();
  KH_STAMP(5);
  // ---- placement with the carry-in (as k_chunk_place)
  const long long xr = (long long)s_x;
  long long p = excl.A > xr + excl.n ? excl.A : xr + excl.n;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    const long long st = p > (long long)b ? p : (long long)b;
    start[b] = (uint32_t)st;
    p = st + cb[j];
    cnt[b] = 0;                 // becomes the fill counter
  }
  if (tid == KH_CHUNK_THREADS - 1) s_pend = p;
  __syncthreads();
  uint32_t pr[KH_DD_M / KH_CHUNK_THREADS];       // slot (relative to the chunk) every record of this lane went to
#pragma unroll
  for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
    pr[it] = 0;
    if ((rep_mask >> it) & 1u) {
      const uint32_t x = it * KH_CHUNK_THREADS + tid;
      const uint32_t b = hb[it];
      const uint32_t r = atomicAdd(&cnt[b], 1u);
      const uint32_t prel = start[b] + r;
      pr[it] = prel;
      uint32_t dist = prel - b;
      if (KIND == KHK_RH && dist > 127u) { atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u); dist = 127u; }
      if (prel < KH_L + KH_FSPILL) simg[prel] = (uint16_t)(x | ((dist < 31u ? dist : 31u) << 11));   // record index | distance code
      else kh_slot_st(P.New.s + ((Sc + prel) & mask_n), lk[x], (uint32_t)liv[x], KIND == KHK_RH ? (0x80u | dist) : 0x00u);
    }
  }
  __syncthreads();
  KH_STAMP(6);
  if (SRC == 0 && P.nodup && c != 0) {
    // equal keys share their home bucket, hence sit in one group of consecutive slots [start[b], start[b] + cnt[b]): every
    // element compares itself with the elements of its group BEHIND it (0.4 of them on average at load 0.8; walking the whole
    // group to find oneself first made this check a quarter of the chunk's time)
    bool dup = false;
    uint32_t gend[KH_DD_M / KH_CHUNK_THREADS];                      // pr[it] walks over the slots behind the record up to gend[it]
    unsigned long long mykey[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      gend[it] = 0;
      if ((rep_mask >> it) & 1u) {
        const uint32_t b = hb[it];
        gend[it] = start[b] + cnt[b];
        if (gend[it] > KH_L + KH_FSPILL) { dup = true; gend[it] = 0; }       // part of the group went past the image: cannot be checked here
      }
      ++pr[it];
      mykey[it] = lk[it * KH_CHUNK_THREADS + tid];
    }
    // the first KH_DUPK elements behind every record are looked up without a branch (all LDS reads of a lane in flight together:
    // two dependent LDS latencies for the whole step); only groups longer than that (one bucket in 700 at load 0.8) enter the loop
    constexpr uint32_t KH_DUPK = 3;
    uint32_t ei[KH_DD_M / KH_CHUNK_THREADS][KH_DUPK];
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it)
#pragma unroll
      for (uint32_t d = 0; d < KH_DUPK; ++d) ei[it][d] = simg[pr[it] + d < gend[it] ? pr[it] + d : 0u] & 0x7FFu;
    bool more = false;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
#pragma unroll
      for (uint32_t d = 0; d < KH_DUPK; ++d)
        if (kh_keq(lk[ei[it][d]], mykey[it], P.seed.xk) && pr[it] + d < gend[it]) dup = true;
      pr[it] += KH_DUPK;
      more = more || pr[it] < gend[it];
    }
    while (__any(more)) {
      more = false;
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
        if (pr[it] < gend[it]) {
          if (kh_keq(lk[simg[pr[it]] & 0x7FFu], mykey[it], P.seed.xk)) dup = true;
          ++pr[it];
          more = more || pr[it] < gend[it];
        }
      }
    }
    if (__any(dup) && (tid & 63) == 0) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
  }
  KH_STAMP(7);
  long long pend = s_pend;
  if (pend < (long long)KH_L) pend = KH_L;
  const uint32_t lo = (uint32_t)xr;
  const uint32_t hi = pend < (long long)(KH_L + KH_FSPILL) ? (uint32_t)pend : (KH_L + KH_FSPILL);
  for (uint32_t s0 = lo + tid; s0 < hi; s0 += KH_CHUNK_THREADS) {
    KhSlot* dst = P.New.s + ((Sc + s0) & mask_n);
    const uint32_t e = simg[s0];
    if (e == 0xFFFFu) kh_slot_st(dst, 0, 0, kh_empty_info<KIND>());
    else {
      const uint32_t x = e & 0x7FFu;
      const uint64_t key = lk[x];
      uint32_t ib = 0x00u;
      if (KIND == KHK_RH) {     // distance = slot - home: the 5-bit code next to the record index; the rare long ones are re-derived from the key
        uint32_t dist = e >> 11;
        if (dist == 31u) {
          dist = s0 - (uint32_t)((kh_hash64<HASH>(key, P.seed) & mask_n) - Sc);
          if (dist > 127u) dist = 127u;
        }
        ib = 0x80u | dist;
      }
      kh_slot_st(dst, key, (uint32_t)liv[x], ib);      // one coalesced 16-byte store per slot: key, value and info byte together
    }
  }
  KH_STAMP(8);
}

// ---------------------------------------------------------------------------------------------
// The bulk build of a duplicate-free batch, lean: k_build_fused<KIND, HASH, 0> with nodup = 1 over ONE source of 12-byte records, i.e.
// the benchmark case (distinct keys into an empty table), as its own kernel so that its LDS can be cut to what that case needs:
// values as 32-bit words (no stream positions travel with 12-byte records), home counts / fill counters as packed 16-bit fields
// and slot starts as 16-bit words (a chunk holds < 2048 records), no de-dup set.  36.7 KB instead of 53.6 KB: FOUR workgroups per CU
// instead of three (the kernel is bound by latency and instruction issue, 77 % of the issue slots at three).  Same protocol as
// k_build_fused (published granules, one-deep look-back, flags, early give-up vote, chunk 0 parked for the tail launch); any
// duplicate -- found by the same group check after the placement, or by an all-pairs check in chunk 0, which is not placed here --
// raises KH_FLAG_FUSE_INVALID and the host repeats the batch with 16-byte records.
// ---------------------------------------------------------------------------------------------
#ifdef KH_TRACE
#define KH_STAMP_L(i) do { if (threadIdx.x == 0 && blockIdx.x >= 20000 && blockIdx.x < 20512) kh_trace[(blockIdx.x - 20000) * 12 + (i)] = clock64(); } while (0)
#else
#define KH_STAMP_L(i)
#endif
#define KH_LEAN_M 2016u          // records a chunk of the lean build may hold (its staging arrays: 384 bytes under KH_DD_M's, which is what lets the
                                 // sorted index below fit next to them in a quarter of a CU's LDS)
template <int KIND, int HASH>
__global__ __launch_bounds__(KH_CHUNK_THREADS, 8) void k_build_lean(KhFusedParams P) {      // (8 waves per SIMD = 4 workgroups per CU: <= 64 VGPRs, <= 96 SGPRs)
  __shared__ unsigned long long lk[KH_LEAN_M];
  __shared__ uint32_t lv[KH_LEAN_M];
  __shared__ __align__(16) uint32_t cnt16[KH_L / 2];           // two 16-bit counters per word: home counts, then fill counters (= group sizes at the end)
  __shared__ __align__(8) uint16_t start0[KH_L];                // first slot of every home bucket's group if nothing ran over from the chunk before
  // The records are counting-sorted by home bucket WITHOUT the carry-in of the chunk before: sidx[npre[b] + r] = record r of bucket b (npre =
  // records in the buckets before b).  Everything up to and including the duplicate check works on that order, so the look-back's HBM round
  // trip (a device-scope load: ~3 K cycles, 14 % of a chunk's time when the workgroup waited for it) is spent under the sort and the check;
  // only the slot image -- which needs the carry-in: slot = max(start0[b] + r, carry + npre[b] + r) -- is laid out afterwards, over the
  // then dead sort arrays.
  __shared__ __align__(16) uint16_t u_sort[KH_L + KH_LEAN_M];   // npre[KH_L] | sidx[KH_LEAN_M]; later simg[KH_L + KH_FSPILL]
  uint16_t* npre = u_sort;
  uint16_t* sidx = u_sort + KH_L;
  uint16_t* simg = u_sort;
  static_assert(KH_L + KH_LEAN_M >= KH_L + KH_FSPILL, "the slot image fits over the sort arrays");
  static_assert(((KH_L + KH_FSPILL) * 2) % 16 == 0 && (KH_L * 2) % 16 == 0, "filled with 16-byte stores");
  __shared__ KhMP32 s_wtot[KH_CHUNK_THREADS / 64];
  __shared__ uint32_t s_x, s_abort;
  __shared__ long long s_pend;
  const uint32_t tid = threadIdx.x;
  const uint64_t cap = P.New.cap, mask_n = cap - 1;
  const uint32_t nch = (uint32_t)(cap >> KH_LB);
  const uint32_t c = blockIdx.x;
  const uint64_t Sc = (uint64_t)c * KH_L;
  const unsigned long long VALID = 1ull << 63;
  KH_STAMP_L(0);
  if (tid == 0) s_abort = (uint32_t)__hip_atomic_load(&P.est[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (one histogram-free or exact source of 12-byte records: addressed directly, no source table)
  const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
  // All (<= 4) records of a lane are requested before the first one is stored (clamped indices, no branch per record).  A histogram-free
  // partition's records sit in a slot whose place does not depend on its fill: they are requested TOGETHER with the slot's cursor (indices
  // clamped to the slot, whatever lies behind the fill is dropped below) -- one HBM round trip per chunk, not cursor-then-records.
  const KhRec12* src; uint32_t m;
  KhRec12 r3[KH_DD_M / KH_CHUNK_THREADS];
  unsigned long long kreg[KH_DD_M / KH_CHUNK_THREADS];
  if (P.src.slot[0]) {
    const uint64_t b = (uint64_t)q * P.src.slot[0];
    src = reinterpret_cast<const KhRec12*>(P.src.rec[0]) + b;
    const uint32_t last = (uint32_t)P.src.slot[0] - 1u;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; r3[it] = src[i < last ? i : last]; }
    const uint64_t e = P.src.cur[0][q] - b;
    m = (uint32_t)(e < P.src.slot[0] ? e : P.src.slot[0]);
  } else {
    const uint64_t b = P.src.off[0][q];
    src = reinterpret_cast<const KhRec12*>(P.src.rec[0]) + b;
    m = (uint32_t)(P.src.off[0][q + 1] - b);
    const uint32_t last = m ? m - 1u : 0u;
    const KhRec12* from = m ? src : reinterpret_cast<const KhRec12*>(P.src.rec[0]);      // (an empty partition at the end of the buffer: any valid address)
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) { const uint32_t i = it * KH_CHUNK_THREADS + tid; r3[it] = from[i < last ? i : last]; }
  }
  kh_lds_fill16(cnt16, sizeof(cnt16), 0u);
  __syncthreads();
  const bool aborted = s_abort != 0;
  if (m >= KH_LEAN_M || aborted) {
    if (tid == 0) {
      if (!aborted) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // ---- records into LDS, home counts (the keys stay in registers for the hash and the duplicate check)
  uint32_t hb[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
  for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
    const uint32_t i = it * KH_CHUNK_THREADS + tid;
    kreg[it] = r3[it].klo | ((uint64_t)r3[it].khi << 32);
    hb[it] = 0;
    if (i < m) {
      lk[i] = kreg[it]; lv[i] = r3[it].val;
      hb[it] = (uint32_t)((kh_hash64<HASH>(kreg[it], P.seed) & mask_n) - Sc) & (KH_L - 1u);
      atomicAdd(&cnt16[hb[it] >> 1], 1u << (16 * (hb[it] & 1)));
    }
  }
  KH_STAMP_L(1);
  __syncthreads();
  KH_STAMP_L(2);
  uint32_t cb[KH_HOMES_PER_THREAD];
  KhMP32 v; v.A = KH_MP32_NEG; v.n = 0;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    cb[j] = (cnt16[b >> 1] >> (16 * (b & 1))) & 0xFFFFu;
    KhMP32 h; h.A = (int)(b + cb[j]); h.n = (int)cb[j];
    v = kh_mp_combine(v, h);
  }
  KhMP32 total;
  const KhMP32 excl = kh_block_scan_mp32(v, s_wtot, &total);
  const uint32_t n_c = (uint32_t)total.n;      // == m: every record stands for itself
  const long long spill0 = total.A > (long long)KH_L ? total.A - (long long)KH_L : 0;
  const bool early = n_c + KH_XB <= KH_L;
  KH_STAMP_L(3);
  if (c == 0) {     // circular table: chunk 0 follows the last chunk -> publish, park, and leave the placement to the tail launch
    // (not sorted here: every record is compared with the records behind it instead -- one workgroup, once per build)
    bool dup = false;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      const uint32_t x = it * KH_CHUNK_THREADS + tid;
      if (x < m) {
        for (uint32_t y = x + 1; y < m; ++y) dup = dup || kh_keq(lk[y], kreg[it], P.seed.xk);
      }
    }
    if (__any(dup) && (tid & 63) == 0) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
    if (tid == 0) {
      if (!early) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[0], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      P.maxidx[0] = 0;
    }
#pragma unroll
    for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) P.homecnt0[tid * KH_HOMES_PER_THREAD + j] = (uint16_t)cb[j];
    for (uint32_t x = tid; x < m; x += KH_CHUNK_THREADS) { P.ck0[x] = lk[x]; P.cv0[x] = lv[x]; }
    return;
  }
  // ---- publish (as k_build_fused); the look-back's load is issued here and collected after the duplicate check
  unsigned long long w0 = 0;
  if (tid == 0) {
    if (c < 64) {     // the first 64 chunks vote on the duplicate ratio (all records distinct here: the vote can only confirm)
      const unsigned long long mine = ((unsigned long long)n_c << 32) | m;
      const unsigned long long tot = atomicAdd(&P.est[0], mine) + mine;
      const uint32_t sn = (uint32_t)(tot >> 32), sm = (uint32_t)tot;
      if ((c == 63 || (nch < 64 && c == nch - 1)) && sm > 0) {
        const double dhat = (double)P.base_size + (double)P.n_total * (double)sn / (double)sm * 1.15;
        if (dhat <= (double)P.half_max_load) {
          atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
          __hip_atomic_store(&P.est[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (early) __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w0 = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    P.maxidx[c] = 0;
  }
  // ---- group starts without the carry-in, record counts in front of every bucket
  {
    long long p = excl.A > (long long)excl.n ? excl.A : excl.n;
    uint32_t np = (uint32_t)excl.n;
    uint32_t st4[KH_HOMES_PER_THREAD], np4[KH_HOMES_PER_THREAD];
#pragma unroll
    for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
      const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
      const long long st = p > (long long)b ? p : (long long)b;
      st4[j] = (uint32_t)st & 0xFFFFu;         // (< KH_L + 2048: fits 16 bits)
      np4[j] = np;
      p = st + cb[j];
      np += cb[j];
    }
    static_assert(KH_HOMES_PER_THREAD == 4, "two packed counter words, one 8-byte store of starts / counts per thread");
    reinterpret_cast<uint2*>(start0)[tid] = make_uint2(st4[0] | (st4[1] << 16), st4[2] | (st4[3] << 16));
    reinterpret_cast<uint2*>(npre)[tid] = make_uint2(np4[0] | (np4[1] << 16), np4[2] | (np4[3] << 16));
    if (tid == KH_CHUNK_THREADS - 1) s_pend = p;      // end of the layout without a carry-in
    // (the fill counters: every thread owns the two words of its four homes)
    cnt16[tid * (KH_HOMES_PER_THREAD / 2)] = 0; cnt16[tid * (KH_HOMES_PER_THREAD / 2) + 1] = 0;
  }
  __syncthreads();
  // ---- counting sort by home bucket: rank inside the group from the fill counter
  uint32_t pj[KH_DD_M / KH_CHUNK_THREADS];       // slot without carry-in (low half) | sorted index (high half) of every record of this lane
#pragma unroll
  for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
    pj[it] = 0;
    const uint32_t x = it * KH_CHUNK_THREADS + tid;
    if (x < m) {
      const uint32_t b = hb[it];
      const uint32_t r = (atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1))) >> (16 * (b & 1))) & 0xFFFFu;
      uint32_t j = npre[b] + r;
      if (j >= KH_LEAN_M) j = KH_LEAN_M - 1u;      // (cannot happen with consistent counts; keeps a wild index inside the array)
      sidx[j] = (uint16_t)x;
      pj[it] = (start0[b] + r) | (j << 16);
    }
  }
  __syncthreads();
  KH_STAMP_L(4);
  {
    // equal keys share their home bucket, hence sit next to each other in the sorted order: every record compares itself with the
    // records of its group BEHIND it (0.4 of them on average at load 0.8).  The first KH_DUPK are looked up without a branch (all LDS
    // reads of a lane in flight together); only groups longer than that (one bucket in 700 at load 0.8) enter the loop
    bool dup = false;
    uint32_t jn[KH_DD_M / KH_CHUNK_THREADS], gend[KH_DD_M / KH_CHUNK_THREADS];
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
      gend[it] = 0;
      const uint32_t x = it * KH_CHUNK_THREADS + tid;
      if (x < m) {
        const uint32_t b = hb[it];
        gend[it] = npre[b] + ((cnt16[b >> 1] >> (16 * (b & 1))) & 0xFFFFu);
        if (gend[it] > KH_LEAN_M) gend[it] = KH_LEAN_M;
      }
      jn[it] = (pj[it] >> 16) + 1u;
    }
    constexpr uint32_t KH_DUPK = 3;
    uint32_t ei[KH_DD_M / KH_CHUNK_THREADS][KH_DUPK];
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it)
#pragma unroll
      for (uint32_t d = 0; d < KH_DUPK; ++d) ei[it][d] = sidx[jn[it] + d < gend[it] ? jn[it] + d : 0u];
    bool more = false;
#pragma unroll
    for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
#pragma unroll
      for (uint32_t d = 0; d < KH_DUPK; ++d)
        if (kh_keq(lk[ei[it][d] < KH_LEAN_M ? ei[it][d] : 0u], kreg[it], P.seed.xk) && jn[it] + d < gend[it]) dup = true;
      jn[it] += KH_DUPK;
      more = more || jn[it] < gend[it];
    }
    while (__any(more)) {
      more = false;
#pragma unroll
      for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
        if (jn[it] < gend[it]) {
          const uint32_t e = sidx[jn[it]];
          if (kh_keq(lk[e < KH_LEAN_M ? e : 0u], kreg[it], P.seed.xk)) dup = true;
          ++jn[it];
          more = more || jn[it] < gend[it];
        }
      }
    }
    if (__any(dup) && (tid & 63) == 0) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
  }
  KH_STAMP_L(5);
  // ---- look back: the word requested above, or a poll if the chunk before had not published yet
  if (tid == 0) {
    unsigned long long w = w0;
    const long long t0 = clock64();
    while (!(w & VALID)) {
      if (clock64() - t0 > P.poll_limit) { atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u); w = VALID; break; }   // bounded: then the general path
      __builtin_amdgcn_s_sleep(4);
      w = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t x = (uint32_t)((w >> 32) & 0x7FFFFFFFu);
    if (x > KH_XB) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);       // a carry chain: general path
    if (!early) {
      const long long e = (long long)x + n_c;
      const long long pe = total.A > e ? total.A : e;
      const long long sp = pe > (long long)KH_L ? pe - (long long)KH_L : 0;
      __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)sp << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_x = x > KH_XB ? 0u : x;          // (flagged: the result is discarded; keep the positions inside what the image and the counters assume)
  }
  __syncthreads();                      // (also: the sort arrays are dead, the slot image takes their place)
  KH_STAMP_L(6);
  kh_lds_fill16(simg, (KH_L + KH_FSPILL) * 2u, 0xFFFFFFFFu);
  __syncthreads();
  // ---- slot image with the carry-in: slot = max(slot without it, carry + sorted index)
  const uint32_t xr = s_x;
#pragma unroll
  for (uint32_t it = 0; it < KH_DD_M / KH_CHUNK_THREADS; ++it) {
    const uint32_t x = it * KH_CHUNK_THREADS + tid;
    if (x < m) {
      const uint32_t b = hb[it];
      const uint32_t p0 = pj[it] & 0xFFFFu, pc = (pj[it] >> 16) + xr;
      const uint32_t prel = p0 > pc ? p0 : pc;
      uint32_t dist = prel - b;
      if (KIND == KHK_RH && dist > 127u) { atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u); dist = 127u; }
      if (prel < KH_L + KH_FSPILL) simg[prel] = (uint16_t)(x | ((dist < 31u ? dist : 31u) << 11));   // record index | distance code
      else kh_slot_st(P.New.s + ((Sc + prel) & mask_n), lk[x], lv[x], KIND == KHK_RH ? (0x80u | dist) : 0x00u);
    }
  }
  __syncthreads();
  long long pend = s_pend;
  if (pend < (long long)xr + n_c) pend = (long long)xr + n_c;
  if (pend < (long long)KH_L) pend = KH_L;
  const uint32_t lo = xr;
  const uint32_t hi = pend < (long long)(KH_L + KH_FSPILL) ? (uint32_t)pend : (KH_L + KH_FSPILL);
  for (uint32_t s0 = lo + tid; s0 < hi; s0 += KH_CHUNK_THREADS) {
    KhSlot* dst = P.New.s + ((Sc + s0) & mask_n);
    const uint32_t e = simg[s0];
    if (e == 0xFFFFu) kh_slot_st(dst, 0, 0, kh_empty_info<KIND>());
    else {
      const uint32_t x = e & 0x7FFu;
      const uint64_t key = lk[x];
      uint32_t ib = 0x00u;
      if (KIND == KHK_RH) {
        uint32_t dist = e >> 11;
        if (dist == 31u) {
          dist = s0 - (uint32_t)((kh_hash64<HASH>(key, P.seed) & mask_n) - Sc);
          if (dist > 127u) dist = 127u;
        }
        ib = 0x80u | dist;
      }
      kh_slot_st(dst, key, lv[x], ib);
    }
  }
  KH_STAMP_L(7);
}

// ---------------------------------------------------------------------------------------------
// Robin Hood batch erase as an ORDERED stream (hashmap_robinhood.hpp:1294-1356 applied to a whole batch): a Robin Hood table holds its
// elements in the order of their home buckets, so what a batch of backward-shift deletes leaves is the same sequence without the erased
// elements, every survivor at slot = max(home, slot of the one before + 1) -- a (max,+) scan over the chunk's slots IN SLOT ORDER.  Nothing
// is sorted, counted per bucket or ranked with atomics: a lane keeps its slots in registers (rows of 64 consecutive slots per wave: every
// load and store of a wave is 1 KB), looks each element up in the chunk's erase keys (chained per home bucket in LDS; home from the info
// byte, no hash), the scan runs through DPP, the survivors enter themselves into an image of the new chunk in LDS and the rows are stored
// from there (every slot once, coalesced: survivors stored straight from registers, empties in a second sweep, left half-written lines
// behind and ran no faster than the staging form).  40.8 KB of LDS, <= 64 VGPRs: four workgroups per CU; 0.84-0.9 ms for 10^7 of 10^8 keys = the
// 4.4 GB it moves at 5 TB/s, against 1.05-1.15 ms of k_build_fused<.., 3>, which it replaces wherever a chunk's erase keys fit one per lane
// (same protocol: granules with early publication, one-deep look-back, chunk 0 parked for the tail launch, flags -> the caller's mark +
// re-layout path); survivors whose slot cannot depend on the carry-in are laid out before the look-back's word is collected.
// ---------------------------------------------------------------------------------------------
#define KH_ES_MAXK 832u          // erase keys one chunk may receive (slot of the histogram-free partition: checked by the host); two per lane at most.
                                 // (832: the reference benchmark's own shape -- 10^7 keys out of a 2^25-bucket table, slots of 799 -- fits, and so does the LDS: 40.8 KB)
// inclusive (max,+) scan over the 64 lanes of a wave through DPP (see kh_mp32_dpp_step)
__device__ __forceinline__ KhMP32 kh_wave_scan_mp32(KhMP32 v) {
  v = kh_mp32_dpp_step<0x111, 0xF>(v);
  v = kh_mp32_dpp_step<0x112, 0xF>(v);
  v = kh_mp32_dpp_step<0x114, 0xF>(v);
  v = kh_mp32_dpp_step<0x118, 0xF>(v);
  v = kh_mp32_dpp_step<0x142, 0xA>(v);
  v = kh_mp32_dpp_step<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ KhMP32 kh_wave_last_mp32(KhMP32 v) {
  KhMP32 r;
  r.A = __builtin_amdgcn_readlane(v.A, 63);
  r.n = __builtin_amdgcn_readlane(v.n, 63);
  return r;
}
template <int HASH>
__global__ __launch_bounds__(KH_CHUNK_THREADS, 8) void k_erase_stream(KhFusedParams P) {
  constexpr uint32_t EH = KH_L / 2;                          // chain heads: home buckets b and b + 1024 share one (the walk compares keys)
  __shared__ __align__(16) uint32_t ehead[EH];               // head of the chain of erase keys (index + 1)
  __shared__ unsigned long long ek[KH_ES_MAXK];
  __shared__ uint16_t enext[KH_ES_MAXK];
  // the new chunk (+ what runs over), slot by slot: survivors enter themselves where they land, the rows are stored from here
  __shared__ unsigned long long ikey[KH_L + KH_FSPILL];
  __shared__ uint32_t ival[KH_L + KH_FSPILL];
  __shared__ __align__(16) uint8_t iinfo[KH_L + KH_FSPILL];  // 0 = empty slot (the table's own code)
  __shared__ KhMP32 s_wtot[KH_CHUNK_THREADS / 64];
  __shared__ uint32_t s_x;
  constexpr uint32_t NW = KH_CHUNK_THREADS / 64;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t cap = P.New.cap, mask_n = cap - 1;          // (same capacity as the source)
  const uint32_t c = blockIdx.x;
  const uint64_t Sc = (uint64_t)c * KH_L;
  const unsigned long long VALID = 1ull << 63;
  const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
  const KhSlot* __restrict__ old = P.R.Old.s;
  const uint32_t xk = P.seed.xk;
  const bool drop_marked = P.R.drop_marked != 0;
  // Slot order inside the workgroup: wave w owns the 256 slots behind 256 w, as four rows of 64 -- lane l holds slots 256 w + 64 k + l,
  // k = 0..3 (every load and store of a wave covers 1 KB of consecutive slots); the last wave goes on with the run-over window (128 slots
  // behind the chunk: what ran over from THIS chunk sits at its front) as two more rows.
  // Everything the chunk reads is requested at once: its erase keys (one per lane, clamped to the partition's slot: the fill arrives with
  // them; exact offsets -- small tables -- cost a dependent load in front of the keys, the slots are under way by then), the slots, the window.
  constexpr uint32_t ROWS = 6;
  const uint32_t base = wid * 256u + lane;
  const uint32_t nrows = wid == NW - 1 ? ROWS : 4u;          // (wave-uniform)
  const bool fixed = P.src.slot[0] != 0;
  uint4 w[ROWS];
  if (!fixed) {
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) if (k < nrows) w[k] = kh_slot_ld(old + ((Sc + base + 64u * k) & mask_n));
  }
  const uint64_t kb = fixed ? (uint64_t)q * P.src.slot[0] : P.src.off[0][q];
  const uint64_t kcount = fixed ? P.src.slot[0] : P.src.off[0][q + 1] - kb;
  const uint64_t* src8 = kcount ? reinterpret_cast<const uint64_t*>(P.src.rec[0]) + kb : reinterpret_cast<const uint64_t*>(old);      // (no key: any valid address)
  const uint32_t klast = kcount ? (uint32_t)(kcount < KH_ES_MAXK ? kcount : KH_ES_MAXK) - 1u : 0u;
  const unsigned long long mykey = src8[tid < klast ? tid : klast];
  const unsigned long long mykey2 = src8[tid + KH_CHUNK_THREADS < klast ? tid + KH_CHUNK_THREADS : klast];
  static_assert(KH_ES_MAXK <= 2 * KH_CHUNK_THREADS, "two erase keys per lane");
  if (fixed) {
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) if (k < nrows) w[k] = kh_slot_ld(old + ((Sc + base + 64u * k) & mask_n));
  }
  uint64_t fill = kcount;
  if (fixed) { fill = P.src.cur[0][q] - kb; if (fill > kcount) fill = kcount; }
  const uint32_t m_e = fill > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)fill;
  kh_lds_fill16(ehead, sizeof(ehead), 0u);
  kh_lds_fill16(iinfo, sizeof(iinfo), 0u);
  static_assert((KH_L + KH_FSPILL) % 16 == 0, "filled with 16-byte stores");
  __syncthreads();
  if (m_e > KH_ES_MAXK) {            // (fixed slots: the host sends only partitions whose slot fits; exact offsets: a partition this full -> the caller's other path)
    if (tid == 0) { atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u); __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return;
  }
  if (tid < m_e) {
    ek[tid] = mykey;
    const uint32_t b = (uint32_t)((kh_hash64<HASH>(mykey, P.seed) & mask_n) - Sc) & (KH_L - 1u);
    enext[tid] = (uint16_t)atomicExch(&ehead[b & (EH - 1u)], tid + 1u);
  }
  if (tid + KH_CHUNK_THREADS < m_e) {
    const uint32_t i = tid + KH_CHUNK_THREADS;
    ek[i] = mykey2;
    const uint32_t b = (uint32_t)((kh_hash64<HASH>(mykey2, P.seed) & mask_n) - Sc) & (KH_L - 1u);
    enext[i] = (uint16_t)atomicExch(&ehead[b & (EH - 1u)], i + 1u);
  }
  __syncthreads();
  // ---- (max,+) scan in slot order: row by row inside the wave (DPP), the waves' totals through LDS.  A slot counts if it holds an element
  // of THIS chunk (home from the info byte) whose key is not among the erase keys of its home bucket.
  // (composites are kept packed, A | n << 16: a survivor's A lies in [1, 2^13), and an A of 0 stands for "nothing yet" -- it loses every max)
  auto pack = [](KhMP32 x) -> uint32_t { return (uint32_t)(x.A < 0 ? 0 : x.A) | ((uint32_t)x.n << 16); };
  auto unpack = [](uint32_t p) -> KhMP32 { KhMP32 x; x.A = (int)(p & 0xFFFFu); x.n = (int)(p >> 16); return x; };
  const KhMP32 ident = {KH_MP32_NEG, 0};
  uint32_t keep = 0;
  uint32_t inc[ROWS];                // inclusive composite of every slot of mine
  KhMP32 wrun = ident;
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    inc[k] = 0;
    if (k < nrows) {
      const int srel = (int)(base + 64u * k);
      const uint32_t inf = w[k].w & 0xFFu;
      const int home = srel - (int)(inf & 0x7Fu);
      bool st = inf >= 0x80u && home >= 0 && home < (int)KH_L && !(drop_marked && (w[k].w & KH_INFO_ERASE_MARK));
      if (st && m_e) {
        const unsigned long long key = kh_slot_key(w[k]);
        uint32_t e = ehead[(uint32_t)home & (EH - 1u)];
        while (e) {
          if (kh_keq(ek[e - 1u], key, xk)) { st = false; break; }
          e = enext[e - 1u];
        }
      }
      KhMP32 h = ident;
      if (st) { keep |= 1u << k; h.A = home + 1; h.n = 1; }
      const KhMP32 sc = kh_wave_scan_mp32(h);
      inc[k] = pack(kh_mp_combine(wrun, sc));
      wrun = kh_mp_combine(wrun, kh_wave_last_mp32(sc));
    }
  }
  if (lane == 0) s_wtot[wid] = wrun;
  __syncthreads();
  KhMP32 wpre = ident, total = ident;
  for (uint32_t ww = 0; ww < NW; ++ww) {
    const KhMP32 x = s_wtot[ww];
    if (ww < wid) wpre = kh_mp_combine(wpre, x);
    total = kh_mp_combine(total, x);
  }
  const uint32_t n_c = (uint32_t)total.n;
  const long long spill0 = total.A > (int)KH_L ? (long long)total.A - (long long)KH_L : 0;
  const bool early = n_c + KH_XB <= KH_L;
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) inc[k] = pack(kh_mp_combine(wpre, unpack(inc[k])));
  if (c == 0) {     // circular table: chunk 0 follows the last chunk -> publish, park (survivors in slot order, home counts), tail launch places
    if (tid == 0) {
      if (!early) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[0], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      P.maxidx[0] = 0;
    }
    uint32_t* hcnt = ival;                           // (home counts of the parked chunk)
    kh_lds_fill16(hcnt, KH_L * 4u, 0u);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) {
      if ((keep >> k) & 1u) {
        const uint32_t idx = (inc[k] >> 16) - 1u;
        if (idx < KH_DD_M) { P.ck0[idx] = kh_slot_key(w[k]); P.cv0[idx] = w[k].z; }
        atomicAdd(&hcnt[(base + 64u * k) - (w[k].w & 0x7Fu)], 1u);
      }
    }
    __syncthreads();
    for (uint32_t b = tid; b < KH_L; b += KH_CHUNK_THREADS) P.homecnt0[b] = (uint16_t)hcnt[b];
    return;
  }
  // ---- publish; request the word of the chunk before; lay out what cannot depend on it meanwhile
  unsigned long long w0 = 0;
  if (tid == 0) {
    if (early) __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w0 = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    P.maxidx[c] = 0;
  }
  // a survivor's slot = max(A, carry + n) - 1 with (A, n) its inclusive composite; A >= KH_XB + n: no carry-in (<= KH_XB, or the launch is
  // discarded) can matter -> entered into the image at once; the others after the look-back
  auto enter = [&](const uint4& vv, uint32_t srel, uint32_t pos) {
    uint32_t dist = pos - (srel - (vv.w & 0x7Fu));
    if (dist > 127u) { atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u); dist = 127u; }
    if (pos < KH_L + KH_FSPILL) { ikey[pos] = kh_slot_key(vv); ival[pos] = vv.z; iinfo[pos] = (uint8_t)(0x80u | dist); }
  };
  uint32_t late = 0;
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    if ((keep >> k) & 1u) {
      const KhMP32 ic = unpack(inc[k]);
      if (ic.A >= (int)KH_XB + ic.n) enter(w[k], base + 64u * k, (uint32_t)ic.A - 1u);
      else late |= 1u << k;
    }
  }
  if (tid == 0) {
    unsigned long long ww = w0;
    const long long t0 = clock64();
    while (!(ww & VALID)) {
      if (clock64() - t0 > P.poll_limit) { atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u); ww = VALID; break; }   // bounded: then the caller's other path
      __builtin_amdgcn_s_sleep(4);
      ww = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t x = (uint32_t)((ww >> 32) & 0x7FFFFFFFu);
    if (x > KH_XB) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
    if (!early) {
      const long long e = (long long)x + n_c;
      const long long pe = (long long)total.A > e ? (long long)total.A : e;
      const long long sp = pe > (long long)KH_L ? pe - (long long)KH_L : 0;
      __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)sp << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_x = x > KH_XB ? 0u : x;
  }
  __syncthreads();
  const int xr = (int)s_x;
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    if ((late >> k) & 1u) {
      const KhMP32 ic = unpack(inc[k]);
      const int e = ic.A > xr + ic.n ? ic.A : xr + ic.n;
      enter(w[k], base + 64u * k, (uint32_t)e - 1u);
    }
  }
  __syncthreads();
  // ---- the new chunk, row by row: every slot of [carry-in, end of the layout) is stored once, 1 KB per wave and store
  {
    const int tA = total.A > 0 ? total.A : 0;
    int pend = tA > xr + (int)n_c ? tA : xr + (int)n_c;          // end of the layout
    if (pend < (int)KH_L) pend = (int)KH_L;
    if (pend > (int)(KH_L + KH_FSPILL)) pend = (int)(KH_L + KH_FSPILL);
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) {
      const int s0 = (int)(base + 64u * k);
      if (k < nrows && s0 >= xr && s0 < pend) {
        const uint32_t ib = iinfo[s0];
        KhSlot* dst = P.New.s + ((Sc + (uint32_t)s0) & mask_n);
        if (ib == 0u) kh_slot_st(dst, 0, 0, kh_empty_info<KHK_RH>());
        else kh_slot_st(dst, ikey[s0], ival[s0], ib);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Insert into a loaded Robin Hood table at EQUAL capacity as an ordered stream (what k_build_fused<.., 2> does, for batches of up to
// KH_IS_MAXR records per chunk): the chunk's elements stay in registers in slot order (rows of 64 slots per wave, as k_erase_stream) and
// never pass through a staging area or the de-dup set; the batch's records of the chunk (a few hundred) are chained per home bucket in LDS,
// settle their own duplicates by walking their chain (smallest stream position wins; Reducer = std::plus: the winner takes the sum), and
// every element of the table looks its bucket's chain up -- a record whose key the table holds is dropped (or added to the element).  Home
// counts (elements first: their rank inside a bucket is what the counter returns; then the surviving records), a (max,+) scan over the
// buckets, and every entry enters the image of the new chunk at max(start0[b] + rank, carry + npre[b] + rank); rows stored coalesced.
// One HBM round trip in front (cursor, records and slots together) instead of three.  Same protocol and fall-backs as k_build_fused<.., 2>
// (granules, look-back, early give-up vote on the duplicate ratio, chunk 0 parked, flags -> general path).  52.8 KB of LDS: three workgroups per CU.
// ---------------------------------------------------------------------------------------------
#define KH_IS_MAXR 640u          // records of the batch one chunk may receive
template <int HASH>
__global__ __launch_bounds__(KH_CHUNK_THREADS, 6) void k_insert_stream(KhFusedParams P) {
  __shared__ unsigned long long ikey[KH_L + KH_FSPILL];      // the new chunk (+ what runs over), slot by slot
  __shared__ uint32_t ival[KH_L + KH_FSPILL];
  __shared__ __align__(16) uint8_t iinfo[KH_L + KH_FSPILL];  // 0 = empty slot
  __shared__ __align__(16) uint32_t cnt16[KH_L / 2];         // two 16-bit counters per word: entries per home bucket (the value returned = rank inside the bucket)
  __shared__ __align__(16) uint16_t start0[KH_L];            // first slot of a bucket's group without a carry-in; before the scan: chain heads (u32[1024])
  __shared__ __align__(8) uint16_t npre[KH_L];               // entries in the buckets before
  __shared__ unsigned long long rk[KH_IS_MAXR];              // the batch's records of this chunk: key, stream position + 1 << 32 | value
  __shared__ unsigned long long riv[KH_IS_MAXR];
  __shared__ uint16_t rnx[KH_IS_MAXR];                       // chain link
  __shared__ uint8_t rst[KH_IS_MAXR];                        // 0: survives; bit 0: a record with the same key came earlier; bit 1: the table holds the key
  __shared__ KhMP32 s_wtot[KH_CHUNK_THREADS / 64];
  __shared__ uint32_t s_x, s_abort, s_new, s_max;
  uint32_t* heads = reinterpret_cast<uint32_t*>(start0);
  constexpr uint32_t NW = KH_CHUNK_THREADS / 64, ROWS = 6;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t cap = P.New.cap, mask_n = cap - 1;          // (same capacity as the source)
  const uint32_t nch = (uint32_t)(cap >> KH_LB);
  const uint32_t c = blockIdx.x;
  const uint64_t Sc = (uint64_t)c * KH_L;
  const unsigned long long VALID = 1ull << 63;
  const uint32_t q = P.PB ? (__brev(c) >> (32 - P.PB)) : 0u;
  const KhSlot* __restrict__ old = P.R.Old.s;
  const uint32_t xk = P.seed.xk;
  const bool plus = P.mode == KH_DEDUP_PLUS;
  if (tid == 0) { s_abort = (uint32_t)__hip_atomic_load(&P.est[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_new = 0; s_max = 0; }
  // ---- everything the chunk reads, requested at once (fixed slots; exact offsets cost a dependent load in front of the records)
  const uint32_t base = wid * 256u + lane;
  const uint32_t nrows = wid == NW - 1 ? ROWS : 4u;
  const bool fixed = P.src.slot[0] != 0;
  uint4 w[ROWS];
  if (!fixed) {
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) if (k < nrows) w[k] = kh_slot_ld(old + ((Sc + base + 64u * k) & mask_n));
  }
  const uint64_t kb = fixed ? (uint64_t)q * P.src.slot[0] : P.src.off[0][q];
  const uint64_t kcount = fixed ? P.src.slot[0] : P.src.off[0][q + 1] - kb;
  const ulonglong2* srcr = kcount ? P.src.rec[0] + kb : reinterpret_cast<const ulonglong2*>(old);
  const uint32_t rlast = kcount ? (uint32_t)(kcount < KH_IS_MAXR ? kcount : KH_IS_MAXR) - 1u : 0u;
  ulonglong2 myrec[2];
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) { const uint32_t i = tid + u * KH_CHUNK_THREADS; myrec[u] = srcr[i < rlast ? i : rlast]; }
  static_assert(KH_IS_MAXR <= 2 * KH_CHUNK_THREADS, "two records per lane");
  if (fixed) {
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) if (k < nrows) w[k] = kh_slot_ld(old + ((Sc + base + 64u * k) & mask_n));
  }
  uint64_t fill = kcount;
  if (fixed) { fill = P.src.cur[0][q] - kb; if (fill > kcount) fill = kcount; }
  const uint32_t m_r = fill > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)fill;
  kh_lds_fill16(cnt16, sizeof(cnt16), 0u);
  kh_lds_fill16(start0, sizeof(start0), 0u);          // (the chain heads)
  kh_lds_fill16(iinfo, sizeof(iinfo), 0u);
  static_assert((KH_L + KH_FSPILL) % 16 == 0, "filled with 16-byte stores");
  __syncthreads();
  if (m_r > KH_IS_MAXR || s_abort != 0) {
    if (tid == 0) {
      if (s_abort == 0) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[c], VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // ---- the records: into LDS, chained per home bucket (buckets b and b + 1024 share a chain: the walks compare keys)
  uint32_t hbr[2];
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) {
    const uint32_t i = tid + u * KH_CHUNK_THREADS;
    hbr[u] = 0;
    if (i < m_r) {
      rk[i] = myrec[u].x; riv[i] = myrec[u].y + (1ull << 32);       // position + 1: 0 is "already in the table"
      rst[i] = 0;
      hbr[u] = (uint32_t)((kh_hash64<HASH>(myrec[u].x, P.seed) & mask_n) - Sc) & (KH_L - 1u);
      rnx[i] = (uint16_t)atomicExch(&heads[hbr[u] & 1023u], i + 1u);
    }
  }
  __syncthreads();
  // ---- duplicates inside the batch: the record with the smallest stream position stands for its key (and, Reducer = std::plus, takes the sum)
  uint32_t rsum[2];
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) {
    const uint32_t i = tid + u * KH_CHUNK_THREADS;
    rsum[u] = 0;
    if (i < m_r) {
      const unsigned long long key = myrec[u].x, mine = myrec[u].y + (1ull << 32);
      uint32_t sum = (uint32_t)mine;
      bool later = false;
      uint32_t e = heads[hbr[u] & 1023u];
      while (e) {
        const uint32_t j = e - 1u;
        if (j != i && kh_keq(rk[j], key, xk)) {
          const unsigned long long other = riv[j];
          if ((other >> 32) < (mine >> 32)) later = true;
          sum += (uint32_t)other;
        }
        e = rnx[j];
      }
      if (later) rst[i] = 1;
      rsum[u] = sum;
    }
  }
  __syncthreads();
  if (plus) {       // (uniform) the winners carry the sums from here on
#pragma unroll
    for (uint32_t u = 0; u < 2; ++u) {
      const uint32_t i = tid + u * KH_CHUNK_THREADS;
      if (i < m_r && rst[i] == 0) riv[i] = (riv[i] & 0xFFFFFFFF00000000ull) | rsum[u];
    }
    __syncthreads();
  }
  // ---- the table's elements, in slot order: each looks its bucket's chain up (a record of the same key is dropped, or added), then counts
  uint32_t keep = 0;
  uint32_t rank[ROWS];               // home bucket << 16 | rank inside it
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    rank[k] = 0;
    if (k < nrows) {
      const int srel = (int)(base + 64u * k);
      const uint32_t inf = w[k].w & 0xFFu;
      const int home = srel - (int)(inf & 0x7Fu);
      if (inf >= 0x80u && home >= 0 && home < (int)KH_L) {
        keep |= 1u << k;
        const unsigned long long key = kh_slot_key(w[k]);
        uint32_t e = heads[(uint32_t)home & 1023u];
        while (e) {
          const uint32_t j = e - 1u;
          if (kh_keq(rk[j], key, xk)) {
            if (plus && rst[j] == 0) w[k].z += (uint32_t)riv[j];      // (one element per key, one winner per key: no race)
            rst[j] |= 2;      // (byte writes of different lanes to one record: the same bit)
          }
          e = rnx[j];
        }
        const uint32_t r = (atomicAdd(&cnt16[(uint32_t)home >> 1], 1u << (16 * (home & 1))) >> (16 * (home & 1))) & 0xFFFFu;
        rank[k] = ((uint32_t)home << 16) | r;
      }
    }
  }
  __syncthreads();
  // ---- the surviving records count behind the elements of their bucket
  uint32_t rrank[2];
  uint32_t my_max = 0, my_new = 0;
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) {
    const uint32_t i = tid + u * KH_CHUNK_THREADS;
    rrank[u] = 0xFFFFFFFFu;
    if (i < m_r && rst[i] == 0) {
      const uint32_t b = hbr[u];
      rrank[u] = (atomicAdd(&cnt16[b >> 1], 1u << (16 * (b & 1))) >> (16 * (b & 1))) & 0xFFFFu;
      ++my_new;
      const uint32_t ix = (uint32_t)(riv[i] >> 32);                  // stream position + 1 of a NEW key's first occurrence
      my_max = ix > my_max ? ix : my_max;
    }
  }
  my_new = kh_wave_sum(my_new);
  my_max = kh_wave_max(my_max);
  if (lane == 0) { if (my_new) atomicAdd(&s_new, my_new); if (my_max && P.mode == KH_DEDUP_FIRST) atomicMax(&s_max, my_max); }
  __syncthreads();
  // ---- (max,+) scan over the home buckets: group starts without the carry-in, entries in front of every bucket
  uint32_t cb[KH_HOMES_PER_THREAD];
  KhMP32 v; v.A = KH_MP32_NEG; v.n = 0;
#pragma unroll
  for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
    const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
    cb[j] = (cnt16[b >> 1] >> (16 * (b & 1))) & 0xFFFFu;
    KhMP32 h; h.A = (int)(b + cb[j]); h.n = (int)cb[j];
    v = kh_mp_combine(v, h);
  }
  KhMP32 total;
  const KhMP32 excl = kh_block_scan_mp32(v, s_wtot, &total);
  const uint32_t n_c = (uint32_t)total.n;
  const uint32_t n_new = s_new, n_old = n_c - n_new;
  const long long spill0 = total.A > (int)KH_L ? (long long)total.A - (long long)KH_L : 0;
  const bool early = n_c + KH_XB <= KH_L;
  {
    long long p = excl.A > (long long)excl.n ? excl.A : excl.n;
    uint32_t np = (uint32_t)excl.n;
    uint32_t st4[KH_HOMES_PER_THREAD], np4[KH_HOMES_PER_THREAD];
#pragma unroll
    for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) {
      const uint32_t b = tid * KH_HOMES_PER_THREAD + j;
      const long long st = p > (long long)b ? p : (long long)b;
      st4[j] = (uint32_t)st & 0xFFFFu;
      np4[j] = np;
      p = st + cb[j];
      np += cb[j];
    }
    static_assert(KH_HOMES_PER_THREAD == 4, "one 8-byte store of starts / counts per thread");
    reinterpret_cast<uint2*>(start0)[tid] = make_uint2(st4[0] | (st4[1] << 16), st4[2] | (st4[3] << 16));      // (the chain heads are dead: every walk ended two barriers ago)
    reinterpret_cast<uint2*>(npre)[tid] = make_uint2(np4[0] | (np4[1] << 16), np4[2] | (np4[3] << 16));
  }
  __syncthreads();
  if (c == 0) {     // circular table: chunk 0 follows the last chunk -> publish, park (entries in bucket order, home counts), tail launch places
    if (tid == 0) {
      if (!early) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
      __hip_atomic_store(&P.pub[0], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      P.maxidx[0] = s_max;
    }
#pragma unroll
    for (uint32_t j = 0; j < KH_HOMES_PER_THREAD; ++j) P.homecnt0[tid * KH_HOMES_PER_THREAD + j] = (uint16_t)cb[j];
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) {
      if ((keep >> k) & 1u) {
        const uint32_t idx = npre[rank[k] >> 16] + (rank[k] & 0xFFFFu);
        if (idx < KH_DD_M) { P.ck0[idx] = kh_slot_key(w[k]); P.cv0[idx] = w[k].z; }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < 2; ++u) {
      if (rrank[u] != 0xFFFFFFFFu) {
        const uint32_t i = tid + u * KH_CHUNK_THREADS;
        const uint32_t idx = npre[hbr[u]] + rrank[u];
        if (idx < KH_DD_M) { P.ck0[idx] = rk[i]; P.cv0[idx] = (uint32_t)riv[i]; }
      }
    }
    return;
  }
  // ---- publish, vote, request the word of the chunk before
  unsigned long long w0 = 0;
  if (tid == 0) {
    if (c < 64) {     // the first 64 chunks vote on the duplicate ratio (as k_build_fused<.., 2>)
      const unsigned long long mine = ((unsigned long long)n_new << 32) | m_r;     // new distinct keys, records
      const unsigned long long tot = atomicAdd(&P.est[0], mine) + mine;
      const uint32_t sn = (uint32_t)(tot >> 32), sm = (uint32_t)tot;
      if ((c == 63 || (nch < 64 && c == nch - 1)) && sm > 0) {
        const double dhat = (double)P.base_size + (double)P.n_total * (double)sn / (double)sm * 1.15;
        if (dhat <= (double)P.half_max_load) {
          atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
          __hip_atomic_store(&P.est[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (early) __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)spill0 << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w0 = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    P.maxidx[c] = s_max;
  }
  // ---- every entry into the image: slot = max(start0[b] + rank, carry + npre[b] + rank); what cannot depend on the carry-in (<= KH_XB) at once
  auto enter = [&](unsigned long long key, uint32_t val, uint32_t b, uint32_t pos) {
    uint32_t dist = pos - b;
    if (dist > 127u) { atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u); dist = 127u; }
    if (pos < KH_L + KH_FSPILL) { ikey[pos] = key; ival[pos] = val; iinfo[pos] = (uint8_t)(0x80u | dist); }
    else kh_slot_st(P.New.s + ((Sc + pos) & mask_n), key, val, 0x80u | dist);
  };
  uint32_t late = 0;
  uint32_t pj[ROWS + 2];             // slot without carry-in | sorted index << 16
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    pj[k] = 0;
    if ((keep >> k) & 1u) {
      const uint32_t b = rank[k] >> 16, r = rank[k] & 0xFFFFu;
      const uint32_t p0 = start0[b] + r, j = npre[b] + r;
      pj[k] = (p0 & 0xFFFFu) | (j << 16);
      if (p0 >= KH_XB + j) enter(kh_slot_key(w[k]), w[k].z, b, p0); else late |= 1u << k;
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) {
    pj[ROWS + u] = 0;
    if (rrank[u] != 0xFFFFFFFFu) {
      const uint32_t i = tid + u * KH_CHUNK_THREADS;
      const uint32_t b = hbr[u], r = rrank[u];
      const uint32_t p0 = start0[b] + r, j = npre[b] + r;
      pj[ROWS + u] = (p0 & 0xFFFFu) | (j << 16);
      if (p0 >= KH_XB + j) enter(rk[i], (uint32_t)riv[i], b, p0); else late |= 1u << (ROWS + u);
    }
  }
  if (tid == 0) {
    unsigned long long ww = w0;
    const long long t0 = clock64();
    while (!(ww & VALID)) {
      if (clock64() - t0 > P.poll_limit) { atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u); ww = VALID; break; }   // bounded: then the general path
      __builtin_amdgcn_s_sleep(4);
      ww = __hip_atomic_load(&P.pub[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t x = (uint32_t)((ww >> 32) & 0x7FFFFFFFu);
    if (x > KH_XB) atomicOr(&P.flags[KH_FLAG_FUSE_INVALID], 1u);
    if (!early) {
      const long long e = (long long)x + n_c;
      const long long pe = (long long)total.A > e ? (long long)total.A : e;
      const long long sp = pe > (long long)KH_L ? pe - (long long)KH_L : 0;
      __hip_atomic_store(&P.pub[c], VALID | ((unsigned long long)sp << 32) | n_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_x = x > KH_XB ? 0u : x;
  }
  __syncthreads();
  const uint32_t xr = s_x;
#pragma unroll
  for (uint32_t k = 0; k < ROWS; ++k) {
    if ((late >> k) & 1u) {
      const uint32_t p0 = pj[k] & 0xFFFFu, pc = (pj[k] >> 16) + xr;
      enter(kh_slot_key(w[k]), w[k].z, rank[k] >> 16, p0 > pc ? p0 : pc);
    }
  }
#pragma unroll
  for (uint32_t u = 0; u < 2; ++u) {
    if ((late >> (ROWS + u)) & 1u) {
      const uint32_t i = tid + u * KH_CHUNK_THREADS;
      const uint32_t p0 = pj[ROWS + u] & 0xFFFFu, pc = (pj[ROWS + u] >> 16) + xr;
      enter(rk[i], (uint32_t)riv[i], hbr[u], p0 > pc ? p0 : pc);
    }
  }
  __syncthreads();
  // ---- the new chunk, row by row: every slot of [carry-in, end of the layout) is stored once, 1 KB per wave and store
  {
    const int tA = total.A > 0 ? total.A : 0;
    int pend = tA > (int)xr + (int)n_c ? tA : (int)xr + (int)n_c;
    if (pend < (int)KH_L) pend = (int)KH_L;
    if (pend > (int)(KH_L + KH_FSPILL)) pend = (int)(KH_L + KH_FSPILL);
#pragma unroll
    for (uint32_t k = 0; k < ROWS; ++k) {
      const int s0 = (int)(base + 64u * k);
      if (k < nrows && s0 >= (int)xr && s0 < pend) {
        const uint32_t ib = iinfo[s0];
        KhSlot* dst = P.New.s + ((Sc + (uint32_t)s0) & mask_n);
        if (ib == 0u) kh_slot_st(dst, 0, 0, kh_empty_info<KHK_RH>());
        else kh_slot_st(dst, ikey[s0], ival[s0], ib);
      }
    }
  }
  (void)n_old;
}

// carry-in of chunk 0 = run-over of the last chunk (circular table)
// (also what the tail placement of chunk 0 needs besides: its list offsets {0, 0} and its list length = the count field of pub[0] --
//  one launch instead of a kernel, a fill and a copy)
__global__ void k_fused_tail_carry(const unsigned long long* __restrict__ pub, uint32_t nch, long long* __restrict__ xcarry0,
                                   uint64_t* __restrict__ noff0 = nullptr, uint32_t* __restrict__ ncnt0 = nullptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    xcarry0[0] = (long long)((pub[nch - 1] >> 32) & 0x7FFFFFFFull);
    if (noff0) { noff0[0] = 0; noff0[1] = 0; }
    if (ncnt0) ncnt0[0] = (uint32_t)pub[0];
  }
}

// totals of a fused build: sum of the per-chunk counts (low word of the published granules) and max of maxidx.
// totals[0..1] are zero at launch; every workgroup reduces its slice and adds it with ONE atomic per value.
__global__ __launch_bounds__(1024) void k_fused_totals(const unsigned long long* __restrict__ pub, const uint32_t* __restrict__ maxidx, uint32_t nch,
                                                       unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long ws[16];
  __shared__ uint32_t wm[16];
  unsigned long long sum = 0; uint32_t mx = 0;
  for (uint32_t c = blockIdx.x * 1024 + threadIdx.x; c < nch; c += gridDim.x * 1024) { sum += pub[c] & 0xFFFFFFFFull; const uint32_t v = maxidx[c]; mx = v > mx ? v : mx; }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); const uint32_t o = __shfl_down(mx, off, 64); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = sum; wm[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0; uint32_t m = 0;
    for (int w = 0; w < 16; ++w) { t += ws[w]; m = wm[w] > m ? wm[w] : m; }
    if (t) atomicAdd(&totals[0], t);
    if (m) atomicMax(&totals[1], (unsigned long long)m);
  }
}

// RH displacement histogram (REPROBE_STAT-style oracle)
__global__ void k_disp_hist(const KhSlot* __restrict__ slots, uint64_t cap, unsigned long long* __restrict__ out128) {
  __shared__ uint32_t h[128];
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) { uint32_t b = slots[i].info & 0xFFu; if (b >= 0x80u) atomicAdd(&h[b & 0x7Fu], 1u); }
  __syncthreads();
  if (threadIdx.x < 128 && h[threadIdx.x]) atomicAdd(&out128[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// SoA view of the table for the host-side exports (to_vector, export_info, export_slots): keys / values / info bytes and the
// occupied flags the generic compaction takes; any output may be null
template <int KIND>
__global__ void k_unpack_slots(const KhSlot* __restrict__ slots, uint64_t cap, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                               uint8_t* __restrict__ info, uint8_t* __restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) {
    const uint4 w = kh_slot_ld(slots + i);
    if (keys) keys[i] = kh_slot_key(w);
    if (vals) vals[i] = w.z;
    if (info) info[i] = (uint8_t)(w.w & 0xFFu);
    if (flags) flags[i] = kh_is_occupied<KIND>(w.w & 0xFFu) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// multi-GPU sharding: stable partition of (key,value) by rank = hash(key, seed) mod p
// (distributed_batched_robinhood_map.hpp:513-534 key_to_rank, :632-741 assign_count_permute)
// ---------------------------------------------------------------------------------------------
#define KH_SHARD_THREADS 512
#define KH_SHARD_TILE (KH_SHARD_THREADS * 8)     // 8 consecutive items per lane; fewer, larger tiles keep the [rank][tile] offset scan short
#define KH_SHARD_MAXR 64
struct KhShardAdj { long long a[8]; __host__ __device__ KhShardAdj() { for (int i = 0; i < 8; ++i) a[i] = 0; } };
// entries (r, i) -> tile_off[r * ntiles + ntiles * i / pieces] for r <= p... (the offsets at the piece boundaries of a kh_shard_plan)
__global__ void k_shard_piece_bounds(const uint64_t* __restrict__ tile_off, uint32_t ntiles, uint32_t p, uint32_t pieces, uint64_t* __restrict__ out /* [p][pieces+1] */) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p * (pieces + 1)) return;
  const uint32_t r = j / (pieces + 1), i = j % (pieces + 1);
  out[j] = tile_off[(uint64_t)r * ntiles + (uint64_t)ntiles * i / pieces];
}
template <int HASH>
__device__ __forceinline__ uint32_t kh_rank_of(uint64_t key, KhSeed seed, uint32_t p, uint32_t pmask) {
  uint64_t h = kh_hash64<HASH>(key, seed);
  return pmask ? (uint32_t)(h & pmask) : (uint32_t)(h % p);
}
template <int HASH>
__global__ void k_shard_count(const uint64_t* __restrict__ keys, uint64_t n, KhSeed seed, uint32_t p, uint32_t pmask,
                              uint32_t* __restrict__ tile_counts /* [p][ntiles] */, uint32_t ntiles) {
  __shared__ uint32_t h[KH_SHARD_MAXR];
  if (threadIdx.x < KH_SHARD_MAXR) h[threadIdx.x] = 0;
  __syncthreads();
  uint64_t base = (uint64_t)blockIdx.x * KH_SHARD_TILE;
  if (p <= 8) {
    // per-lane counts in 16-bit fields of two 64-bit words, reduced over the wave with shuffles: 8 LDS atomics per wave
    // instead of one per key on 8 hot bins
    unsigned long long c0 = 0, c1 = 0;
    // all eight keys of a lane are requested before the first one is hashed (clamped indices: a branch per key makes the
    // compiler wait for every load in turn)
    uint64_t key[KH_SHARD_TILE / KH_SHARD_THREADS];
#pragma unroll
    for (uint32_t k = 0; k < KH_SHARD_TILE / KH_SHARD_THREADS; ++k) {
      const uint64_t i = base + threadIdx.x + k * KH_SHARD_THREADS;
      key[k] = keys[i < n ? i : n - 1];
    }
#pragma unroll
    for (uint32_t k = 0; k < KH_SHARD_TILE / KH_SHARD_THREADS; ++k) {
      const uint64_t i = base + threadIdx.x + k * KH_SHARD_THREADS;
      if (i < n) {
        const uint32_t r = kh_rank_of<HASH>(key[k], seed, p, pmask);
        if (r < 4) c0 += 1ull << (16 * r); else c1 += 1ull << (16 * (r - 4));
      }
    }
    for (int off = 32; off > 0; off >>= 1) { c0 += __shfl_down(c0, off, 64); c1 += __shfl_down(c1, off, 64); }
    if ((threadIdx.x & 63) == 0)
      for (uint32_t r = 0; r < p; ++r) {
        const uint32_t c = (uint32_t)(((r < 4 ? c0 : c1) >> (16 * (r & 3))) & 0xFFFFu);
        if (c) atomicAdd(&h[r], c);
      }
  } else {
    for (uint32_t j = threadIdx.x; j < KH_SHARD_TILE; j += KH_SHARD_THREADS) {
      uint64_t i = base + j;
      if (i < n) atomicAdd(&h[kh_rank_of<HASH>(keys[i], seed, p, pmask)], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < p) tile_counts[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}
template <int HASH>
__global__ void k_shard_scatter(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, KhSeed seed,
                                uint32_t p, uint32_t pmask, const uint64_t* __restrict__ tile_off /* [p][ntiles] exclusive */,
                                uint32_t ntiles, uint64_t* __restrict__ ok, uint32_t* __restrict__ ov) {
  // stable: lane t owns items [8t, 8t+8) of the tile; per rank, an exclusive scan over lanes gives the order
  __shared__ uint32_t wtot[KH_SHARD_THREADS / 64][KH_SHARD_MAXR];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint64_t base = (uint64_t)blockIdx.x * KH_SHARD_TILE + (uint64_t)tid * 8;
  uint64_t key[8]; uint32_t rk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint64_t i = base + j;
    key[j] = i < n ? keys[i] : 0;
    rk[j] = i < n ? kh_rank_of<HASH>(key[j], seed, p, pmask) : 0xFFFFFFFFu;
  }
  for (uint32_t r = 0; r < p; ++r) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) c += (rk[j] == r) ? 1u : 0u;
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) wtot[wid][r] = incl;
    __syncthreads();
    uint32_t wpre = 0;
    for (uint32_t w = 0; w < wid; ++w) wpre += wtot[w][r];
    uint64_t pos = tile_off[(uint64_t)r * ntiles + blockIdx.x] + wpre + incl - c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (rk[j] == r) {
        ok[pos] = key[j];
        if (vals) ov[pos] = vals[base + j];
        ++pos;
      }
    }
  }
}

// p <= 8 ranks (one node): all per-rank prefix sums in one pass -- the 8 per-lane counters (<= 8 each) travel as
// 16-bit fields of two 64-bit words through a single wave scan -- and the tile is staged in LDS in (rank, input
// order) so that the write-out is coalesced.  Stable, like the generic kernel.
template <int HASH>
__global__ __launch_bounds__(KH_SHARD_THREADS) void k_shard_scatter8(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n,
                                                          KhSeed seed, uint32_t p, uint32_t pmask,
                                                          const uint64_t* __restrict__ tile_off /* [p][ntiles] exclusive */,
                                                          uint32_t ntiles, uint64_t* __restrict__ ok, uint32_t* __restrict__ ov,
                                                          uint32_t tile0 = 0, KhShardAdj adj = KhShardAdj()) {
  // (tile0 / adj: the launch covers the tiles [tile0, tile0 + gridDim.x) of a larger batch whose offsets tile_off holds -- one piece
  //  of a kh_shard_plan; keys / vals / n are the piece's own, adj[r] turns the batch-wide offset of rank r into the piece's)
  __shared__ uint64_t lk[KH_SHARD_TILE];
  __shared__ uint32_t lv[KH_SHARD_TILE];
  __shared__ unsigned long long wtot[KH_SHARD_THREADS / 64][2];
  __shared__ uint32_t rank_off[9];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t tbase = (uint64_t)blockIdx.x * KH_SHARD_TILE;
  const uint64_t base = tbase + (uint64_t)tid * 8;
  const uint32_t tile_len = (n - tbase) < KH_SHARD_TILE ? (uint32_t)(n - tbase) : KH_SHARD_TILE;
  uint64_t key[8]; uint32_t val[8]; uint32_t rk[8];
  unsigned long long c0 = 0, c1 = 0;     // counts of ranks 0-3 / 4-7, 16 bits each
  // (all of a lane's loads first, clamped indices: see k_shard_count)
#pragma unroll
  for (int j = 0; j < 8; ++j) { const uint64_t i = base + j; key[j] = keys[i < n ? i : n - 1]; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { const uint64_t i = base + j; val[j] = vals ? vals[i < n ? i : n - 1] : 0u; }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint64_t i = base + j;
    rk[j] = 0xFFu;
    if (i < n) {
      rk[j] = kh_rank_of<HASH>(key[j], seed, p, pmask);
      if (rk[j] < 4) c0 += 1ull << (16 * rk[j]); else c1 += 1ull << (16 * (rk[j] - 4));
    }
  }
  unsigned long long i0 = c0, i1 = c1;
  for (int off = 1; off < 64; off <<= 1) {
    unsigned long long o0 = __shfl_up(i0, off, 64), o1 = __shfl_up(i1, off, 64);
    if (lane >= (uint32_t)off) { i0 += o0; i1 += o1; }
  }
  if (lane == 63) { wtot[wid][0] = i0; wtot[wid][1] = i1; }
  __syncthreads();
  unsigned long long e0 = i0 - c0, e1 = i1 - c1, t0 = 0, t1 = 0;
  for (uint32_t w = 0; w < KH_SHARD_THREADS / 64; ++w) {
    if (w < wid) { e0 += wtot[w][0]; e1 += wtot[w][1]; }
    t0 += wtot[w][0]; t1 += wtot[w][1];
  }
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t r = 0; r < 8; ++r) {
      rank_off[r] = run;
      run += (uint32_t)(((r < 4 ? t0 : t1) >> (16 * (r & 3))) & 0xFFFFu);
    }
    rank_off[8] = run;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (rk[j] != 0xFFu) {
      const uint32_t r = rk[j];
      const uint32_t within = (uint32_t)(((r < 4 ? e0 : e1) >> (16 * (r & 3))) & 0xFFFFu);
      const uint32_t s = rank_off[r] + within;
      lk[s] = key[j]; lv[s] = val[j];
      if (r < 4) e0 += 1ull << (16 * r); else e1 += 1ull << (16 * (r - 4));
    }
  }
  __syncthreads();
  for (uint32_t s = tid; s < tile_len; s += KH_SHARD_THREADS) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 1; k < 8; ++k) r += (s >= rank_off[k]) ? 1u : 0u;
    const uint64_t pos = tile_off[(uint64_t)r * ntiles + tile0 + blockIdx.x] + (uint64_t)adj.a[r] + (s - rank_off[r]);
    ok[pos] = lk[s];
    if (vals) ov[pos] = lv[s];
  }
}

// ---------------------------------------------------------------------------------------------
// HyperLogLog register update (reference hyperloglog64.hpp:175-188 internal_update):
//   v = hash << ignored_msb ; register index = top `precision` bits of v ; rank = clz((v << precision) | mask) + 1
//   with mask = low (precision + ignored_msb) bits set ; register = max(register, rank).
// Registers are accumulated per workgroup in LDS (precision <= 13) and merged with global atomicMax.
// ---------------------------------------------------------------------------------------------
template <int HASH, bool FROM_KEYS>
__global__ void k_hll_update(const uint64_t* __restrict__ in, uint64_t n, KhSeed seed, uint32_t precision, uint32_t ignored,
                             uint32_t* __restrict__ regs, int use_lds) {
  extern __shared__ __align__(16) uint32_t kh_dyn_smem[];
  const uint32_t m = 1u << precision;
  if (use_lds) {
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) kh_dyn_smem[i] = 0;
    __syncthreads();
  }
  const uint64_t lzc_mask = ~0ull >> (64 - precision - ignored);
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t hv = FROM_KEYS ? kh_hash64<HASH>(in[i], seed) : in[i];
    const uint64_t v = hv << ignored;
    const uint32_t r = (uint32_t)(v >> (64 - precision));
    const uint32_t rank = (uint32_t)__clzll((long long)((v << precision) | lzc_mask)) + 1u;
    if (use_lds) atomicMax(&kh_dyn_smem[r], rank); else atomicMax(&regs[r], rank);
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) { const uint32_t v = kh_dyn_smem[j]; if (v) atomicMax(&regs[j], v); }
  }
}
__global__ void k_hll_merge(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t m) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { const uint32_t a = dst[i], b = src[i]; dst[i] = a > b ? a : b; }
}

// ---------------------------------------------------------------------------------------------
// k-mer generation front end (the reference gets this from kmerind's KmerParser, which is absent):
// every window of k valid bases (A,C,G,T, either case) of a byte sequence becomes one 2-bit packed k-mer, first
// base in the most significant position (bliss::common::Kmer::nextFromChar order), A=0 C=1 G=2 T=3; any other byte
// (newline, N, ...) breaks the run.  CANON: min(k-mer, reverse complement).  One lane rolls over 64 start positions.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t kh_dna_code(uint32_t c) {
  c &= 0xDFu;
  return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}
// Tile of 4096 start positions per 256-lane workgroup.  A lane packs ITS 16 bases (one 16-byte load, coalesced) into a 32-bit word
// (2 bits per base, first base in the top bits) plus a 16-bit mask of the bytes that are no base; the words go to LDS.  The 16
// windows that start in a lane's word are then cut out of three consecutive words with funnel shifts: no byte-wise rolling loop,
// no strided global access.  (The first version read the text one byte at a time at a stride of 64 bytes between lanes and
// wrote 8-byte k-mers at a stride of 512: 47 GB/s.)  Two passes over the text: count the valid windows per tile, scan, then
// emit them compacted and in order (staged in LDS, written coalesced) -- the full-size k-mer / flag arrays are gone.
#define KH_KM_TILE 4096
#define KH_KM_THREADS 256
struct KhKmerWin { uint64_t a; uint64_t lo; uint64_t inv; };     // bases 0..31 | bases 32..47 in the top half | 48 invalid bits (base b at bit 47 - b)
__device__ __forceinline__ void kh_km_pack_tile(const uint8_t* __restrict__ seq, uint64_t n, uint64_t tile0, uint32_t* words, uint16_t* invs) {
  const uint32_t tid = threadIdx.x;
  // lanes 0..255 pack the tile's words, lanes 0..1 also the two halo words behind it
  for (uint32_t w = tid; w < KH_KM_TILE / 16 + 2; w += KH_KM_THREADS) {
    const uint64_t p0 = tile0 + (uint64_t)w * 16;
    uint8_t b[16];
    if (p0 + 16 <= n && ((reinterpret_cast<uintptr_t>(seq) + p0) & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4*>(seq + p0);
      const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 16; ++q) b[q] = (uint8_t)(vv[q >> 2] >> (8 * (q & 3)));
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) b[q] = p0 + q < n ? seq[p0 + q] : (uint8_t)'\n';
    }
    uint32_t word = 0, inv = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const uint32_t c = kh_dna_code(b[q]);
      word = (word << 2) | (c & 3u);
      inv = (inv << 1) | (c > 3u ? 1u : 0u);
    }
    words[w] = word; invs[w] = (uint16_t)inv;
  }
}
__device__ __forceinline__ KhKmerWin kh_km_window(const uint32_t* words, const uint16_t* invs, uint32_t t) {
  KhKmerWin W;
  W.a = ((uint64_t)words[t] << 32) | words[t + 1];
  W.lo = (uint64_t)words[t + 2] << 32;
  W.inv = ((uint64_t)invs[t] << 32) | ((uint64_t)invs[t + 1] << 16) | invs[t + 2];
  return W;
}
// the window that starts at base j (0..15) of the lane's word: valid iff none of its k bytes is a non-base
__device__ __forceinline__ bool kh_km_valid(const KhKmerWin& W, uint32_t j, uint32_t k) {
  const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
  return ((W.inv >> (48u - j - k)) & kmask) == 0;
}
__device__ __forceinline__ uint64_t kh_km_forward(const KhKmerWin& W, uint32_t j, uint32_t k) {
  const uint64_t x = j ? ((W.a << (2 * j)) | (W.lo >> (64 - 2 * j))) : W.a;       // bases j.. left-aligned
  return x >> (64 - 2 * k);
}
__global__ __launch_bounds__(KH_KM_THREADS) void k_kmers_count(const uint8_t* __restrict__ seq, uint64_t n, uint32_t k, uint32_t* __restrict__ sums) {
  __shared__ uint32_t words[KH_KM_TILE / 16 + 2];
  __shared__ uint16_t invs[KH_KM_TILE / 16 + 2];
  __shared__ uint32_t wsum[KH_KM_THREADS / 64];
  const uint64_t tile0 = (uint64_t)blockIdx.x * KH_KM_TILE;
  kh_km_pack_tile(seq, n, tile0, words, invs);
  __syncthreads();
  const KhKmerWin W = kh_km_window(words, invs, threadIdx.x);
  uint32_t c = 0;
#pragma unroll
  for (uint32_t j = 0; j < 16; ++j) c += kh_km_valid(W, j, k) ? 1u : 0u;      // (bytes behind the text's end were packed as non-bases)
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
template <bool CANON>
__global__ __launch_bounds__(KH_KM_THREADS) void k_kmers_emit(const uint8_t* __restrict__ seq, uint64_t n, uint32_t k, const uint64_t* __restrict__ tile_off,
                                                              uint64_t* __restrict__ out) {
  __shared__ uint32_t words[KH_KM_TILE / 16 + 2];
  __shared__ uint16_t invs[KH_KM_TILE / 16 + 2];
  __shared__ uint32_t wtot[KH_KM_THREADS / 64];
  __shared__ uint64_t stage[KH_KM_TILE];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t tile0 = (uint64_t)blockIdx.x * KH_KM_TILE;
  kh_km_pack_tile(seq, n, tile0, words, invs);
  __syncthreads();
  const KhKmerWin W = kh_km_window(words, invs, tid);
  uint32_t vmask = 0;
#pragma unroll
  for (uint32_t j = 0; j < 16; ++j) vmask |= kh_km_valid(W, j, k) ? (1u << j) : 0u;
  const uint32_t mine = (uint32_t)__popc(vmask);
  uint32_t incl = mine;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += o; }
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  uint32_t pos = incl - mine, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < KH_KM_THREADS / 64; ++w) { const uint32_t c = wtot[w]; if (w < wid) pos += c; total += c; }
#pragma unroll
  for (uint32_t j = 0; j < 16; ++j) {
    if ((vmask >> j) & 1u) {
      const uint64_t fw = kh_km_forward(W, j, k);
      uint64_t v = fw;
      if (CANON) { const uint64_t rc = kh_revcomp(fw, k); v = fw < rc ? fw : rc; }
      stage[pos++] = v;
    }
  }
  __syncthreads();
  const uint64_t o = tile_off[blockIdx.x];
  for (uint32_t i = tid; i < total; i += KH_KM_THREADS) out[o + i] = stage[i];
}

// ---------------------------------------------------------------------------------------------
// FASTQ record structure on the device: records are 4 lines (@id, sequence, +, quality).  The line number of
// every byte is the number of '\n' before it: tile sums + scan + an in-tile prefix.  Bytes that are not on a sequence line
// (line % 4 != 1) become '\n', which the k-mer kernel treats as a run break -- so id / '+' / quality text (which may well
// consist of the letters ACGT) never yields k-mers, and k-mers never span reads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_newline_tile_sums(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t wsum[4];
  const uint64_t base = (uint64_t)blockIdx.x * KH_CMP_TILE + (uint64_t)threadIdx.x * 8;
  uint32_t c = 0;
  if (base + 8 <= n && (reinterpret_cast<uintptr_t>(text) & 7u) == 0) {      // one 8-byte load instead of eight dependent byte loads
    const uint64_t w = *reinterpret_cast<const uint64_t*>(text + base);
#pragma unroll
    for (int j = 0; j < 8; ++j) c += ((w >> (8 * j)) & 0xFFu) == (uint64_t)'\n' ? 1u : 0u;
  } else {
    for (int j = 0; j < 8; ++j) if (base + j < n && text[base + j] == '\n') ++c;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_fastq_mask(const uint8_t* __restrict__ text, uint64_t n, const uint64_t* __restrict__ tile_off,
                                                    uint8_t* __restrict__ out) {
  __shared__ uint32_t wtot[4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * KH_CMP_TILE + (uint64_t)tid * 8;
  uint8_t c[8];
  uint32_t mine = 0;
  if (base + 8 <= n && (reinterpret_cast<uintptr_t>(text) & 7u) == 0) {
    const uint64_t w = *reinterpret_cast<const uint64_t*>(text + base);
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = (uint8_t)(w >> (8 * j));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = base + j < n ? text[base + j] : (uint8_t)0;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) mine += c[j] == '\n' ? 1u : 0u;
  uint32_t incl = mine;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += o; }
  if (lane == 63) wtot[wid] = incl;
  __syncthreads();
  uint64_t line = tile_off[blockIdx.x] + (incl - mine);
  for (uint32_t w = 0; w < wid; ++w) line += wtot[w];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < n) out[base + j] = ((line & 3u) == 1u && c[j] != '\n') ? c[j] : (uint8_t)'\n';
    if (c[j] == '\n') ++line;
  }
}

// ---------------------------------------------------------------------------------------------
// Small batches in place.  A handful of keys does not justify re-laying out a table of 10^8 elements (1.3 ms): one lane
// applies them one after the other with the reference's own single-key algorithms -- Robin Hood insert with displacement
// (hashmap_robinhood.hpp:522-624), backward-shift erase (:1294-1356), linear-probe insert into the first deleted slot of
// the probe path or the first empty one (hashmap_linearprobe.hpp:430-513).  The host takes this path only when no call of
// the batch can trigger a doubling (size + n <= max_load), so the serial semantics (first value wins, update overwrites,
// std::plus adds) are exactly the reference's.  The Robin Hood info array stays canonical: it is the reference's own
// algorithm.  A displacement chain that would push an element past distance 127 is detected by a read-only dry run of
// the chain BEFORE the key is applied: the key and the rest of the batch are left to the general path (out[1] = keys done).
// ---------------------------------------------------------------------------------------------
enum { KH_SMALL_FIRST = 0, KH_SMALL_UPDATE = 1, KH_SMALL_PLUS = 2, KH_SMALL_ERASE = 3 };
template <int KIND, int HASH>
__global__ void k_small_batch(KhSlots T, const char* __restrict__ kbase, uint32_t kstride, const char* __restrict__ vbase, uint32_t vstride,
                              uint32_t vconst, uint32_t n, int op, KhSeed seed, unsigned long long* __restrict__ out /* [0] new/erased, [1] keys done */,
                              uint32_t* __restrict__ flags) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint64_t mask = T.cap - 1;
  KhSlot* const S = T.s;
  unsigned long long changed = 0;
  uint32_t i = 0;
  for (; i < n; ++i) {
    const uint64_t key = *reinterpret_cast<const uint64_t*>(kbase + (uint64_t)i * kstride);
    const uint32_t val = vbase ? *reinterpret_cast<const uint32_t*>(vbase + (uint64_t)i * vstride) : vconst;
    uint64_t p = kh_hash64<HASH>(key, seed) & mask;
    if (KIND == KHK_RH) {
      // ---- find phase (shared by insert and erase)
      uint32_t reprobe = 0x80u;
      bool found = false;
      for (;;) {
        const uint4 w = kh_slot_ld(S + p);
        const uint32_t inf = w.w & 0xFFu;
        if (inf < reprobe) break;
        if (inf == reprobe && kh_keq(kh_slot_key(w), key, seed.xk)) { found = true; break; }
        ++reprobe; p = (p + 1) & mask;
        if (reprobe > 0xFFu) break;
      }
      if (op == KH_SMALL_ERASE) {
        if (!found) continue;
        uint64_t q = (p + 1) & mask;
        for (;;) {                                          // occupied, distance >= 1: moves one slot towards its home
          const uint4 w = kh_slot_ld(S + q);
          if ((w.w & 0xFFu) <= 0x80u) break;
          kh_slot_st(S + p, kh_slot_key(w), w.z, (w.w & 0xFFu) - 1u);
          p = q; q = (q + 1) & mask;
        }
        kh_slot_st(S + p, 0, 0, 0x00u);
        ++changed;
        continue;
      }
      if (found) {
        if (op == KH_SMALL_UPDATE) S[p].val = val;
        else if (op == KH_SMALL_PLUS) S[p].val += val;
        continue;
      }
      if (reprobe > 0xFFu) break;                         // would sit past distance 127: general path decides
      // ---- dry run of the displacement chain from (p, reprobe): distances only, nothing is written
      {
        uint64_t pp = p; uint32_t r = reprobe; bool over = false;
        for (;;) {
          const uint32_t cur = S[pp].info & 0xFFu;
          if (cur == 0x00u) break;
          if (cur < r) r = cur;                           // the resident is displaced and travels on with its own distance
          ++r; pp = (pp + 1) & mask;
          if (r > 0xFFu) { over = true; break; }
        }
        if (over) break;
      }
      // ---- insert with displacement
      uint64_t ck = key; uint32_t cv = val; uint32_t r = reprobe;
      for (;;) {
        const uint4 w = kh_slot_ld(S + p);
        const uint32_t cur = w.w & 0xFFu;
        if (cur == 0x00u) { kh_slot_st(S + p, ck, cv, r); break; }
        if (cur < r) {
          kh_slot_st(S + p, ck, cv, r);
          ck = kh_slot_key(w); cv = w.z; r = cur;
        }
        ++r; p = (p + 1) & mask;
      }
      ++changed;
    } else {
      // linear probing: the key, or the first deleted slot of its probe path, or the first empty slot
      uint64_t ins = KH_NONE;
      bool found = false;
      for (uint64_t step = 0; step < T.cap; ++step) {
        const uint4 w = kh_slot_ld(S + p);
        const uint32_t inf = w.w & 0xFFu;
        if (inf == 0x40u) { if (ins == KH_NONE) ins = p; break; }
        if (inf >= 0x80u) { if (ins == KH_NONE) ins = p; }
        else if (kh_keq(kh_slot_key(w), key, seed.xk)) { found = true; break; }
        p = (p + 1) & mask;
      }
      if (found) {
        if (op == KH_SMALL_UPDATE) S[p].val = val;
        else if (op == KH_SMALL_PLUS) S[p].val += val;
        continue;
      }
      if (ins == KH_NONE) { atomicOr(&flags[KH_FLAG_INTERNAL], 1u); break; }     // full table: cannot happen below max_load
      kh_slot_st(S + ins, key, val, 0x00u);
      ++changed;
    }
  }
  out[0] = changed;
  out[1] = i;
}


// ---------------------------------------------------------------------------------------------
// Batches of middle size (10^2 .. 10^5 keys) into a LARGE Robin Hood table, in place.  The reference inserts such a batch
// in O(batch) (hashmap_robinhood.hpp:522-624); re-laying out the whole table for it costs O(table) (1.3-1.8 ms at 2^27
// buckets: 10^4 keys would run slower than one CPU thread).  Here the table is cut into REGIONS of KH_IP_L consecutive slots
// and every region is owned by ONE lane, which applies the keys whose home bucket lies in its region one after the other with
// the reference's single-key algorithms -- insert with displacement, backward-shift erase.  An operation that would read or
// write a slot outside the owner's region (the displacement chain or the shift runs over the region's end) is not started:
// the key goes to a deferred list.  No slot is touched by two lanes of one launch, so no inter-workgroup visibility is
// needed inside a launch (per-XCD L2s are not coherent with each other).  The deferred keys (a few %: a chain crosses a
// given boundary with probability ~ cluster length / 512) are binned again with the regions shifted by half a region --
// the old boundaries are interior now -- and what is deferred a second time (chains longer than half a region, bins that
// overflowed twice) is applied by a single lane.
// The keys of an insert are DISTINCT and ABSENT from the table (k_dedup has folded duplicates and tested membership), so the
// order in which they are applied does not matter: the Robin Hood layout is a function of the key set alone.
// The host takes this path for at most ~1 key per region (n <= capacity / 512): every lane then runs a chain of a few dozen
// dependent memory accesses; beyond that the whole-table re-layout, which streams, is faster.
// ---------------------------------------------------------------------------------------------
#define KH_IP_LB 9
#define KH_IP_L (1u << KH_IP_LB)  // slots per region
#define KH_IP_CAP 8              // keys binned per region and pass; more go to the deferred list
enum { KH_IP_INSERT = 0, KH_IP_ERASE = 1 };
enum { KH_IP_DONE = 0, KH_IP_LEAVES_REGION = 1, KH_IP_TOO_FAR = 2, KH_IP_ABSENT = 3 };

// The single-key algorithms below read the table in WINDOWS of KH_IP_W consecutive slots, all loads of a window in flight at
// once: a chain of 15 dependent slot accesses (what the textbook loops are) costs 15 memory round trips, two windows cost two.

#define KH_IP_W 16
// Robin Hood insert of a key known to be absent (hashmap_robinhood.hpp:522-624 without the equality test).  BOUNDED: every
// slot the decision depends on must lie in [reg_start, reg_start + KH_IP_L) (circular), otherwise nothing is written.
// Pass 1 walks the displacement chain on the info bytes alone (distances only: where does the chain end, does a distance
// pass 127, does it leave the region); pass 2 runs the same walk for real.
template <bool BOUNDED>
__device__ __forceinline__ int kh_rh_insert_absent(KhSlot* S, uint64_t mask, uint64_t home, uint64_t key, uint32_t val, uint64_t reg_start) {
  {
    uint64_t p0 = home;
    uint32_t r = 0x80u;                                   // distance code of the element that is looking for a slot
    bool end = false;
    while (!end) {
      uint32_t inf[KH_IP_W];
#pragma unroll
      for (int j = 0; j < KH_IP_W; ++j) inf[j] = S[(p0 + j) & mask].info & 0xFFu;
#pragma unroll
      for (int j = 0; j < KH_IP_W; ++j) {
        if (!end) {
          if (BOUNDED && ((p0 + j - reg_start) & mask) >= KH_IP_L) return KH_IP_LEAVES_REGION;
          const uint32_t cur = inf[j];
          if (cur == 0x00u) end = true;
          else {
            if (cur < r) r = cur;                         // the resident is displaced and travels on with its own distance
            ++r;
            if (r > 0xFFu) return KH_IP_TOO_FAR;
          }
        }
      }
      p0 += KH_IP_W;
    }
  }
  uint64_t p0 = home;
  uint64_t ck = key; uint32_t cv = val, r = 0x80u;
  bool done = false;
  while (!done) {
    uint4 w[KH_IP_W];
#pragma unroll
    for (int j = 0; j < KH_IP_W; ++j) w[j] = kh_slot_ld(S + ((p0 + j) & mask));
#pragma unroll
    for (int j = 0; j < KH_IP_W; ++j) {
      if (!done) {
        const uint32_t cur = w[j].w & 0xFFu;
        if (cur == 0x00u) { kh_slot_st(S + ((p0 + j) & mask), ck, cv, r); done = true; }
        else {
          if (cur < r) {
            kh_slot_st(S + ((p0 + j) & mask), ck, cv, r);
            ck = kh_slot_key(w[j]); cv = w[j].z; r = cur;
          }
          ++r;
        }
      }
    }
    p0 += KH_IP_W;
  }
  return KH_IP_DONE;
}

// Robin Hood erase by backward shift (hashmap_robinhood.hpp:1294-1356).  Pass 1 finds the key and the end of the shift (the
// first slot behind it that is empty or holds an element at its home); pass 2 moves the elements in between one slot down.
template <bool BOUNDED>
__device__ __forceinline__ int kh_rh_erase_one(KhSlot* S, uint64_t mask, uint64_t home, uint64_t key, uint64_t reg_start, uint32_t xk) {
  uint64_t at = 0, len = 0;                               // slot of the key; elements to move
  {
    uint64_t p0 = home;
    uint32_t r = 0x80u;
    int state = 0;                                        // 0 looking for the key, 1 looking for the end of the shift, 2 done
    while (state != 2) {
      uint4 w[KH_IP_W];
#pragma unroll
      for (int j = 0; j < KH_IP_W; ++j) w[j] = kh_slot_ld(S + ((p0 + j) & mask));
#pragma unroll
      for (int j = 0; j < KH_IP_W; ++j) {
        if (state != 2) {
          if (BOUNDED && ((p0 + j - reg_start) & mask) >= KH_IP_L) return KH_IP_LEAVES_REGION;
          const uint32_t inf = w[j].w & 0xFFu;
          if (state == 0) {
            if (inf < r) return KH_IP_ABSENT;
            if (inf == r && kh_keq(kh_slot_key(w[j]), key, xk)) { at = (p0 + j) & mask; state = 1; }
            else { ++r; if (r > 0xFFu) return KH_IP_ABSENT; }
          } else {
            if (inf <= 0x80u) state = 2; else ++len;
          }
        }
      }
      p0 += KH_IP_W;
    }
  }
  uint64_t p0 = (at + 1) & mask;
  for (uint64_t moved = 0; moved < len; moved += KH_IP_W) {
    uint4 w[KH_IP_W];
#pragma unroll
    for (int j = 0; j < KH_IP_W; ++j) w[j] = kh_slot_ld(S + ((p0 + j) & mask));
#pragma unroll
    for (int j = 0; j < KH_IP_W; ++j)
      if (moved + j < len) kh_slot_st(S + ((p0 + j - 1) & mask), kh_slot_key(w[j]), w[j].z, (w[j].w & 0xFFu) - 1u);
    p0 += KH_IP_W;
  }
  kh_slot_st(S + ((at + len) & mask), 0, 0, 0x00u);
  return KH_IP_DONE;
}

struct KhInplaceParams {
  KhSlots T; KhSeed seed;
  uint32_t ofs;                            // region r = slots [r * KH_IP_L + ofs, (r + 1) * KH_IP_L + ofs), circular
  // the input list, one of: (key, value) records | key / value arrays (in_v null: values 0) | the per-partition lists
  // k_dedup wrote (list q: part_cnt[q] entries of in_k / in_v from part_off[q] on; one workgroup per partition)
  const ulonglong2* in_rec;
  const uint64_t* in_k; const uint32_t* in_v;
  const uint64_t* part_off; const uint32_t* part_cnt; uint32_t nparts;
  uint64_t n; const unsigned long long* n_dev;  // list length: n_dev != null overrides n (a count produced on the device)
  uint32_t* cnt;                           // [regions] zero at launch
  ulonglong2* bins;                        // [regions * KH_IP_CAP]
  ulonglong2* defer; unsigned long long* n_defer;      // keys not applied by this pass (n_defer zero at launch)
  unsigned long long* n_done;              // keys inserted / erased (accumulates over the passes)
  unsigned long long* n_in;                // partition form: total list length (zero at launch)
  uint32_t* flags;
};

template <int HASH>
__device__ __forceinline__ void kh_ip_bin_one(const KhInplaceParams& P, ulonglong2 rec, uint64_t mask) {
  const uint64_t home = kh_hash64<HASH>(rec.x, P.seed) & mask;
  const uint32_t r = (uint32_t)(((home - P.ofs) & mask) >> KH_IP_LB);
  const uint32_t rank = atomicAdd(&P.cnt[r], 1u);
  if (rank < KH_IP_CAP) P.bins[(uint64_t)r * KH_IP_CAP + rank] = rec;
  else P.defer[atomicAdd(P.n_defer, 1ull)] = rec;
}
template <int HASH>
__global__ void k_ip_bin(KhInplaceParams P) {
  const uint64_t mask = P.T.cap - 1;
  if (P.part_off) {
    for (uint32_t q = blockIdx.x; q < P.nparts; q += gridDim.x) {
      const uint64_t b = P.part_off[q];
      const uint32_t c = P.part_cnt[q];
      if (threadIdx.x == 0 && c) atomicAdd(P.n_in, (unsigned long long)c);
      for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
        ulonglong2 rec; rec.x = P.in_k[b + i]; rec.y = P.in_v[b + i];
        kh_ip_bin_one<HASH>(P, rec, mask);
      }
    }
    return;
  }
  const uint64_t n = P.n_dev ? (uint64_t)*P.n_dev : P.n;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    ulonglong2 rec;
    if (P.in_rec) rec = P.in_rec[i];
    else { rec.x = P.in_k[i]; rec.y = P.in_v ? P.in_v[i] : 0u; }
    kh_ip_bin_one<HASH>(P, rec, mask);
  }
}

// one LANE per region
#define KH_IP_THREADS 1024
template <int HASH, int OP>
__global__ __launch_bounds__(KH_IP_THREADS) void k_ip_apply(KhInplaceParams P) {
  const uint64_t mask = P.T.cap - 1;
  const uint32_t regions = (uint32_t)(P.T.cap >> KH_IP_LB);
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t done = 0;
  if (r < regions) {
    const uint32_t c = P.cnt[r] < KH_IP_CAP ? P.cnt[r] : KH_IP_CAP;
    const uint64_t reg_start = ((uint64_t)r * KH_IP_L + P.ofs) & mask;
    for (uint32_t j = 0; j < c; ++j) {
      const ulonglong2 rec = P.bins[(uint64_t)r * KH_IP_CAP + j];
      const uint64_t home = kh_hash64<HASH>(rec.x, P.seed) & mask;
      const int st = OP == KH_IP_INSERT ? kh_rh_insert_absent<true>(P.T.s, mask, home, rec.x, (uint32_t)rec.y, reg_start)
                                        : kh_rh_erase_one<true>(P.T.s, mask, home, rec.x, reg_start, P.seed.xk);
      if (st == KH_IP_DONE) ++done;
      else if (st == KH_IP_LEAVES_REGION) P.defer[atomicAdd(P.n_defer, 1ull)] = rec;
      else if (st == KH_IP_TOO_FAR) atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u);
    }
  }
  // one atomic per WORKGROUP of 1024 lanes: a same-address atomic per wave (4096 of them for 2^18 regions) serialises in the L2 at
  // ~15 ns each -- 60 of the 77 us this kernel took for 10^4 keys
  __shared__ uint32_t s_done[16];
  for (int off = 32; off > 0; off >>= 1) done += __shfl_down(done, off, 64);
  if ((threadIdx.x & 63) == 0) s_done[threadIdx.x >> 6] = done;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) tot += s_done[w];
    if (tot) atomicAdd(P.n_done, (unsigned long long)tot);
  }
}

// what two binned passes could not place: one lane, unbounded
template <int HASH, int OP>
__global__ void k_ip_serial(KhInplaceParams P) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint64_t n = P.n_dev ? (uint64_t)*P.n_dev : P.n;
  const uint64_t mask = P.T.cap - 1;
  unsigned long long done = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const ulonglong2 rec = P.in_rec[i];
    const uint64_t home = kh_hash64<HASH>(rec.x, P.seed) & mask;
    const int st = OP == KH_IP_INSERT ? kh_rh_insert_absent<false>(P.T.s, mask, home, rec.x, (uint32_t)rec.y, 0)
                                      : kh_rh_erase_one<false>(P.T.s, mask, home, rec.x, 0, P.seed.xk);
    if (st == KH_IP_DONE) ++done;
    else if (st == KH_IP_TOO_FAR) atomicOr(&P.flags[KH_FLAG_PROBE_OVERFLOW], 1u);
  }
  if (done) atomicAdd(P.n_done, done);
}
