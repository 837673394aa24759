// C-ABI over the CPU oracle (TEST INFRASTRUCTURE ONLY -- see kh_oracle.hpp header).
// Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include "kh_oracle.hpp"
#include <algorithm>
#include <chrono>

using namespace kh_oracle;

namespace {
struct Table {
  int kind;  // 0 = RH, 1 = LP
  RobinHood* rh;
  LinearProbe* lp;
};
}  // namespace

extern "C" {

uint64_t ora_hash_u64(int hash_id, uint64_t key, uint64_t seed) { return hash_u64(hash_id, key, seed); }

void ora_hash_batch(int hash_id, uint64_t seed, const uint64_t* keys, uint64_t n, uint64_t* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = hash_u64(hash_id, keys[i], seed);
}

// general-length murmur3 (pins the restatement against smhasher for every tail length)
void ora_murmur3_x86_128(const void* key, int len, uint32_t seed, uint32_t* out4) { murmur3_x86_128(key, len, seed, out4); }
void ora_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t* out2) { murmur3_x64_128(key, len, seed, out2); }

uint64_t ora_next_power_of_2(uint64_t x) { return next_power_of_2(x); }
uint64_t ora_load_threshold(uint64_t buckets, float lf) { return load_threshold(buckets, lf); }

void* ora_create(int kind, uint64_t capacity, float min_lf, float max_lf, int hash_id, uint64_t seed) {
  Table* t = new Table();
  t->kind = kind; t->rh = nullptr; t->lp = nullptr;
  if (kind == 0) t->rh = new RobinHood(capacity, min_lf, max_lf, hash_id, seed);
  else t->lp = new LinearProbe(capacity, min_lf, max_lf, hash_id, seed);
  return t;
}
void ora_set_key_transform(void* h, uint32_t xk) {      // only on an empty table
  Table* t = static_cast<Table*>(h);
  if (t->kind == 0) t->rh->xk = xk; else t->lp->xk = xk;
}
uint64_t ora_pre_transform(uint64_t key, uint32_t xk) { return pre_transform(key, xk); }
void ora_destroy(void* h) {
  Table* t = static_cast<Table*>(h);
  delete t->rh; delete t->lp; delete t;
}
#define DISPATCH(expr_rh, expr_lp) (t->kind == 0 ? (expr_rh) : (expr_lp))

uint64_t ora_size(void* h) { Table* t = static_cast<Table*>(h); return DISPATCH(t->rh->size(), t->lp->size()); }
uint64_t ora_capacity(void* h) { Table* t = static_cast<Table*>(h); return DISPATCH(t->rh->capacity(), t->lp->capacity()); }
uint64_t ora_max_load(void* h) { Table* t = static_cast<Table*>(h); return DISPATCH(t->rh->max_load, t->lp->max_load); }
uint64_t ora_min_load(void* h) { Table* t = static_cast<Table*>(h); return DISPATCH(t->rh->min_load, t->lp->min_load); }
void ora_set_min_load_factor(void* h, float f) { Table* t = static_cast<Table*>(h); if (t->kind == 0) t->rh->set_min_load_factor(f); else t->lp->set_min_load_factor(f); }
void ora_set_max_load_factor(void* h, float f) { Table* t = static_cast<Table*>(h); if (t->kind == 0) t->rh->set_max_load_factor(f); else t->lp->set_max_load_factor(f); }
void ora_clear(void* h) { Table* t = static_cast<Table*>(h); if (t->kind == 0) t->rh->clear(); else t->lp->clear(); }
void ora_reserve(void* h, uint64_t n) { Table* t = static_cast<Table*>(h); if (t->kind == 0) t->rh->reserve(n); else t->lp->reserve(n); }
// returns 0 ok, 1 = logic_error thrown (LP full)
int ora_rehash(void* h, uint64_t b) {
  Table* t = static_cast<Table*>(h);
  try { if (t->kind == 0) t->rh->rehash(b); else t->lp->rehash(b); } catch (std::logic_error&) { return 1; }
  return 0;
}
int ora_probe_overflow(void* h) { Table* t = static_cast<Table*>(h); return t->kind == 0 ? (t->rh->probe_overflow ? 1 : 0) : 0; }

// batch insert == insert(Iter,Iter) / insert(vector const&); returns #inserted, -1 on LP logic_error
int64_t ora_insert(void* h, const uint64_t* keys, const uint32_t* vals, uint64_t n) {
  Table* t = static_cast<Table*>(h);
  try { return int64_t(DISPATCH(t->rh->insert_batch(keys, vals, n), t->lp->insert_batch(keys, vals, n))); }
  catch (std::logic_error&) { return -1; }
}
// single insert(k,v): returns 1 inserted / 0 duplicate
int ora_insert_one(void* h, uint64_t key, uint32_t val) {
  Table* t = static_cast<Table*>(h);
  return DISPATCH(t->rh->insert(key, val).second, t->lp->insert(key, val).second) ? 1 : 0;
}
void ora_update_one(void* h, uint64_t key, uint32_t val) {
  Table* t = static_cast<Table*>(h);
  if (t->kind == 0) t->rh->update(key, val); else t->lp->update(key, val);
}
// count(Iter,Iter): 0/1 per query, input order
void ora_count(void* h, const uint64_t* keys, uint64_t n, uint8_t* out) {
  Table* t = static_cast<Table*>(h);
  for (uint64_t i = 0; i < n; ++i) out[i] = uint8_t(DISPATCH(t->rh->count(keys[i]), t->lp->count(keys[i])));
}
// per-query find: found flag + value (value untouched on miss)
void ora_find(void* h, const uint64_t* keys, uint64_t n, uint32_t* out_vals, uint8_t* out_found) {
  Table* t = static_cast<Table*>(h);
  const size_t none = std::numeric_limits<size_t>::max();
  for (uint64_t i = 0; i < n; ++i) {
    size_t p = DISPATCH(t->rh->find_pos(keys[i]), t->lp->find_pos(keys[i]));
    bool f = (t->kind == 0) ? (p < t->rh->buckets) : (p != none);
    out_found[i] = f ? 1 : 0;
    if (f) out_vals[i] = DISPATCH(t->rh->container[p].second, t->lp->container[p].second);
  }
}
// find(Iter,Iter): compacted (key,value) hits in query order; returns number of hits
uint64_t ora_find_compact(void* h, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals) {
  Table* t = static_cast<Table*>(h);
  const size_t none = std::numeric_limits<size_t>::max();
  uint64_t m = 0;
  for (uint64_t i = 0; i < n; ++i) {
    size_t p = DISPATCH(t->rh->find_pos(keys[i]), t->lp->find_pos(keys[i]));
    bool f = (t->kind == 0) ? (p < t->rh->buckets) : (p != none);
    if (f) {
      out_keys[m] = DISPATCH(t->rh->container[p].first, t->lp->container[p].first);
      out_vals[m] = DISPATCH(t->rh->container[p].second, t->lp->container[p].second);
      ++m;
    }
  }
  return m;
}
// erase(Iter,Iter); returns #erased, -1 on LP logic_error (shrinking rehash with no room)
int64_t ora_erase(void* h, const uint64_t* keys, uint64_t n) {
  Table* t = static_cast<Table*>(h);
  try { return int64_t(DISPATCH(t->rh->erase_batch(keys, n), t->lp->erase_batch(keys, n))); }
  catch (std::logic_error&) { return -1; }
}
// erase(key) (single-key form: may halve)
int ora_erase_one(void* h, uint64_t key) {
  Table* t = static_cast<Table*>(h);
  return int(DISPATCH(t->rh->erase(key), t->lp->erase(key)));
}
void ora_export_info(void* h, uint8_t* out) {
  Table* t = static_cast<Table*>(h);
  if (t->kind == 0) std::copy(t->rh->info.begin(), t->rh->info.end(), out);
  else std::copy(t->lp->info.begin(), t->lp->info.end(), out);
}
void ora_export_slots(void* h, uint64_t* keys, uint32_t* vals) {
  Table* t = static_cast<Table*>(h);
  size_t n = DISPATCH(t->rh->buckets, t->lp->buckets);
  for (size_t i = 0; i < n; ++i) {
    keys[i] = DISPATCH(t->rh->container[i].first, t->lp->container[i].first);
    vals[i] = DISPATCH(t->rh->container[i].second, t->lp->container[i].second);
  }
}
// to_vector(): occupied (key,value) pairs in slot order; returns count
uint64_t ora_to_vector(void* h, uint64_t* keys, uint32_t* vals) {
  Table* t = static_cast<Table*>(h);
  size_t n = DISPATCH(t->rh->buckets, t->lp->buckets);
  uint64_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    bool occ = (t->kind == 0) ? (t->rh->info[i] >= RobinHood::NORMAL) : LinearProbe::is_normal(t->lp->info[i]);
    if (occ) {
      keys[m] = DISPATCH(t->rh->container[i].first, t->lp->container[i].first);
      vals[m] = DISPATCH(t->rh->container[i].second, t->lp->container[i].second);
      ++m;
    }
  }
  return m;
}
// Robin Hood displacement histogram (REPROBE_STAT oracle): out[d] = #slots with distance d
void ora_displacement_histogram(void* h, uint64_t* out128) {
  Table* t = static_cast<Table*>(h);
  for (int i = 0; i < 128; ++i) out128[i] = 0;
  if (t->kind != 0) return;
  for (size_t i = 0; i < t->rh->buckets; ++i)
    if (t->rh->info[i] >= RobinHood::NORMAL) ++out128[t->rh->info[i] & 0x7F];
}

// timed phases for bench.py's cpu_baseline leg (kind "port"): seconds spent in the batch call
double ora_timed_insert(void* h, const uint64_t* keys, const uint32_t* vals, uint64_t n) {
  auto t0 = std::chrono::steady_clock::now();
  ora_insert(h, keys, vals, n);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}
double ora_timed_find(void* h, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals, uint64_t* n_found) {
  auto t0 = std::chrono::steady_clock::now();
  *n_found = ora_find_compact(h, keys, n, out_keys, out_vals);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---- HyperLogLog restatement
void* ora_hll_create(uint32_t precision, uint32_t ignore_msb, int hash_id, uint64_t seed) { return new HyperLogLog64(precision, ignore_msb, hash_id, seed); }
void ora_hll_destroy(void* h) { delete static_cast<HyperLogLog64*>(h); }
void ora_hll_update(void* h, const uint64_t* keys, uint64_t n) { HyperLogLog64* p = static_cast<HyperLogLog64*>(h); for (uint64_t i = 0; i < n; ++i) p->update(keys[i]); }
void ora_hll_update_via_hashval(void* h, const uint64_t* hv, uint64_t n) { HyperLogLog64* p = static_cast<HyperLogLog64*>(h); for (uint64_t i = 0; i < n; ++i) p->update_via_hashval(hv[i]); }
void ora_hll_merge(void* h, void* o) { static_cast<HyperLogLog64*>(h)->merge(*static_cast<HyperLogLog64*>(o)); }
void ora_hll_clear(void* h) { static_cast<HyperLogLog64*>(h)->clear(); }
double ora_hll_estimate(void* h) { return static_cast<HyperLogLog64*>(h)->estimate(); }
void ora_hll_registers(void* h, uint8_t* out) { HyperLogLog64* p = static_cast<HyperLogLog64*>(h); std::copy(p->registers.begin(), p->registers.end(), out); }

}  // extern "C"
