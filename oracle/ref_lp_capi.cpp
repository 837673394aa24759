// ref_lp_capi.cpp -- C-ABI driver over the REAL reference linear-probing table.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
// kmerhash/hashmap_linearprobe.hpp from the reference tree where it lies (-I/root/reference/include,
// see oracle/Makefile) and is compiled into oracle/_ref/libref_lp.so (git-ignored build output).
// hashmap_linearprobe.hpp needs nothing outside the reference tree (no stand-in headers are written);
// -mlzcnt is required because the non-LZCNT branch of math_utils.hpp:78 does not compile.
//
// The hash functor handed to the reference template is the oracle's murmur3/identity/farm restatement
// (the reference's own scalar functors live in hash_new.hpp, which hard-includes third-party sources
// that are absent here; the murmur3 restatement is pinned separately against smhasher).
#include <cassert>
#include <iostream>
#include <iomanip>
#include <limits>
#include <tuple>
#include <stdexcept>
#include <string>
#include <cstdint>
#include <chrono>

#include "kmerhash/hashmap_linearprobe.hpp"
#include "kh_oracle.hpp"

namespace {
int g_hash_id = 1;
uint64_t g_seed = 43;
struct OraHash {
  int id; uint64_t seed;
  OraHash() : id(g_hash_id), seed(g_seed) {}
  uint64_t operator()(uint64_t const& k) const { return kh_oracle::hash_u64(id, k, seed); }
};
typedef ::fsc::hashmap_linearprobe_doubling<uint64_t, uint32_t, OraHash> RefBase;
struct RefLP : public RefBase {
  RefLP(size_t cap, float mn, float mx) : RefBase(cap, mn, mx) {}
  size_t cap() const { return this->buckets; }
  size_t maxload() const { return this->max_load; }
  size_t minload() const { return this->min_load; }
  void export_info(uint8_t* out) const { for (size_t i = 0; i < this->info_container.size(); ++i) out[i] = this->info_container[i].info; }
  void export_slots(uint64_t* k, uint32_t* v) const {
    for (size_t i = 0; i < this->container.size(); ++i) { k[i] = this->container[i].first; v[i] = this->container[i].second; }
  }
};
}  // namespace

extern "C" {

void* ref_lp_create(uint64_t capacity, float min_lf, float max_lf, int hash_id, uint64_t seed) {
  g_hash_id = hash_id; g_seed = seed;
  return new RefLP(capacity, min_lf, max_lf);
}
void ref_lp_destroy(void* h) { delete static_cast<RefLP*>(h); }
uint64_t ref_lp_size(void* h) { return static_cast<RefLP*>(h)->size(); }
uint64_t ref_lp_capacity(void* h) { return static_cast<RefLP*>(h)->cap(); }
uint64_t ref_lp_max_load(void* h) { return static_cast<RefLP*>(h)->maxload(); }
uint64_t ref_lp_min_load(void* h) { return static_cast<RefLP*>(h)->minload(); }
void ref_lp_set_min_load_factor(void* h, float f) { static_cast<RefLP*>(h)->set_min_load_factor(f); }
void ref_lp_set_max_load_factor(void* h, float f) { static_cast<RefLP*>(h)->set_max_load_factor(f); }
void ref_lp_clear(void* h) { static_cast<RefLP*>(h)->clear(); }
void ref_lp_reserve(void* h, uint64_t n) { static_cast<RefLP*>(h)->reserve(n); }
int ref_lp_rehash(void* h, uint64_t b) {
  try { static_cast<RefLP*>(h)->rehash(b); } catch (std::logic_error&) { return 1; }
  return 0;
}
// insert(vector<value_type> const&)  (hashmap_linearprobe.hpp:549)
int64_t ref_lp_insert(void* h, const uint64_t* keys, const uint32_t* vals, uint64_t n) {
  RefLP* t = static_cast<RefLP*>(h);
  std::vector<std::pair<uint64_t, uint32_t> > in(n);
  for (uint64_t i = 0; i < n; ++i) in[i] = std::make_pair(keys[i], vals[i]);
  size_t before = t->size();
  try { t->insert(in); } catch (std::logic_error&) { return -1; }
  return int64_t(t->size() - before);
}
int ref_lp_insert_one(void* h, uint64_t key, uint32_t val) { return static_cast<RefLP*>(h)->insert(key, val).second ? 1 : 0; }
void ref_lp_update_one(void* h, uint64_t key, uint32_t val) { static_cast<RefLP*>(h)->update(key, val); }
// count(Iter,Iter) over keys (:664)
void ref_lp_count(void* h, const uint64_t* keys, uint64_t n, uint8_t* out) {
  RefLP* t = static_cast<RefLP*>(h);
  std::vector<size_t> c = t->count(keys, keys + n);
  for (uint64_t i = 0; i < n; ++i) out[i] = uint8_t(c[i]);
}
// find(Iter,Iter) over keys (:854): compacted hits in query order
uint64_t ref_lp_find_compact(void* h, const uint64_t* keys, uint64_t n, uint64_t* out_keys, uint32_t* out_vals) {
  RefLP* t = static_cast<RefLP*>(h);
  std::vector<std::pair<uint64_t, uint32_t> > r = t->find(keys, keys + n);
  for (size_t i = 0; i < r.size(); ++i) { out_keys[i] = r[i].first; out_vals[i] = r[i].second; }
  return r.size();
}
// erase(Iter,Iter) (:1042)
int64_t ref_lp_erase(void* h, const uint64_t* keys, uint64_t n) {
  try { return int64_t(static_cast<RefLP*>(h)->erase(keys, keys + n)); } catch (std::logic_error&) { return -1; }
}
int ref_lp_erase_one(void* h, uint64_t key) { return int(static_cast<RefLP*>(h)->erase(key)); }
void ref_lp_export_info(void* h, uint8_t* out) { static_cast<RefLP*>(h)->export_info(out); }
void ref_lp_export_slots(void* h, uint64_t* k, uint32_t* v) { static_cast<RefLP*>(h)->export_slots(k, v); }
uint64_t ref_lp_to_vector(void* h, uint64_t* keys, uint32_t* vals) {
  std::vector<std::pair<uint64_t, uint32_t> > r = static_cast<RefLP*>(h)->to_vector();
  for (size_t i = 0; i < r.size(); ++i) { keys[i] = r[i].first; vals[i] = r[i].second; }
  return r.size();
}
// timed phases for bench.py's cpu_baseline leg (kind "reference")
double ref_lp_timed_insert(void* h, const uint64_t* keys, const uint32_t* vals, uint64_t n) {
  RefLP* t = static_cast<RefLP*>(h);
  std::vector<std::pair<uint64_t, uint32_t> > in(n);
  for (uint64_t i = 0; i < n; ++i) in[i] = std::make_pair(keys[i], vals[i]);
  auto t0 = std::chrono::steady_clock::now();
  t->insert(in);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}
double ref_lp_timed_count(void* h, const uint64_t* keys, uint64_t n, uint64_t* n_found) {
  RefLP* t = static_cast<RefLP*>(h);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<size_t> c = t->count(keys, keys + n);
  auto t1 = std::chrono::steady_clock::now();
  uint64_t s = 0; for (size_t i = 0; i < c.size(); ++i) s += c[i];
  *n_found = s;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
