// ref_hll_capi.cpp -- C-ABI driver over the REAL reference HyperLogLog (TEST INFRASTRUCTURE ONLY).
// Contains no reference code: #includes kmerhash/hyperloglog64.hpp (and, through it, kmerhash/mem_utils.hpp) from the
// reference tree where it lies; without -DUSE_MPI the header needs nothing outside that tree.  Also exercises the
// reference's io_utils.hpp serialize_vector/deserialize_vector (binary dump format `size_t elsize, size_t n, raw`).
#include <cassert>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "kmerhash/hyperloglog64.hpp"
#include "kmerhash/io_utils.hpp"
#include "kh_oracle.hpp"

namespace {
int g_hash_id = 1;
uint64_t g_seed = 43;
struct OraHash {
  int id; uint64_t seed;
  OraHash() : id(g_hash_id), seed(g_seed) {}
  uint64_t operator()(uint64_t const& k) const { return kh_oracle::hash_u64(id, k, seed); }
};
typedef hyperloglog64<uint64_t, OraHash, 12> HLL12;
struct RefHLL : public HLL12 {
  explicit RefHLL(uint8_t ign) : HLL12(ign) {}
  void regs(uint8_t* out) const { for (size_t i = 0; i < this->registers.size(); ++i) out[i] = this->registers[i]; }
};
}  // namespace

extern "C" {
void* ref_hll_create(uint32_t ignore_msb, int hash_id, uint64_t seed) { g_hash_id = hash_id; g_seed = seed; return new RefHLL(uint8_t(ignore_msb)); }
void ref_hll_destroy(void* h) { delete static_cast<RefHLL*>(h); }
void ref_hll_update(void* h, const uint64_t* keys, uint64_t n) { static_cast<RefHLL*>(h)->update(keys, n); }
void ref_hll_update_via_hashval(void* h, const uint64_t* hv, uint64_t n) { static_cast<RefHLL*>(h)->update_via_hashval(hv, n); }
void ref_hll_merge(void* h, void* o) { static_cast<RefHLL*>(h)->merge(*static_cast<RefHLL*>(o)); }
void ref_hll_clear(void* h) { static_cast<RefHLL*>(h)->clear(); }
double ref_hll_estimate(void* h) { return static_cast<RefHLL*>(h)->estimate(); }
void ref_hll_registers(void* h, uint8_t* out) { static_cast<RefHLL*>(h)->regs(out); }

// io_utils.hpp:57-103
void ref_serialize_pairs(const uint64_t* keys, const uint32_t* vals, uint64_t n, const char* path) {
  std::vector<std::pair<uint64_t, uint32_t> > v(n);
  for (uint64_t i = 0; i < n; ++i) v[i] = std::make_pair(keys[i], vals[i]);
  serialize_vector(v, std::string(path));
}
// returns n, or -1 on the reference's logic_error (element size mismatch)
int64_t ref_deserialize_pairs(const char* path, uint64_t* keys, uint32_t* vals, uint64_t cap) {
  try {
    std::vector<std::pair<uint64_t, uint32_t> > v = deserialize_vector<std::pair<uint64_t, uint32_t> >(std::string(path));
    for (uint64_t i = 0; i < v.size() && i < cap; ++i) { keys[i] = v[i].first; vals[i] = v[i].second; }
    return int64_t(v.size());
  } catch (std::logic_error&) { return -1; }
}
void ref_serialize_u64(const uint64_t* keys, uint64_t n, const char* path) {
  std::vector<uint64_t> v(keys, keys + n);
  serialize_vector(v, std::string(path));
}
}  // extern "C"
