// smhasher_capi.cpp -- thin C-ABI over the smhasher MurmurHash3 copy that ships with scikit-learn
// in this image (sklearn/utils/src/MurmurHash3.cpp, public domain, A. Appleby).  smhasher is the
// third-party dependency the reference's scalar murmur functors call (hash_new.hpp:83,218-233; pinned
// version not recorded: the kmerind submodule tracks a branch).  TEST INFRASTRUCTURE ONLY: used by
// tests/golden/make_golden.py to generate tests/golden/murmur3_kat.npz.
#include <cstdint>
#include "MurmurHash3.h"
extern "C" {
void smh_x86_128(const void* key, int len, uint32_t seed, void* out16) { MurmurHash3_x86_128(key, len, seed, out16); }
void smh_x64_128(const void* key, int len, uint32_t seed, void* out16) { MurmurHash3_x64_128(key, len, seed, out16); }
}
