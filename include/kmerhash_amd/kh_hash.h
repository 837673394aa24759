// kh_hash.h -- 64-bit hashes of one packed 64-bit k-mer, per lane (host + device).  ONE definition for the kernels
// (kmerhash_amd/csrc/kh_kernels.h) and for the host-side functors of the C++ shim (hashmap.hpp); plain C++11.
//
// murmur3_x86_128 is the hash fsc::hash::murmur3avx64 computes 8 keys at a time with AVX2
// (reference murmurhash3_64_avx.hpp:1083-1169, constants :1511-1521).  The AVX2 code splits every
// 64-bit key into two 32-bit lanes because it has no 64-bit multiply; a CDNA4 lane natively is a
// 32-bit integer ALU, so the per-lane form below *is* the natural layout: one key per lane, k1 = low
// word, k2 = high word, ~45 VALU ops, no cross-lane traffic, and the key batch is read with one
// coalesced 8 B/lane load.  For an 8-byte key there is no 16-byte body block: only the tail (len=8)
// and the finaliser run, and h3 = h4 = seed never receive key material.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KH_HD __host__ __device__ __forceinline__
#else
#define KH_HD inline
#endif

enum { KHH_IDENTITY = 0, KHH_MURMUR3_X86 = 1, KHH_MURMUR3_X64 = 2, KHH_FARM = 3 };

KH_HD uint32_t kh_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
KH_HD uint64_t kh_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
KH_HD uint64_t kh_rotr64(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }

KH_HD uint32_t kh_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
KH_HD uint64_t kh_fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

// MurmurHash3_x86_128(key, len = 8, seed) -> h1 | h2 << 32   (murmurhash3_64_avx.hpp:1083-1169)
KH_HD uint64_t kh_murmur3_x86_128_lo64(uint64_t key, uint32_t seed) {
  const uint32_t c1 = 0x239b961bu, c2 = 0xab0e9789u, c3 = 0x38b34ae5u;
  uint32_t h1 = seed, h2 = seed, h3 = seed, h4 = seed;
  uint32_t k1 = (uint32_t)key, k2 = (uint32_t)(key >> 32);
  k2 *= c2; k2 = kh_rotl32(k2, 16); k2 *= c3; h2 ^= k2;
  k1 *= c1; k1 = kh_rotl32(k1, 15); k1 *= c2; h1 ^= k1;
  h1 ^= 8u; h2 ^= 8u; h3 ^= 8u; h4 ^= 8u;
  h1 += h2; h1 += h3; h1 += h4; h2 += h1; h3 += h1; h4 += h1;
  h1 = kh_fmix32(h1); h2 = kh_fmix32(h2); h3 = kh_fmix32(h3); h4 = kh_fmix32(h4);
  h1 += h2; h1 += h3; h1 += h4; h2 += h1;
  return (uint64_t)h1 | ((uint64_t)h2 << 32);
}

// MurmurHash3_x64_128(key, len = 8, seed)[0]   (fsc::hash::murmur, hash_new.hpp:206-235)
KH_HD uint64_t kh_murmur3_x64_128_h0(uint64_t key, uint32_t seed) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  uint64_t k1 = key;
  k1 *= c1; k1 = kh_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  h1 ^= 8u; h2 ^= 8u;
  h1 += h2; h2 += h1;
  h1 = kh_fmix64(h1); h2 = kh_fmix64(h2);
  h1 += h2;
  return h1;
}

// farmhash util::Hash64WithSeed(key, 8, seed): HashLen0to16 branch + Hash64WithSeeds(k2, seed)
// (fsc::hash::farm, hash_new.hpp:309-328; published google/farmhash algorithm; parity unpinned)
KH_HD uint64_t kh_farm_hashlen16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul; a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul; b ^= (b >> 47);
  return b * mul;
}
KH_HD uint64_t kh_farm64_seed(uint64_t key, uint64_t seed) {
  const uint64_t k2 = 0x9ae16a3b2f90404fULL;
  const uint64_t mul = k2 + 16;
  uint64_t a = key + k2, b = key;
  uint64_t c = kh_rotr64(b, 37) * mul + a;
  uint64_t d = (kh_rotr64(a, 25) + b) * mul;
  uint64_t h = kh_farm_hashlen16(c, d, mul);
  return kh_farm_hashlen16(h - k2, seed, 0x9ddfea08eb382d69ULL);
}

template <int HASH>
KH_HD uint64_t kh_hash64(uint64_t key, uint64_t seed) {
  if (HASH == KHH_IDENTITY) return key;
  else if (HASH == KHH_MURMUR3_X86) return kh_murmur3_x86_128_lo64(key, (uint32_t)seed);
  else if (HASH == KHH_MURMUR3_X64) return kh_murmur3_x64_128_h0(key, (uint32_t)seed);
  else return kh_farm64_seed(key, seed);
}

// ---- key transform in front of the hash and inside key equality: fsc::TransformedHash<Key, Hash, PreTransform> with
// PreTransform = bliss::kmer::transform::lex_less (hash_new.hpp:387-1134; "bimolecule" tables: a k-mer and its reverse
// complement are ONE key, test/unit/test_hashmap_robinhood_doubling.cpp:560-626).  xk = 0: identity; xk = k (1..32): the key is a
// 2-bit packed DNA k-mer (first base most significant, A0 C1 G2 T3) and stands for min(key, reverse complement).
struct KhSeed {
  uint64_t s; uint32_t xk;
};
KH_HD uint64_t kh_revcomp(uint64_t x, uint32_t k) {
  x = ~x;                                                                   // complement: 3 - base
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  x = (x >> 32) | (x << 32);
  return x >> (64 - 2 * k);
}
KH_HD uint64_t kh_xf(uint64_t key, uint32_t xk) {
  if (!xk) return key;
  const uint64_t rc = kh_revcomp(key, xk);
  return rc < key ? rc : key;
}
KH_HD bool kh_keq(uint64_t a, uint64_t b, uint32_t xk) { return xk ? kh_xf(a, xk) == kh_xf(b, xk) : a == b; }
template <int HASH>
KH_HD uint64_t kh_hash64(uint64_t key, KhSeed hs) { return kh_hash64<HASH>(kh_xf(key, hs.xk), hs.s); }
