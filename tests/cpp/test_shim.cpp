// C++ side of the drop-in boundary: the reference's own test strategy (test/unit/test_hashmap_robinhood_doubling.cpp
// :97-334, test_hashmap_linearprobe_doubling.cpp:84-195) written against the shim: differential against
// std::unordered_map::emplace (first value wins), through the reference's member names.  The map type is passed
// as a 5-parameter template-template exactly as BenchmarkHashTables.cpp:1037-1048 does.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <unordered_map>
#include <list>
#include <vector>

#include "kmerhash/hashmap_robinhood.hpp"
#include "kmerhash/hashmap_linearprobe.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

// a Kmer-like key: 8-byte trivially copyable wrapper of one word (bliss::common::Kmer<31,DNA,uint64_t> shape)
struct Kmer31 {
  uint64_t data;
  Kmer31() : data(0) {}
  explicit Kmer31(uint64_t d) : data(d) {}
  Kmer31(Kmer31 const& o) : data(o.data) {}                      // user-declared copy operations, like bliss::common::Kmer
  Kmer31& operator=(Kmer31 const& o) { data = o.data; return *this; }
  bool operator==(Kmer31 const& o) const { return data == o.data; }
  bool operator<(Kmer31 const& o) const { return data < o.data; }
};
namespace std { template <> struct hash<Kmer31> { size_t operator()(Kmer31 const& k) const { return k.data; } }; }

// the benchmark's own equality functor (BenchmarkHashTables.cpp:169-180): a global-namespace template, not std::equal_to
template <class T>
struct equal_to {
  using result_type = bool;
  using first_argument_type = T;
  using second_argument_type = T;
  inline constexpr bool operator()(T const& lhs, T const& rhs) const { return lhs == rhs; }
};
struct never_equal { bool operator()(uint64_t const&, uint64_t const&) const { return false; } };

template <template <typename, typename, typename, typename, typename> class MAP, typename Key, typename Hash>
void differential(const char* name, size_t n, bool is_rh) {
  using Map = MAP<Key, uint32_t, Hash, ::equal_to<Key>, ::std::allocator<::std::pair<Key, uint32_t> > >;
  std::default_random_engine gen(17);
  std::uniform_int_distribution<uint64_t> dist(2, (uint64_t(1) << 62) - 2);
  std::vector<std::pair<Key, uint32_t> > input;
  std::unordered_map<uint64_t, uint32_t> gold;
  for (size_t i = 0; i < n; ++i) {
    uint64_t k = dist(gen);
    if (i % 3 == 0 && !input.empty()) k = kmerhash_amd::detail::key_bits(input[gen() % input.size()].first);
    input.push_back(std::make_pair(Key(k), uint32_t(i)));
    gold.emplace(k, uint32_t(i));
  }
  Map map;
  map.set_min_load_factor(0.35f);
  map.set_max_load_factor(0.8f);
  map.insert(input);
  CHECK(map.size() == gold.size());
  // sorted to_vector() == sorted gold
  auto v = map.to_vector();
  CHECK(v.size() == gold.size());
  for (auto& kv : v) { auto it = gold.find(kmerhash_amd::detail::key_bits(kv.first)); CHECK(it != gold.end() && it->second == kv.second); }
  // iteration over begin()/end()
  size_t cnt = 0;
  for (auto it = map.begin(); it != map.end(); ++it) { CHECK(gold.count(kmerhash_amd::detail::key_bits(it->first)) == 1); ++cnt; }
  CHECK(cnt == gold.size());
  // single-key count / find, batch count over pairs and over keys, batch find
  size_t q = std::min<size_t>(n / 2 + 1, input.size());
  std::vector<Key> qk;
  for (size_t i = 0; i < q; ++i) qk.push_back(input[i].first);
  for (size_t i = 0; i < 50; ++i) qk.push_back(Key(dist(gen) | (uint64_t(1) << 63)));   // misses
  auto counts = map.count(qk.begin(), qk.end());
  auto counts2 = map.count(input.begin(), input.begin() + q);
  CHECK(counts.size() == qk.size() && counts2.size() == q);
  for (size_t i = 0; i < qk.size(); ++i) CHECK(counts[i] == gold.count(kmerhash_amd::detail::key_bits(qk[i])));
  for (size_t i = 0; i < q; ++i) CHECK(counts2[i] == 1);
  auto found = map.find(qk.begin(), qk.end());
  CHECK(found.size() == q);
  for (size_t i = 0; i < q; ++i) { CHECK(found[i].first == qk[i]); CHECK(found[i].second == gold[kmerhash_amd::detail::key_bits(qk[i])]); }
  // the same queries through raw pointers, const_iterators (borrowed as they lie) and a non-contiguous container (gathered)
  {
    const Key* p0 = qk.data();
    auto f2 = map.find(p0, p0 + qk.size());
    auto c2 = map.count(qk.cbegin(), qk.cend());
    std::list<Key> lk(qk.begin(), qk.end());
    auto f3 = map.find(lk.begin(), lk.end());
    auto c4 = map.count(lk.begin(), lk.end());
    CHECK(f2.size() == found.size() && f3.size() == found.size() && c2 == counts && c4 == counts);
    for (size_t i = 0; i < found.size(); ++i) { CHECK(f2[i] == found[i]); CHECK(f3[i] == found[i]); }
    std::vector<Key> none;
    CHECK(map.find(none.begin(), none.end()).empty() && map.count(none.data(), none.data()).empty());
  }
  CHECK(map.count(qk[0]) == 1 && map.count(qk.back()) == 0);
  CHECK(map.find(qk[0]) != map.end() && map.find(qk[0])->second == gold[kmerhash_amd::detail::key_bits(qk[0])]);
  CHECK(map.find(qk.back()) == map.end());
  // insert(key,val) duplicate keeps the first value; update overwrites
  auto r = map.insert(qk[0], 4242u);
  CHECK(!r.second && r.first->second == gold[kmerhash_amd::detail::key_bits(qk[0])]);
  CHECK(map.update(qk[0], 4242u)->second == 4242u);
  // erase first half: size equality, erased keys gone, kept keys present
  std::vector<Key> uniq;
  for (auto& kv : gold) uniq.push_back(Key(kv.first));
  std::sort(uniq.begin(), uniq.end());
  size_t half = uniq.size() / 2;
  CHECK(map.erase(uniq.begin(), uniq.begin() + half) == half);
  CHECK(map.size() == uniq.size() - half);
  auto c3 = map.count(uniq.begin(), uniq.end());
  for (size_t i = 0; i < uniq.size(); ++i) CHECK(c3[i] == (i < half ? 0u : 1u));
  CHECK(map.erase(uniq[half]) == 1 && map.erase(uniq[half]) == 0);
  if (is_rh) CHECK(map.capacity() >= map.size());
  map.clear();
  CHECK(map.size() == 0 && map.begin() == map.end());
  std::printf("%s ok (n=%zu, distinct=%zu)\n", name, n, gold.size());
}

// The reference's bimolecule-mode test (test/unit/test_hashmap_robinhood_doubling.cpp:560-626 with Transform = lex_less):
//   THash = fsc::TransformedHash<Kmer, Hash, Transform>, Equal1 = fsc::TransformedComparator<Kmer, std::equal_to, Transform>,
//   gold = std::unordered_map<Kmer, uint32_t, THash, Equal1>; sorted to_vector() of both must be equal: a k-mer and its
//   reverse complement are one key, the bits stored are those of the first occurrence.
template <template <typename, typename, typename, typename, typename> class MAP, unsigned K, template <typename> class Hash>
void bimolecule(const char* name, size_t n) {
  using Kmer = kmerhash_amd::dna_kmer<K>;
  using THash = ::fsc::TransformedHash<Kmer, Hash, ::bliss::kmer::transform::lex_less>;
  using Equal1 = ::fsc::TransformedComparator<Kmer, std::equal_to, ::bliss::kmer::transform::lex_less>;
  using Map = MAP<Kmer, uint32_t, THash, Equal1, ::std::allocator<::std::pair<Kmer, uint32_t> > >;
  static_assert(::fsc::hash::batch_traits<THash>::get_batch_size(0) == THash::batch_size && THash::batch_size == 16, "batch_traits sees TransformedHash::batch_size");
  static_assert(::fsc::hash::batch_traits<::fsc::hash::murmur3avx64<Kmer> >::get_batch_size(0) == 8, "murmur3avx64 hashes 8 keys per AVX2 batch in the reference");
  static_assert(::fsc::hash::batch_traits<std::equal_to<Kmer> >::get_batch_size(0) == 1, "no batch_size member: 1");
  std::default_random_engine gen(K);
  const uint64_t top = K < 32 ? ((uint64_t(1) << (2 * K)) - 1) : ~uint64_t(0);
  std::uniform_int_distribution<uint64_t> dist(0, top);
  std::vector<std::pair<Kmer, uint32_t> > input;
  std::unordered_map<Kmer, uint32_t, THash, Equal1> gold;
  for (size_t i = 0; i < n; ++i) {
    Kmer k(dist(gen));
    if (i % 3 == 1) k = input[gen() % input.size()].first.reverse_complement();      // the other strand of an earlier k-mer
    if (i % 7 == 3) k = input[gen() % input.size()].first;                           // an exact repeat
    input.push_back(std::make_pair(k, uint32_t(i)));
    gold.emplace(k, uint32_t(i));
  }
  Map map;
  map.set_min_load_factor(0.35f);
  map.set_max_load_factor(0.8f);
  map.insert(input);
  auto cmp = [](std::pair<Kmer, uint32_t> const& x, std::pair<Kmer, uint32_t> const& y) { return x.first == y.first ? x.second < y.second : x.first < y.first; };
  std::vector<std::pair<Kmer, uint32_t> > test_vals = map.to_vector(), gold_vals(gold.begin(), gold.end());
  CHECK(test_vals.size() == gold_vals.size() && gold_vals.size() < n);
  std::sort(test_vals.begin(), test_vals.end(), cmp);
  std::sort(gold_vals.begin(), gold_vals.end(), cmp);
  CHECK(std::equal(test_vals.begin(), test_vals.end(), gold_vals.begin()));
  // either strand finds the stored pair
  std::vector<Kmer> q;
  for (size_t i = 0; i < std::min<size_t>(n, 3000); ++i) q.push_back(i % 2 ? input[i].first.reverse_complement() : input[i].first);
  auto found = map.find(q.begin(), q.end());
  auto counts = map.count(q.begin(), q.end());
  CHECK(found.size() == q.size());
  for (size_t i = 0; i < q.size(); ++i) { auto it = gold.find(q[i]); CHECK(counts[i] == 1 && it != gold.end() && found[i].first == it->first && found[i].second == it->second); }
  CHECK(map.find(q[1]) != map.end() && map.find(q[1])->first == gold.find(q[1])->first);
  // the batch form of the functor (device) == its single-key form (host)
  THash th;
  std::vector<uint64_t> hv(q.size());
  th(q.data(), q.size(), hv.data());
  for (size_t i = 0; i < q.size(); ++i) CHECK(hv[i] == th(q[i]) && hv[i] == th(q[i].reverse_complement()));
  // erase by the other strand
  std::vector<Kmer> e;
  for (size_t i = 0; i < 500; ++i) e.push_back(input[i].first.reverse_complement());
  size_t ne = map.erase(e.begin(), e.end());
  for (auto& k : e) gold.erase(k);
  CHECK(map.size() == gold.size() && ne > 0);
  std::printf("%s ok (n=%zu, distinct under lex_less=%zu)\n", name, n, gold.size() + ne);
}

// the reference benchmark's find phase as it is written: single-key const calls in a loop (BenchmarkHashTables.cpp:1134-1138).
// After 64 such calls without a mutation the shim probes a host copy of the slot array: 10^6 calls must take well under a second
// and agree with the batch form; a mutation drops the copy
template <template <typename, typename, typename, typename, typename> class MAP>
void single_key_loop(const char* name) {
  using Map = MAP<uint64_t, uint32_t, fsc::hash::murmur3avx64<uint64_t>, ::equal_to<uint64_t>, ::std::allocator<::std::pair<uint64_t, uint32_t> > >;
  const size_t n = 2000000, nq = 1000000;
  std::vector<std::pair<uint64_t, uint32_t> > input(n);
  uint64_t s = 99;
  auto next = [&s]() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  for (size_t i = 0; i < n; ++i) input[i] = std::make_pair(next() >> 2, uint32_t(i));
  Map map;
  map.set_min_load_factor(0.35f); map.set_max_load_factor(0.8f);
  map.insert(input);
  std::vector<uint64_t> q(nq);
  for (size_t i = 0; i < nq; ++i) q[i] = i % 5 ? input[(i * 7919) % n].first : (next() | (uint64_t(1) << 63));      // 80 % hits
  auto batch = map.find(q.begin(), q.end());
  auto bcount = map.count(q.begin(), q.end());
  const auto t0 = std::chrono::steady_clock::now();
  size_t hits = 0, bi = 0;
  for (size_t i = 0; i < nq; ++i) {
    auto it = map.find(q[i]);
    if (it != map.end()) { CHECK(bi < batch.size() && it->first == batch[bi].first && it->second == batch[bi].second); ++bi; ++hits; }
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CHECK(hits == batch.size() && hits == nq - nq / 5);
  size_t c1 = 0;
  for (size_t i = 0; i < nq; ++i) { const size_t c = map.count(q[i]); CHECK(c == bcount[i]); c1 += c; }
  CHECK(c1 == hits);
  CHECK(sec < 1.0);
  // a mutation drops the host copy: the next single-key calls see it (and go to the GPU again)
  const uint64_t fresh = (uint64_t(1) << 63) | 12345u;
  CHECK(map.find(fresh) == map.end());
  map.insert(fresh, 77u);
  CHECK(map.find(fresh) != map.end() && map.find(fresh)->second == 77u && map.count(fresh) == 1);
  CHECK(map.erase(q[1]) == 1 && map.count(q[1]) == 0 && map.find(q[1]) == map.end());
  for (size_t i = 0; i < 200; ++i) CHECK(map.count(q[1]) == 0 && map.count(fresh) == 1);      // (through a fresh host copy from the 65th call on)
  std::printf("%s: %zu single-key find calls in %.3f s (%.0f ns each), equal to the batch form\n", name, nq, sec, sec / nq * 1e9);
}

int main() {
  differential<fsc::hashmap_robinhood_doubling, uint64_t, fsc::hash::murmur3avx64<uint64_t> >("rh/u64/murmur3avx64", 100000, true);
  differential<fsc::hashmap_robinhood_doubling, uint64_t, std::hash<uint64_t> >("rh/u64/std::hash", 20000, true);
  differential<fsc::hashmap_robinhood_doubling, Kmer31, fsc::hash::farm<Kmer31> >("rh/Kmer31/farm", 30000, true);
  differential<fsc::hashmap_robinhood_doubling, Kmer31, fsc::hash::murmur<Kmer31> >("rh/Kmer31/murmur", 30000, true);
  differential<fsc::hashmap_linearprobe_doubling, uint64_t, fsc::hash::murmur3avx64<uint64_t> >("lp/u64/murmur3avx64", 100000, false);
  differential<fsc::hashmap_linearprobe_doubling, Kmer31, fsc::hash::identity<Kmer31> >("lp/Kmer31/identity", 5000, false);
  // the host operator() of the functors agrees with the device batch hash (used by callers that pre-hash)
  {
    std::vector<uint64_t> k(1000), out(1000);
    for (size_t i = 0; i < k.size(); ++i) k[i] = i * 0x9E3779B97F4A7C15ull + 1;
    fsc::hash::murmur3avx64<uint64_t> h(43);
    CHECK(kh_hash_batch(KH_HASH_MURMUR3_X86_128_LO64, 43, k.data(), k.size(), KH_MEM_HOST, out.data(), 0, nullptr) == KH_OK);
    for (size_t i = 0; i < k.size(); ++i) CHECK(out[i] == h(k[i]));
    CHECK(h(uint64_t(1)) == 0xdbcde6617f85bf2aull);
    CHECK(fsc::hash::murmur<uint64_t>(43)(uint64_t(1)) == 0x252c590efc7e7503ull);
  }
  bimolecule<fsc::hashmap_robinhood_doubling, 31, fsc::hash::farm>("rh/dna_kmer<31>/TransformedHash<farm, lex_less>", 60000);
  bimolecule<fsc::hashmap_robinhood_doubling, 21, fsc::hash::murmur3avx64>("rh/dna_kmer<21>/TransformedHash<murmur3avx64, lex_less>", 20000);
  bimolecule<fsc::hashmap_linearprobe_doubling, 31, fsc::hash::murmur>("lp/dna_kmer<31>/TransformedHash<murmur, lex_less>", 20000);
  // k = 1 and k = 2: most bit patterns are a strand of the same k-mer as their neighbour (the constructor's Equal probe computes what it expects)
  bimolecule<fsc::hashmap_robinhood_doubling, 1, fsc::hash::murmur3avx64>("rh/dna_kmer<1>/TransformedHash<murmur3avx64, lex_less>", 600);
  bimolecule<fsc::hashmap_linearprobe_doubling, 2, fsc::hash::farm>("lp/dna_kmer<2>/TransformedHash<farm, lex_less>", 600);
  single_key_loop<fsc::hashmap_robinhood_doubling>("rh single-key loop");
  single_key_loop<fsc::hashmap_linearprobe_doubling>("lp single-key loop");
  {  // TransformedHash with the identity pre-transform is the plain functor; std::equal_to does not fit a lex_less hash
    using TH = fsc::hash::TransformedHash<uint64_t, fsc::hash::farm>;
    fsc::hashmap_robinhood_doubling<uint64_t, uint32_t, TH> m;
    m.insert(uint64_t(5), 7u);
    CHECK(m.find(uint64_t(5))->second == 7u && TH()(uint64_t(5)) == fsc::hash::farm<uint64_t>()(uint64_t(5)));
    bool threw = false;
    using K31 = kmerhash_amd::dna_kmer<31>;
    try { fsc::hashmap_robinhood_doubling<K31, uint32_t, fsc::TransformedHash<K31, fsc::hash::farm, bliss::kmer::transform::lex_less>, std::equal_to<K31> > bad; }
    catch (std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // an Equal functor that is not bitwise equality is refused at construction
    bool threw = false;
    try { fsc::hashmap_robinhood_doubling<uint64_t, uint32_t, fsc::hash::murmur<uint64_t>, never_equal> bad; }
    catch (std::invalid_argument&) { threw = true; }
    CHECK(threw);
    fsc::hashmap_linearprobe_doubling<uint64_t, uint32_t, fsc::hash::farm<uint64_t> > ok;   // defaults: std::equal_to, 0.2/0.6
    ok.insert(uint64_t(5), 7u);
    CHECK(ok.find(uint64_t(5))->second == 7u);
  }
  std::printf("all shim tests passed\n");
  return 0;
}
