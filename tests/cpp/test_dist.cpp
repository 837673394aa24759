// C++ side of the sharded map (include/kmerhash_amd_dist.h): the reference's distributed contract
// (distributed_batched_robinhood_map.hpp:910-1194,1258,1619,2169) checked against ONE table that receives the same pairs in the
// order the shards receive them (piece, source rank, position) -- first value wins across ranks.
//   part 1: one RCCL rank (communicator bootstrap through khd_unique_id; p = 1 owns every key)
//   part 2: p = 4 and p = 3 ranks as threads of this process on one device (khd_create_local): the sharding / exchange /
//           pipelined streamed insert code of the product over the in-process transport (RCCL refuses two ranks on one GPU)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "kmerhash_amd_dist.h"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
#define OK(c) do { kh_status s__ = (c); if (s__ != KH_OK) { std::printf("status %d at %s:%d: %s\n", (int)s__, __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <typename T> static T* dev(const std::vector<T>& h) {
  T* d = nullptr; CHECK(hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(h.size(), 1) * sizeof(T)) == hipSuccess);
  if (!h.empty()) CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess);
  return d;
}
template <typename T> static std::vector<T> host(const T* d, size_t n) {
  std::vector<T> h(n); if (n) CHECK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess); return h;
}
static std::vector<std::pair<uint64_t, uint32_t> > contents(kh_table* t) {
  uint64_t n = 0; OK(kh_size(t, &n));
  std::vector<uint64_t> k(n + 1); std::vector<uint32_t> v(n + 1); uint64_t m = 0;
  OK(kh_to_vector(t, k.data(), v.data(), &m)); CHECK(m == n);
  std::vector<std::pair<uint64_t, uint32_t> > out(n);
  for (uint64_t i = 0; i < n; ++i) out[i] = std::make_pair(k[i], v[i]);
  std::sort(out.begin(), out.end());
  return out;
}

// keymode 0: every rank draws with replacement from one universe (duplicates everywhere: exact layout of the streamed insert);
// 1: globally distinct keys (the repeatable streamed insert's histogram-free layout holds); 2: distinct, but 300 keys of rank 0
// come again late in rank 1's input -- the first piece's sample cannot see them: KH_ERR_RETRY inside khd_insert, pieces fed again
static void local_group(int P, int pieces, size_t n_per_rank, int keymode = 0) {
  std::vector<khd_map*> maps(P);
  OK(khd_create_local(maps.data(), P, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  // every rank draws from one universe (cross-rank duplicates), own order, own values
  std::vector<std::vector<uint64_t> > keys(P); std::vector<std::vector<uint32_t> > vals(P);
  uint64_t us = 1234;
  std::vector<uint64_t> universe(n_per_rank);
  for (auto& u : universe) u = splitmix(us);
  for (int r = 0; r < P; ++r) {
    uint64_t s = 99 + r;
    for (size_t i = 0; i < n_per_rank; ++i) {
      uint64_t d = (uint64_t(r + 1) << 40) + i;
      keys[r].push_back(keymode == 0 ? universe[splitmix(s) % universe.size()] : splitmix(d));
      vals[r].push_back(uint32_t(r * 10000000u + i));
    }
  }
  if (keymode == 2) for (size_t i = 0; i < 300; ++i) keys[1][n_per_rank - 1000 + i] = keys[0][10 + i];
  // the single table that sees the pairs in receive order: piece-major, then source rank, then position
  kh_table* model = nullptr;
  OK(kh_create(&model, KH_KIND_ROBINHOOD, 8, 4, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, 0));
  {
    std::vector<uint64_t> ck; std::vector<uint32_t> cv;
    for (int pc = 0; pc < pieces; ++pc)
      for (int r = 0; r < P; ++r) {
        // (piece boundaries of khd_insert for up to 8 ranks: kh_shard_plan's, multiples of 4096 pairs)
        const size_t nt = (n_per_rank + 4095) / 4096;
        const size_t a = std::min(n_per_rank, (nt * pc / pieces) * 4096), b = std::min(n_per_rank, (nt * (pc + 1) / pieces) * 4096);
        ck.insert(ck.end(), keys[r].begin() + a, keys[r].begin() + b); cv.insert(cv.end(), vals[r].begin() + a, vals[r].begin() + b);
      }
    uint64_t ni = 0; OK(kh_insert(model, ck.data(), cv.data(), ck.size(), KH_MEM_HOST, &ni));
  }
  const auto gold = contents(model);
  std::map<uint64_t, uint32_t> gmap(gold.begin(), gold.end());
  // queries: own keys + misses
  std::vector<std::vector<uint64_t> > q(P);
  for (int r = 0; r < P; ++r) { uint64_t s = 7 + r; for (size_t i = 0; i < 3000; ++i) q[r].push_back(i % 3 ? keys[r][splitmix(s) % n_per_rank] : (splitmix(s) | 1ull << 63)); }
  std::vector<uint64_t> inserted(P), erased(P), gsize(P), gsize2(P);
  std::vector<int> refed(P, 0);
  std::vector<std::vector<uint64_t> > ck_out(P), fk_out(P); std::vector<std::vector<uint8_t> > c_out(P), f_out(P); std::vector<std::vector<uint32_t> > v_out(P);
  std::vector<std::thread> th;
  for (int r = 0; r < P; ++r)
    th.emplace_back([&, r] {
      CHECK(hipSetDevice(0) == hipSuccess);
      hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
      khd_map* m = maps[r];
      OK(khd_set_stream(m, st));
      uint64_t* dk = dev(keys[r]); uint32_t* dv = dev(vals[r]); uint64_t* dq = dev(q[r]);
      OK(khd_insert(m, dk, dv, keys[r].size(), pieces, 0, &inserted[r]));
      OK(khd_size(m, &gsize[r]));
      const size_t nq = q[r].size();
      uint64_t* ok = dev(std::vector<uint64_t>(nq)); uint8_t* oc = dev(std::vector<uint8_t>(nq)); uint32_t* ov = dev(std::vector<uint32_t>(nq)); uint8_t* of = dev(std::vector<uint8_t>(nq));
      OK(khd_count(m, dq, nq, ok, oc));
      ck_out[r] = host(ok, nq); c_out[r] = host(oc, nq);
      OK(khd_find(m, dq, nq, ok, ov, of));
      fk_out[r] = host(ok, nq); v_out[r] = host(ov, nq); f_out[r] = host(of, nq);
      OK(khd_erase(m, dq, nq, &erased[r]));
      OK(khd_size(m, &gsize2[r]));
      char buf[512]; OK(khd_phase_ms(m, buf, sizeof(buf)));
      if (r == 0) CHECK(std::strstr(buf, "exchange") && std::strstr(buf, "permute"));
      if (r == 0 && keymode == 1 && pieces > 1) CHECK(!std::strstr(buf, "refeed"));
      if (keymode == 2 && pieces > 1) refed[r] = std::strstr(buf, "refeed") != nullptr;
      CHECK(hipStreamSynchronize(st) == hipSuccess);
      hipFree(dk); hipFree(dv); hipFree(dq); hipFree(ok); hipFree(oc); hipFree(ov); hipFree(of);
    });
  for (auto& t : th) t.join();
  if (keymode == 2 && pieces > 1) { int any = 0; for (int r = 0; r < P; ++r) any |= refed[r]; CHECK(any); }      // some rank had to feed its pieces again
  // (a) erase: every queried key that existed is gone, once
  std::vector<uint64_t> allq;
  for (int r = 0; r < P; ++r) allq.insert(allq.end(), q[r].begin(), q[r].end());
  std::sort(allq.begin(), allq.end()); allq.erase(std::unique(allq.begin(), allq.end()), allq.end());
  uint64_t exp_erased = 0; for (auto k : allq) exp_erased += gmap.count(k);
  uint64_t tot_ins = 0, tot_er = 0;
  for (int r = 0; r < P; ++r) { tot_ins += inserted[r]; tot_er += erased[r]; CHECK(gsize[r] == gold.size() && gsize2[r] == gold.size() - exp_erased); }
  CHECK(tot_ins == gold.size() && tot_er == exp_erased);
  // (b) queries: permuted keys are the queries grouped by owner; flags / values are the model's
  for (int r = 0; r < P; ++r) {
    std::vector<uint64_t> a = ck_out[r], b = q[r]; std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); CHECK(a == b && ck_out[r] == fk_out[r]);
    std::vector<uint64_t> hv(q[r].size());
    OK(kh_hash_batch(KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED, ck_out[r].data(), hv.size(), KH_MEM_HOST, hv.data(), 0, nullptr));
    for (size_t i = 0; i < hv.size(); ++i) {
      if (i) CHECK(hv[i] % P >= hv[i - 1] % P);                       // grouped by owner rank, rank 0 first
      auto it = gmap.find(ck_out[r][i]);
      CHECK(c_out[r][i] == (it != gmap.end()) && f_out[r][i] == c_out[r][i]);
      if (it != gmap.end()) CHECK(v_out[r][i] == it->second);          // first value wins across ranks
    }
  }
  // (c) what is left: union of the shards == model minus the erased keys, every key on its owner
  std::vector<std::pair<uint64_t, uint32_t> > uni;
  for (int r = 0; r < P; ++r) {
    auto c = contents(khd_local(maps[r]));
    std::vector<uint64_t> kk, hv(c.size());
    for (auto& e : c) kk.push_back(e.first);
    if (!kk.empty()) OK(kh_hash_batch(KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED, kk.data(), kk.size(), KH_MEM_HOST, hv.data(), 0, nullptr));
    for (auto h : hv) CHECK((int)(h % P) == r);
    uni.insert(uni.end(), c.begin(), c.end());
  }
  std::sort(uni.begin(), uni.end());
  std::vector<std::pair<uint64_t, uint32_t> > exp;
  for (auto& e : gold) if (!std::binary_search(allq.begin(), allq.end(), e.first)) exp.push_back(e);
  CHECK(uni == exp);
  // (d) counting insert (std::plus) through the same pipelined exchange
  std::vector<khd_map*> cm(P);
  OK(khd_create_local(cm.data(), P, 0, KH_KIND_ROBINHOOD, KH_HASH_FARM64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  th.clear();
  for (int r = 0; r < P; ++r)
    th.emplace_back([&, r] {
      CHECK(hipSetDevice(0) == hipSuccess);
      uint64_t* dk = dev(keys[r]); uint64_t ni = 0;
      OK(khd_insert(cm[r], dk, nullptr, keys[r].size(), pieces, 1, &ni));
      hipFree(dk);
    });
  for (auto& t : th) t.join();
  std::map<uint64_t, uint32_t> mult;
  for (int r = 0; r < P; ++r) for (auto k : keys[r]) ++mult[k];
  size_t seen = 0;
  for (int r = 0; r < P; ++r) for (auto& e : contents(khd_local(cm[r]))) { CHECK(mult[e.first] == e.second); ++seen; }
  CHECK(seen == mult.size());
  for (int r = 0; r < P; ++r) { OK(khd_destroy(maps[r])); OK(khd_destroy(cm[r])); }
  OK(kh_destroy(model));
  std::printf("local group p=%d pieces=%d ok (%zu pairs per rank, %zu distinct)\n", P, pieces, n_per_rank, gold.size());
}

int main() {
  {  // one RCCL rank
    char id[KHD_UNIQUE_ID_BYTES];
    OK(khd_unique_id(id));
    khd_map* m = nullptr;
    OK(khd_create(&m, id, 1, 0, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
    CHECK(khd_nranks(m) == 1 && khd_rank(m) == 0);
    std::vector<uint64_t> k(50000); std::vector<uint32_t> v(50000);
    uint64_t s = 5; for (size_t i = 0; i < k.size(); ++i) { k[i] = splitmix(s); v[i] = uint32_t(i); }
    uint64_t* dk = dev(k); uint32_t* dv = dev(v); uint64_t ni = 0, gs = 0, ne = 0;
    OK(khd_insert(m, dk, dv, k.size(), 4, 0, &ni)); CHECK(ni == k.size());
    OK(khd_size(m, &gs)); CHECK(gs == k.size());
    uint64_t* ok = dev(std::vector<uint64_t>(1000)); uint32_t* ov = dev(std::vector<uint32_t>(1000)); uint8_t* of = dev(std::vector<uint8_t>(1000));
    OK(khd_find(m, dk, 1000, ok, ov, of));
    auto hv = host(ov, 1000); auto hf = host(of, 1000); auto hk = host(ok, 1000);
    for (size_t i = 0; i < 1000; ++i) CHECK(hf[i] == 1 && hv[i] == v[i] && hk[i] == k[i]);
    OK(khd_erase(m, dk, 1000, &ne)); CHECK(ne == 1000);
    OK(khd_destroy(m));
    std::printf("rccl single rank ok\n");
  }
  local_group(4, 3, 60000);
  local_group(3, 1, 20000);      // rank = hash % p (not a power of two), one exchange then one bulk insert
  local_group(2, 5, 300000);
  local_group(2, 4, 4000000, 1);   // large enough for the histogram-free layout of the pieces (>= 2^20 pairs per rank at 2^12 partitions)
  local_group(2, 4, 4000000, 2);   // ... and its retry
  std::printf("all dist tests passed\n");
  return 0;
}
