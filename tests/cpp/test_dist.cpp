// C++ side of the sharded map (include/kmerhash_amd_dist.h): the reference's distributed contract
// (distributed_batched_robinhood_map.hpp:910-1194,1258,1619,2169) checked against ONE table that receives the same pairs in the
// order the shards receive them (piece, source rank, position) -- first value wins across ranks.
//   part 1: one RCCL rank (communicator bootstrap through khd_unique_id; p = 1 owns every key), then the same rank with
//           KHD_OPT_FORCE_COLLECTIVES: every RCCL call of the library (ncclAllToAll of the counts, grouped ncclAllToAllv per piece,
//           ncclAllReduce votes) runs as a self-exchange and must give the unsharded table's results
//   part 3: a rank that fails locally (khd_debug_fail_next) at each stage of insert / find / erase: no rank hangs, every rank
//           reports, the maps stay usable;  part 4: pipelined queries (KHD_OPT_QUERY_PIECES) equal the one-piece form
//   part 2: p = 4 and p = 3 ranks as threads of this process on one device (khd_create_local): the sharding / exchange /
//           pipelined streamed insert code of the product over the in-process transport (RCCL refuses two ranks on one GPU)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <chrono>
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
      OK(khd_count(m, dq, nq, ok, oc)); OK(khd_synchronize(m));       // (count / find only queue their work)
      ck_out[r] = host(ok, nq); c_out[r] = host(oc, nq);
      OK(khd_find(m, dq, nq, ok, ov, of)); OK(khd_synchronize(m));
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

// ---- part 3: local failures must reach every rank, and nobody may hang -------------------------------------------------------------
static void failure_votes(int P) {
  std::vector<khd_map*> maps(P);
  OK(khd_create_local(maps.data(), P, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  const size_t n = 50000;
  std::vector<std::vector<uint64_t> > keys(P); std::vector<std::vector<uint32_t> > vals(P);
  for (int r = 0; r < P; ++r) for (size_t i = 0; i < n; ++i) { uint64_t d = (uint64_t(r + 1) << 40) + i; keys[r].push_back(splitmix(d)); vals[r].push_back(uint32_t(i)); }
  // every scenario: rank `bad` fails at `stage` of operation `op` (0 insert, 1 find, 2 erase); st[r] = what rank r got back
  auto run = [&](int op, int stage, int bad, std::vector<kh_status>& st, std::vector<kh_status>& late) {
    st.assign(P, KH_OK); late.assign(P, KH_OK);
    std::vector<std::thread> th;
    for (int r = 0; r < P; ++r)
      th.emplace_back([&, r] {
        CHECK(hipSetDevice(0) == hipSuccess);
        khd_map* m = maps[r];
        uint64_t* dk = dev(keys[r]); uint32_t* dv = dev(vals[r]);
        uint64_t* ok = dev(std::vector<uint64_t>(n)); uint32_t* ov = dev(std::vector<uint32_t>(n)); uint8_t* of = dev(std::vector<uint8_t>(n));
        if (r == bad && stage) OK(khd_debug_fail_next(m, stage));
        uint64_t x = 0;
        if (op == 0) st[r] = khd_insert(m, dk, dv, n, 3, 0, &x);
        else if (op == 1) { st[r] = khd_find(m, dk, n, ok, ov, of); late[r] = khd_synchronize(m); }
        else st[r] = khd_erase(m, dk, 1000, &x);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        hipFree(dk); hipFree(dv); hipFree(ok); hipFree(ov); hipFree(of);
      });
    for (auto& t : th) t.join();
  };
  std::vector<kh_status> st, late;
  auto global_size = [&]() {
    std::vector<uint64_t> sz(P); std::vector<std::thread> th;
    for (int r = 0; r < P; ++r) th.emplace_back([&, r] { CHECK(hipSetDevice(0) == hipSuccess); OK(khd_size(maps[r], &sz[r])); });
    for (auto& t : th) t.join();
    for (int r = 1; r < P; ++r) CHECK(sz[r] == sz[0]);
    return sz[0];
  };
  uint64_t expect = 0;
  for (int stage = 1; stage <= 4; ++stage) {          // insert: all four stages; every rank must report, none may hang
    for (int r = 0; r < P; ++r) for (size_t i = 0; i < n; ++i) { uint64_t d = (uint64_t(stage) << 50) + (uint64_t(r + 1) << 40) + i; keys[r][i] = splitmix(d); }      // fresh keys
    uint64_t own_before = 0; OK(kh_size(khd_local(maps[1]), &own_before));
    run(0, stage, 1, st, late);
    for (int r = 0; r < P; ++r) CHECK(st[r] != KH_OK);
    CHECK(st[1] == KH_ERR_NOMEM);
    CHECK(std::strstr(khd_last_error(maps[1]), "injected") && std::strstr(khd_last_error(maps[0]), "peer rank failed"));
    // the maps stay usable (a collective size).  Stages 1 to 3 fail before any rank builds: nothing inserted anywhere; 4: the
    // healthy ranks hold their share, the failing rank's table is unchanged
    const uint64_t gs = global_size();
    uint64_t own_after = 0; OK(kh_size(khd_local(maps[1]), &own_after));
    CHECK(own_after == own_before);
    if (stage <= 3) CHECK(gs == expect); else CHECK(gs > expect && gs < expect + (uint64_t)P * n);
    run(0, 0, -1, st, late);                            // the same pairs again, nobody fails: first value wins, everything is in
    for (int r = 0; r < P; ++r) CHECK(st[r] == KH_OK);
    expect += (uint64_t)P * n;
    CHECK(global_size() == expect);
  }
  for (int stage = 1; stage <= 3; ++stage) {          // find: stages 1 and 2 are voted at once; stage 3 is reported late to the peers
    run(1, stage, 2 % P, st, late);
    const int bad = 2 % P;
    CHECK(st[bad] == KH_ERR_NOMEM);
    for (int r = 0; r < P; ++r) {
      if (stage <= 2) CHECK(st[r] != KH_OK);
      else if (r != bad) { CHECK(st[r] == KH_OK); CHECK(late[r] == KH_ERR_NOMEM); }
    }
    run(1, 0, -1, st, late);
    for (int r = 0; r < P; ++r) CHECK(st[r] == KH_OK && late[r] == KH_OK);
  }
  for (int stage = 1; stage <= 4; ++stage) {          // erase: nothing is erased anywhere unless every rank sent its real keys (stages 1-3)
    const uint64_t before = global_size();
    run(2, stage, 0, st, late);
    for (int r = 0; r < P; ++r) CHECK(st[r] != KH_OK);
    CHECK(st[0] == KH_ERR_NOMEM);
    const uint64_t after = global_size();
    if (stage <= 3) CHECK(after == before); else CHECK(after < before);
  }
  run(2, 0, -1, st, late);
  for (int r = 0; r < P; ++r) CHECK(st[r] == KH_OK);
  for (int r = 0; r < P; ++r) OK(khd_destroy(maps[r]));
  std::printf("failure votes p=%d ok (insert stages 1-4, find 1-3, erase 1-3: every rank reported, none hung, maps reusable)\n", P);
}

// a rank that never arrives: its peers give up after KHD_OPT_TIMEOUT_MS instead of waiting for ever, and their map is finished
static void absent_peer() {
  const int P = 2;
  std::vector<khd_map*> maps(P);
  OK(khd_create_local(maps.data(), P, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  OK(khd_set_option(maps[1], KHD_OPT_TIMEOUT_MS, 300));
  std::vector<uint64_t> k(1000); uint64_t s = 3; for (auto& x : k) x = splitmix(s);
  uint64_t* dk = dev(k); uint64_t x = 0;
  const auto t0 = std::chrono::steady_clock::now();
  CHECK(khd_erase(maps[1], dk, k.size(), &x) == KH_ERR_HIP);          // rank 0 never calls
  CHECK(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 20.0);
  CHECK(std::strstr(khd_last_error(maps[1]), "never arrived"));
  CHECK(khd_size(maps[1], &x) == KH_ERR_HIP);                         // the map is finished
  CHECK(khd_size(maps[0], &x) == KH_ERR_HIP);                         // ... and so is its peer's: the group has been left
  hipFree(dk);
  for (int r = 0; r < P; ++r) OK(khd_destroy(maps[r]));
  std::printf("absent peer: bounded wait ok\n");
}

// ---- part 4: pipelined queries ----------------------------------------------------------------------------------------------------
static void pipelined_queries(int P, size_t nq, bool timing) {
  std::vector<khd_map*> maps(P);
  OK(khd_create_local(maps.data(), P, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  std::vector<std::vector<uint64_t> > keys(P), q(P); std::vector<std::vector<uint32_t> > vals(P);
  for (int r = 0; r < P; ++r)
    for (size_t i = 0; i < nq; ++i) { uint64_t d = (uint64_t(r + 1) << 40) + i; keys[r].push_back(splitmix(d)); vals[r].push_back(uint32_t(r * 100000000u + i)); }
  for (int r = 0; r < P; ++r) {
    uint64_t s = 17 + r;
    for (size_t i = 0; i < nq; ++i) q[r].push_back(i % 4 ? keys[(r + i) % P][splitmix(s) % nq] : (splitmix(s) | 1ull << 63));     // other ranks' keys + misses
  }
  std::vector<std::vector<uint64_t> > ok1(P), okN(P); std::vector<std::vector<uint32_t> > v1(P), vN(P); std::vector<std::vector<uint8_t> > f1(P), fN(P), c1(P), cN(P);
  std::vector<double> ms1(P), msN(P);
  std::vector<std::thread> th;
  for (int r = 0; r < P; ++r)
    th.emplace_back([&, r] {
      CHECK(hipSetDevice(0) == hipSuccess);
      hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
      khd_map* m = maps[r];
      OK(khd_set_stream(m, st));
      uint64_t* dk = dev(keys[r]); uint32_t* dv = dev(vals[r]); uint64_t* dq = dev(q[r]); uint64_t x = 0;
      OK(khd_insert(m, dk, dv, nq, 2, 0, &x));
      uint64_t* ok = dev(std::vector<uint64_t>(nq)); uint8_t* oc = dev(std::vector<uint8_t>(nq)); uint32_t* ov = dev(std::vector<uint32_t>(nq)); uint8_t* of = dev(std::vector<uint8_t>(nq));
      for (int pieces : {1, 4}) {
        OK(khd_set_option(m, KHD_OPT_QUERY_PIECES, r == 1 && pieces == 4 ? 3 : pieces));      // (ranks may choose differently)
        double best = 1e30;
        for (int rep = 0; rep < (timing ? 4 : 1); ++rep) {
          CHECK(hipMemset(ov, 0xEE, nq * 4) == hipSuccess);
          uint64_t sz = 0; OK(khd_size(m, &sz));              // (collective: the ranks start together)
          auto t0 = std::chrono::steady_clock::now();
          OK(khd_find(m, dq, nq, ok, ov, of)); OK(khd_synchronize(m));
          OK(khd_size(m, &sz));
          best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        OK(khd_count(m, dq, nq, ok, oc)); OK(khd_synchronize(m));
        if (pieces == 1) { ok1[r] = host(ok, nq); v1[r] = host(ov, nq); f1[r] = host(of, nq); c1[r] = host(oc, nq); ms1[r] = best; }
        else { okN[r] = host(ok, nq); vN[r] = host(ov, nq); fN[r] = host(of, nq); cN[r] = host(oc, nq); msN[r] = best; }
      }
      hipFree(dk); hipFree(dv); hipFree(dq); hipFree(ok); hipFree(oc); hipFree(ov); hipFree(of);
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < P; ++r) {
    CHECK(ok1[r] == okN[r] && f1[r] == fN[r] && c1[r] == cN[r] && c1[r] == f1[r]);        // same permuted order, same flags
    size_t hits = 0;
    for (size_t i = 0; i < nq; ++i) if (f1[r][i]) { CHECK(v1[r][i] == vN[r][i]); ++hits; } else CHECK(vN[r][i] == 0u && v1[r][i] == 0u);      // the value of a miss is 0
    CHECK(hits > nq / 2 && hits < nq);
  }
  double a = 0, b = 0; for (int r = 0; r < P; ++r) { a = std::max(a, ms1[r]); b = std::max(b, msN[r]); }
  std::printf("pipelined queries p=%d, %zu finds per rank: 1 piece %.3f ms, 4 pieces %.3f ms (x%.2f)\n", P, nq, a, b, a / b);
  for (int r = 0; r < P; ++r) OK(khd_destroy(maps[r]));
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);      // (progress survives a crash when stdout is a pipe)
  if (argc > 1 && !std::strcmp(argv[1], "--query-timing")) {
    pipelined_queries(4, 10000000, true);
    // the same question over RCCL (one rank, forced self-exchange): what a piece costs in launches when nothing can overlap
    char id[KHD_UNIQUE_ID_BYTES];
    OK(khd_unique_id(id));
    khd_map* m = nullptr;
    OK(khd_create(&m, id, 1, 0, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
    OK(khd_set_option(m, KHD_OPT_FORCE_COLLECTIVES, 1));
    const size_t n = 100000000, nq = 10000000;
    std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; ++i) { uint64_t d = (uint64_t(1) << 40) + i; k[i] = splitmix(d); v[i] = uint32_t(i); }
    uint64_t* dk = dev(k); uint32_t* dv = dev(v); uint64_t x = 0;
    OK(khd_insert(m, dk, dv, n, 4, 0, &x)); CHECK(x == n);
    uint64_t* ok = dev(std::vector<uint64_t>(nq)); uint32_t* ov = dev(std::vector<uint32_t>(nq)); uint8_t* of = dev(std::vector<uint8_t>(nq));
    for (int pieces : {1, 2, 4, 8}) {
      OK(khd_set_option(m, KHD_OPT_QUERY_PIECES, pieces));
      double best = 1e30;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipDeviceSynchronize() == hipSuccess);
        auto t0 = std::chrono::steady_clock::now();
        OK(khd_find(m, dk, nq, ok, ov, of)); OK(khd_synchronize(m));
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      char buf[512]; OK(khd_phase_ms(m, buf, sizeof(buf)));
      std::string ph(buf); for (auto& c : ph) if (c == '\n') c = ' ';
      std::printf("rccl one rank forced, %zu finds in a 1e8-key table, %d piece(s): %.3f ms   [device ms over 6 reps: %s]\n", nq, pieces, best, ph.c_str());
    }
    OK(khd_destroy(m));
    return 0;
  }
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
  {  // one RCCL rank, every collective executed (self-exchange): must equal the unsharded table fed the same pairs
    char id[KHD_UNIQUE_ID_BYTES];
    OK(khd_unique_id(id));
    khd_map* m = nullptr;
    OK(khd_create(&m, id, 1, 0, 0, KH_KIND_ROBINHOOD, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
    OK(khd_set_option(m, KHD_OPT_FORCE_COLLECTIVES, 1));
    OK(khd_set_option(m, KHD_OPT_QUERY_PIECES, 3));
    kh_table* plain = nullptr;
    OK(kh_create(&plain, KH_KIND_ROBINHOOD, 8, 4, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, 0));
    const size_t n = 4500000;       // (capacity 2^23 = 2^12 partitions of ~1100 pairs: the repeatable streamed insert takes its histogram-free layout)
    std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
    uint64_t s = 11; for (size_t i = 0; i < n; ++i) { k[i] = splitmix(s); v[i] = uint32_t(i); }
    for (size_t i = 0; i < 5000; ++i) k[n - 1 - i] = k[i * 7];          // duplicates the first piece's sample cannot see: the retry path over RCCL
    uint64_t* dk = dev(k); uint32_t* dv = dev(v); uint64_t ni = 0, np_ = 0, gs = 0, ne = 0, nep = 0;
    OK(khd_insert(m, dk, dv, n, 4, 0, &ni));
    OK(kh_insert(plain, dk, dv, n, KH_MEM_DEVICE, &np_));
    CHECK(ni == np_ && ni == n - 5000);
    OK(khd_size(m, &gs)); CHECK(gs == ni);                                 // ncclAllReduce(sum)
    CHECK(contents(khd_local(m)) == contents(plain));
    { uint64_t c1 = 0, c2 = 0; OK(kh_capacity(khd_local(m), &c1)); OK(kh_capacity(plain, &c2)); CHECK(c1 == c2);
      std::vector<uint8_t> i1(c1), i2(c2); OK(kh_export_info(khd_local(m), i1.data())); OK(kh_export_info(plain, i2.data())); CHECK(i1 == i2); }
    const size_t nq = 500000;
    std::vector<uint64_t> q(nq); for (size_t i = 0; i < nq; ++i) q[i] = i % 3 ? k[splitmix(s) % n] : splitmix(s);
    uint64_t* dq = dev(q);
    uint64_t* ok = dev(std::vector<uint64_t>(nq)); uint32_t* ov = dev(std::vector<uint32_t>(nq, 7u)); uint8_t* of = dev(std::vector<uint8_t>(nq)); uint8_t* oc = dev(std::vector<uint8_t>(nq));
    uint32_t* pv = dev(std::vector<uint32_t>(nq, 7u)); uint8_t* pf = dev(std::vector<uint8_t>(nq));
    OK(khd_find(m, dq, nq, ok, ov, of)); OK(khd_count(m, dq, nq, nullptr, oc)); OK(khd_synchronize(m));
    OK(kh_find(plain, dq, nq, KH_MEM_DEVICE, pv, pf, nullptr));
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(host(ok, nq) == q && host(of, nq) == host(pf, nq) && host(oc, nq) == host(pf, nq));      // one rank: permuted order == input order
    { auto a = host(ov, nq), b = host(pv, nq); auto f = host(pf, nq);
      for (size_t i = 0; i < nq; ++i) CHECK(f[i] ? a[i] == b[i] : a[i] == 0u); }                   // the value of a miss is 0 in the sharded form
    OK(khd_erase(m, dq, nq, &ne)); OK(kh_erase(plain, dq, nq, KH_MEM_DEVICE, &nep));
    CHECK(ne == nep && ne > 0);
    CHECK(contents(khd_local(m)) == contents(plain));
    // counting insert (std::plus, no values) through the same path
    khd_map* cm = nullptr; kh_table* cplain = nullptr;
    OK(khd_unique_id(id));
    OK(khd_create(&cm, id, 1, 0, 0, KH_KIND_ROBINHOOD, KH_HASH_FARM64, 43, 128, 0.35f, 0.8f, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
    OK(khd_set_option(cm, KHD_OPT_FORCE_COLLECTIVES, 1));
    OK(kh_create(&cplain, KH_KIND_ROBINHOOD, 8, 4, KH_HASH_FARM64, 43, 128, 0.35f, 0.8f, 0));
    OK(khd_insert(cm, dk, nullptr, n, 3, 1, &ni)); OK(kh_insert_reduce_plus(cplain, dk, nullptr, n, KH_MEM_DEVICE, &np_));
    CHECK(ni == np_ && contents(khd_local(cm)) == contents(cplain));
    char buf[512]; OK(khd_phase_ms(m, buf, sizeof(buf)));
    CHECK(std::strstr(buf, "exchange") && std::strstr(buf, "permute") && std::strstr(buf, "refeed"));
    OK(khd_destroy(m)); OK(khd_destroy(cm)); OK(kh_destroy(plain)); OK(kh_destroy(cplain));
    hipFree(dk); hipFree(dv); hipFree(dq); hipFree(ok); hipFree(ov); hipFree(of); hipFree(oc); hipFree(pv); hipFree(pf);
    std::printf("rccl single rank, forced collectives ok (ncclAllToAll counts, grouped ncclAllToAllv x4 pieces + retry, ncclAllReduce votes, pipelined find/count, erase)\n");
  }
  failure_votes(3);
  absent_peer();
  pipelined_queries(3, 300000, false);
  local_group(4, 3, 60000);
  local_group(3, 1, 20000);      // rank = hash % p (not a power of two), one exchange then one bulk insert
  local_group(2, 5, 300000);
  local_group(2, 4, 4000000, 1);   // large enough for the histogram-free layout of the pieces (>= 2^20 pairs per rank at 2^12 partitions)
  local_group(2, 4, 4000000, 2);   // ... and its retry
  std::printf("all dist tests passed\n");
  return 0;
}
