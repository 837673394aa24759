// where does the time of a host-vector find() go?  (PCIe-inclusive path of the C++ shim)
// g++ -std=c++11 -O2 -Iinclude scripts/host_path_timing.cpp -Lkmerhash_amd -lkmerhash_amd -Wl,-rpath,$PWD/kmerhash_amd -o /tmp/hpt
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "kmerhash_amd.h"
#include <sys/mman.h>
#include <fstream>
#include <iostream>
static void advise(void* p, size_t bytes) { uintptr_t a = ((uintptr_t)p + 4095) & ~uintptr_t(4095); uintptr_t e = ((uintptr_t)p + bytes) & ~uintptr_t(4095); if (e > a) madvise((void*)a, e - a, MADV_HUGEPAGE); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const uint64_t n = 100000000, nq = 10000000;
  std::vector<uint64_t> keys(n); std::vector<uint32_t> vals(n);
  uint64_t s = 1;
  for (uint64_t i = 0; i < n; ++i) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; keys[i] = z ^ (z >> 31); vals[i] = (uint32_t)i; }
  kh_table* t;
  if (kh_create(&t, KH_KIND_ROBINHOOD, 8, 4, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, 0.35f, 0.8f, 0) != KH_OK) return 1;
  uint64_t ni = 0;
  for (int rep = 0; rep < 2; ++rep) {
    kh_clear(t);
    double t0 = now();
    kh_insert(t, keys.data(), vals.data(), n, KH_MEM_HOST, &ni);
    std::printf("insert host SoA        %.2f ms (%llu new)\n", (now() - t0) * 1e3, (unsigned long long)ni);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    std::vector<uint64_t> k(keys.begin(), keys.begin() + nq);
    double t1 = now();
    std::vector<std::pair<uint64_t, uint32_t> > out(nq);
    double t2 = now();
    uint64_t nf = 0;
    kh_find_compact_pairs(t, k.data(), nq, KH_MEM_HOST, out.data(), &nf);
    double t3 = now();
    std::vector<uint8_t> c(nq);
    kh_count(t, k.data(), nq, KH_MEM_HOST, c.data());
    double t4 = now();
    std::vector<uint64_t> c8(c.begin(), c.end());
    double t5 = now();
    std::printf("gather %.2f  alloc-out %.2f  find_compact_pairs %.2f (%llu hits)  count %.2f  widen-to-size_t %.2f ms\n",
                (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (unsigned long long)nf, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
  }
  { std::ifstream f("/sys/kernel/mm/transparent_hugepage/enabled"); std::string l; std::getline(f, l); std::cout << "THP: " << l << std::endl; }
  for (int rep = 0; rep < 3; ++rep) {
    double t1 = now();
    std::vector<std::pair<uint64_t, uint32_t> > out;
    out.reserve(nq); advise(out.data(), nq * 16); out.resize(nq);
    double t2 = now();
    uint64_t nf = 0;
    kh_find_compact_pairs(t, keys.data(), nq, KH_MEM_HOST, out.data(), &nf);
    double t3 = now();
    std::vector<uint8_t> c(nq);
    kh_count(t, keys.data(), nq, KH_MEM_HOST, c.data());
    double t4 = now();
    std::vector<uint64_t> c8; c8.reserve(nq); advise(c8.data(), nq * 8); c8.assign(c.begin(), c.end());
    double t5 = now();
    std::printf("[madvise] alloc-out %.2f  find_compact_pairs %.2f (%llu hits)  count %.2f  widen-to-size_t %.2f ms\n",
                (t2 - t1) * 1e3, (t3 - t2) * 1e3, (unsigned long long)nf, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
  }
  kh_destroy(t);
  return 0;
}
