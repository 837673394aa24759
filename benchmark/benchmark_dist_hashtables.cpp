// benchmark_dist_hashtables -- the reference's distributed benchmark driver (benchmark/BenchmarkDistHashTables.cpp:787-1100,
// "benchmarkHT") on the sharded GPU table: same flag set, same phases (insert, count, find, erase over per-rank generated
// pairs, :908-1100), the MPI exchange replaced by RCCL (libkmerhash_amd_dist.so).  Own driver, not derived from the reference file.
//
//   flags of the reference:  -F/--file <key-val binary file>  -C/--count <total pairs>  -R/--repeat-rate <mean multiplicity>
//                            --missing-frac <f>  --max_load <f>  --min_load <f>  --insert_prefetch <n>  --query_prefetch <n>  --hybrid
//   launch (the reference is started by mpirun; there is no MPI here):
//       one process per GPU:  --nranks N --rank r --id-file <path>   (rank 0 writes the RCCL id there, the others wait for it)
//       one process, N ranks as threads on GPU 0 (in-process transport, for one-GPU boxes):  --local-ranks N
//   plus  --pieces k  (pipelined exchange, default 4 when N > 1)  and  -m robinhood|linearprobe
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "kmerhash_amd_dist.h"

#define DIE(...) do { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::exit(1); } while (0)
#define OK(c) do { kh_status s__ = (c); if (s__ != KH_OK) DIE("status %d at %s:%d: %s", (int)s__, __FILE__, __LINE__, #c); } while (0)
#define HIP(c) do { if ((c) != hipSuccess) DIE("HIP error at %s:%d: %s", __FILE__, __LINE__, #c); } while (0)

static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

struct Opt {
  std::string file, id_file, map = "robinhood";
  size_t count = 100000000; float repeats = 8.0f, missing = 0.0f; double max_load = 0.8, min_load = 0.35;
  int nranks = 1, rank = 0, local_ranks = 0, pieces = 0;
};

static void usage() {
  std::fprintf(stderr, "usage: benchmark_dist_hashtables [-F file] [-C count] [-R repeat-rate] [--missing-frac f] [--max_load f] [--min_load f]\n"
                       "       [--insert_prefetch n] [--query_prefetch n] [-m robinhood|linearprobe] [--pieces k]\n"
                       "       (--nranks N --rank r --id-file path | --local-ranks N)\n");
  std::exit(1);
}

// keys of a dumped input (reference io_utils.hpp:57-103: size_t element size, size_t count, raw elements; 8-byte keys or 16-byte pairs)
static std::vector<uint64_t> read_keys(const std::string& fn) {
  std::ifstream f(fn, std::ios::binary);
  if (!f) DIE("cannot open %s", fn.c_str());
  uint64_t es = 0, n = 0;
  f.read(reinterpret_cast<char*>(&es), 8); f.read(reinterpret_cast<char*>(&n), 8);
  if (es != 8 && es != 16) DIE("ERROR: element size mismatch in %s (8 or 16 expected, %llu found)", fn.c_str(), (unsigned long long)es);
  std::vector<uint64_t> raw(n * es / 8);
  f.read(reinterpret_cast<char*>(raw.data()), n * es);
  if (es == 8) return raw;
  std::vector<uint64_t> k(n);
  for (uint64_t i = 0; i < n; ++i) k[i] = raw[2 * i];
  return k;
}

struct Times { double insert = 0, count = 0, find = 0, erase = 0; uint64_t inserted = 0, size = 0, hits = 0, erased = 0, nq = 0; };

static void run_rank(khd_map* m, const Opt& o, int rank, int nranks, Times& T) {
  HIP(hipSetDevice(0 + (o.local_ranks ? 0 : rank)));
  const size_t count = o.count / nranks;
  // ---- input of this rank (:908-936): unique keys (from the file, or generated), each about `repeats` times, values = position
  std::vector<uint64_t> uniq;
  if (!o.file.empty()) {
    std::vector<uint64_t> all = read_keys(o.file);
    for (size_t i = rank; i < all.size(); i += nranks) uniq.push_back(all[i]);
  }
  uint64_t s = 1000 + rank;
  const size_t want = std::max<size_t>(1, (size_t)((double)count / std::max(1.0f, o.repeats)));
  while (uniq.size() < want) uniq.push_back(splitmix(s) >> 2);
  std::vector<uint64_t> keys(count); std::vector<uint32_t> vals(count);
  for (size_t i = 0; i < count; ++i) { keys[i] = o.repeats <= 1.0f ? uniq[i % uniq.size()] : uniq[splitmix(s) % uniq.size()]; vals[i] = (uint32_t)i; }
  // queries: the input keys, a fraction replaced by keys that are not in the table (:944-960)
  std::vector<uint64_t> q(keys);
  for (size_t i = 0; i < q.size(); ++i) if ((double)(splitmix(s) >> 11) / 9007199254740992.0 < o.missing) q[i] = splitmix(s) | (1ull << 63);
  uint64_t *dk, *dq, *ok; uint32_t *dv, *ov; uint8_t* of;
  HIP(hipMalloc((void**)&dk, count * 8 + 8)); HIP(hipMalloc((void**)&dq, count * 8 + 8)); HIP(hipMalloc((void**)&ok, count * 8 + 8));
  HIP(hipMalloc((void**)&dv, count * 4 + 8)); HIP(hipMalloc((void**)&ov, count * 4 + 8)); HIP(hipMalloc((void**)&of, count + 8));
  HIP(hipMemcpy(dk, keys.data(), count * 8, hipMemcpyHostToDevice)); HIP(hipMemcpy(dv, vals.data(), count * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dq, q.data(), count * 8, hipMemcpyHostToDevice));
  const int pieces = o.pieces > 0 ? o.pieces : (nranks > 1 ? 4 : 1);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  uint64_t sz = 0;
  OK(khd_size(m, &sz));                                         // (collective: every rank starts the phase together)
  auto t0 = now(); OK(khd_insert(m, dk, dv, count, pieces, 0, &T.inserted)); OK(khd_size(m, &T.size)); auto t1 = now();
  OK(khd_count(m, dq, count, ok, of)); OK(khd_synchronize(m)); OK(khd_size(m, &sz)); auto t2 = now();
  OK(khd_find(m, dq, count, ok, ov, of)); OK(khd_synchronize(m)); OK(khd_size(m, &sz)); auto t3 = now();
  std::vector<uint8_t> hf(count);
  HIP(hipMemcpy(hf.data(), of, count, hipMemcpyDeviceToHost));
  for (auto b : hf) T.hits += b;
  auto t4 = now(); OK(khd_erase(m, dq, count, &T.erased)); OK(khd_size(m, &sz)); auto t5 = now();
  T.insert = secs(t0, t1); T.count = secs(t1, t2); T.find = secs(t2, t3); T.erase = secs(t4, t5); T.nq = count;
  // self-check: every query that was an input key is found
  uint64_t expect = 0;
  for (size_t i = 0; i < count; ++i) expect += q[i] == keys[i];
  if (T.hits != expect) DIE("SELF-CHECK FAILED: rank %d found %llu of %llu present query keys", rank, (unsigned long long)T.hits, (unsigned long long)expect);
  if (sz + 0 > T.size) DIE("SELF-CHECK FAILED: size grew during erase");
  hipFree(dk); hipFree(dq); hipFree(ok); hipFree(dv); hipFree(ov); hipFree(of);
}

static void report(const Opt& o, int nranks, const std::vector<Times>& T) {
  double ti = 0, tc = 0, tf = 0, te = 0; uint64_t nq = 0, ins = 0, er = 0;
  for (auto& t : T) { ti = std::max(ti, t.insert); tc = std::max(tc, t.count); tf = std::max(tf, t.find); te = std::max(te, t.erase); nq += t.nq; ins += t.inserted; er += t.erased; }
  std::printf("benchmark_dist_hashtables: map %s  ranks %d  total pairs %llu  repeat-rate %.2f  missing-frac %.2f  max_load %.2f  min_load %.2f\n",
              o.map.c_str(), nranks, (unsigned long long)nq, o.repeats, o.missing, o.max_load, o.min_load);
  std::printf("  global size after insert %llu (inserted %llu), erased %llu\n", (unsigned long long)T[0].size, (unsigned long long)ins, (unsigned long long)er);
  const char* names[4] = {"insert", "count", "find", "erase"}; const double tt[4] = {ti, tc, tf, te};
  for (int i = 0; i < 4; ++i) std::printf("  %-7s %9.3f ms  %9.2f M ops/s (slowest rank)\n", names[i], tt[i] * 1e3, nq / tt[i] / 1e6);
}

int main(int argc, char** argv) {
  Opt o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&]() -> const char* { if (i + 1 >= argc) usage(); return argv[++i]; };
    if (a == "-F" || a == "--file") o.file = val();
    else if (a == "-C" || a == "--count") o.count = std::strtoull(val(), nullptr, 10);
    else if (a == "-R" || a == "--repeat-rate") o.repeats = (float)std::atof(val());
    else if (a == "--missing-frac") o.missing = (float)std::atof(val());
    else if (a == "--max_load") o.max_load = std::atof(val());
    else if (a == "--min_load") o.min_load = std::atof(val());
    else if (a == "--insert_prefetch" || a == "--query_prefetch") (void)val();      // software prefetch distances of the CPU tables: accepted, no meaning here
    else if (a == "--hybrid") DIE("--hybrid (OpenMP + MPI tables) is not part of this path: one process (or thread) per GPU");
    else if (a == "-m") o.map = val();
    else if (a == "--nranks") o.nranks = std::atoi(val());
    else if (a == "--rank") o.rank = std::atoi(val());
    else if (a == "--id-file") o.id_file = val();
    else if (a == "--local-ranks") o.local_ranks = std::atoi(val());
    else if (a == "--pieces") o.pieces = std::atoi(val());
    else usage();
  }
  if (o.map != "robinhood" && o.map != "linearprobe") DIE("unknown map type %s (robinhood, linearprobe)", o.map.c_str());
  if (o.missing < 0.f || o.missing > 1.f || o.max_load <= 0 || o.max_load >= 1 || o.min_load < 0 || o.min_load >= o.max_load) DIE("load factors / missing-frac out of range");
  const kh_kind kind = o.map == "robinhood" ? KH_KIND_ROBINHOOD : KH_KIND_LINEARPROBE;
  if (o.local_ranks > 0) {
    std::vector<khd_map*> maps(o.local_ranks);
    OK(khd_create_local(maps.data(), o.local_ranks, 0, kind, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, (float)o.min_load, (float)o.max_load, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
    std::vector<Times> T(o.local_ranks);
    std::vector<std::thread> th;
    for (int r = 0; r < o.local_ranks; ++r) th.emplace_back([&, r] { run_rank(maps[r], o, r, o.local_ranks, T[r]); });
    for (auto& t : th) t.join();
    report(o, o.local_ranks, T);
    for (auto m : maps) khd_destroy(m);
    return 0;
  }
  if (o.nranks < 1 || o.rank < 0 || o.rank >= o.nranks) usage();
  char id[KHD_UNIQUE_ID_BYTES];
  if (o.nranks > 1 && o.id_file.empty()) DIE("--nranks > 1 needs --id-file (rank 0 writes the RCCL id there)");
  if (o.rank == 0) {
    OK(khd_unique_id(id));
    if (!o.id_file.empty()) { std::ofstream f(o.id_file + ".tmp", std::ios::binary); f.write(id, sizeof(id)); f.close(); std::rename((o.id_file + ".tmp").c_str(), o.id_file.c_str()); }
  } else {
    for (int tries = 0; tries < 6000; ++tries) { std::ifstream f(o.id_file, std::ios::binary); if (f && f.read(id, sizeof(id))) break; usleep(10000); if (tries == 5999) DIE("no id file"); }
  }
  khd_map* m = nullptr;
  OK(khd_create(&m, id, o.nranks, o.rank, o.rank, kind, KH_HASH_MURMUR3_X86_128_LO64, 43, 128, (float)o.min_load, (float)o.max_load, KH_HASH_MURMUR3_X86_128_LO64, KHD_DIST_SEED));
  std::vector<Times> T(1);
  run_rank(m, o, o.rank, o.nranks, T[0]);
  if (o.rank == 0) report(o, o.nranks, T);        // (rank 0's own times; every phase ends with a collective, so they bound the slowest rank)
  khd_destroy(m);
  return 0;
}
