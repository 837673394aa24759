// benchmark_hashtables.cpp -- our own driver for the reference's primary single-process benchmark
// (reference benchmark/BenchmarkHashTables.cpp: main :1591, benchmark_hashmap<MAP> :1037-1186), written against
// the drop-in headers.  It keeps the phase sequence and the run-time flags that matter for this path:
//
//     -m robinhood|linearprobe   map type                 (:1399-1409; the GPU tables of this repo)
//     -N <count>                 input pairs, default 100M (:1441)
//     -Q <query_frac>            queries = N / Q, default 10 (README.md:84 run; the source default is 2)
//     -R <repeat_rate>           multiplicity 1..R per key, default 10 (mean 5.5)   (:192-223)
//     --max_load / --min_load    load factors, default 0.8 / 0.35                    (:1053-1056)
//     -r <repeats>               timed repeats of the whole sequence (fresh map each), default 1
//     -F <file>                  replay a dumped input instead of generating one (:241-248): the reference's
//                                serialize_vector format `size_t elsize(=16), size_t n, pair<uint64,uint32>[n]` (io_utils.hpp:57-103)
//     -I iter|index|integrated|sort|shuffle   which insert overload is called (:1457, :1100-1112, :905-963); default index
//     -A dna|dna5|dna16          alphabet (:1448): only dna (2 bits/base, one 64-bit word) exists on this path, others are refused
//     -f                         k-mer fully occupies the machine word: 64 key bits instead of the 31-mer's 62 (:1463)
//     -c                         canonical k-mers (:1464): the generated key becomes min(kmer, reverse complement)
//     --measured_op <op>         estimate|insert|find|count|erase|count2|disabled (:1486): in the reference it selects the phase
//                                bracketed by VTune/LIKWID markers; here every phase is timed, the named one is marked with '*'
//     --insert_prefetch/--query_prefetch <n>   accepted and ignored (software-prefetch distances of the CPU tables, :1474)
//
// Phases, each timed on the host clock around the batch call (host vectors in, host vectors out -- the
// reference's semantics, so the numbers INCLUDE PCIe transfers): insert, find, count, erase, count2.
// Input generation uses splitmix64 instead of glibc rand()/random_shuffle (portable streams).
// The kmerind pieces of the reference driver (TCLAP, BL_BENCH, mxx, Kmer) are not used.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "kmerhash/hashmap_robinhood.hpp"
#include "kmerhash/hashmap_linearprobe.hpp"

namespace {

struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); } };

typedef std::pair<uint64_t, uint32_t> pair_t;

// generate_input (:182-227): key = 62-bit draw (31-mer sanitize); emit (key,i) then draw%repeats more copies; shuffle
// reverse complement of a k-mer packed 2 bits per base (A=0 C=1 G=2 T=3: complement = 3 - base), k bases in the low 2k bits
uint64_t revcomp(uint64_t x, unsigned k) {
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
}

std::vector<pair_t> generate_input(size_t count, size_t repeats, bool full, bool canonical) {
  std::vector<pair_t> out;
  out.reserve(count);
  SplitMix g{23};
  const unsigned kk = full ? 32 : 31;
  for (size_t i = 0; i < count;) {
    uint64_t k = full ? g.next() : (g.next() & ((uint64_t(1) << 62) - 1));
    if (canonical) { const uint64_t rc = revcomp(k, kk); if (rc < k) k = rc; }
    out.push_back(pair_t(k, uint32_t(i))); ++i;
    size_t freq = g.next() % repeats;
    for (size_t j = 0; j < freq && i < count; ++j, ++i) out.push_back(pair_t(k, uint32_t(i)));
  }
  SplitMix sh{29};
  for (size_t i = out.size(); i > 1; --i) std::swap(out[i - 1], out[sh.next() % i]);   // Fisher-Yates
  return out;
}

// deserialize_vector<std::pair<uint64_t,uint32_t>> (io_utils.hpp:83-103): throws std::logic_error on an element-size mismatch
std::vector<pair_t> load_input(std::string const& path) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  size_t hdr[2] = {0, 0};
  if (std::fread(hdr, sizeof(size_t), 2, f) != 2) { std::fclose(f); throw std::runtime_error("truncated header"); }
  if (hdr[0] != sizeof(pair_t)) { std::fclose(f); throw std::logic_error("input element size not as specified "); }
  std::vector<pair_t> v(hdr[1]);
  const size_t got = std::fread(static_cast<void*>(v.data()), sizeof(pair_t), hdr[1], f);
  std::fclose(f);
  if (got != hdr[1]) throw std::runtime_error("truncated file");
  return v;
}

struct Timer {
  std::chrono::steady_clock::time_point t0;
  void start() { t0 = std::chrono::steady_clock::now(); }
  double stop() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// benchmark_hashmap<MAP> (:1037-1186): the map type arrives as a 5-parameter template-template
template <template <typename, typename, typename, typename, typename> class MAP>
void benchmark_hashmap(std::string const& name, std::vector<pair_t> const& input, size_t query_frac, float max_load, float min_load,
                       std::string const& insert_mode, std::string const& measured) {
  using MAP_TYPE = MAP<uint64_t, uint32_t, ::fsc::hash::murmur3avx64<uint64_t>, ::std::equal_to<uint64_t>, ::std::allocator<pair_t> >;
  Timer tm;
  MAP_TYPE map;
  map.set_max_load_factor(max_load);
  map.set_min_load_factor(min_load);
  std::vector<uint64_t> query;
  query.reserve(input.size() / query_frac);
  for (size_t i = 0; i < input.size() / query_frac; ++i) query.push_back(input[i].first);

  tm.start();
  if (insert_mode == "iter") map.insert(input.begin(), input.end());          // insert(Iter, Iter)            (:1110)
  else map.insert(input);                                                      // insert(vector const&); the RH-only integrated /
                                                                               // sort / shuffle variants are the same algorithm
                                                                               // (hashmap_robinhood.hpp:721-836,843,1002)
  double t_ins = tm.stop();
  size_t sz = map.size();
  // the reference's find phase AS WRITTEN (BenchmarkHashTables.cpp:1134-1138): a loop of single-key map.find(q) calls.  The shim
  // serves the first 64 from the GPU and the rest from a host copy of the slot array (one 16 B x capacity transfer, inside the time);
  // the batch form find(Iter,Iter) -- one GPU launch -- is timed next to it
  tm.start();
  size_t result = 0;
  for (auto q : query) { auto iter = map.find(q); if (iter != map.end()) ++result; }
  double t_find1 = tm.stop();
  tm.start(); auto found = map.find(query.begin(), query.end()); double t_find = tm.stop();
  tm.start(); auto counts = map.count(query.begin(), query.end()); double t_count = tm.stop();
  size_t present = 0; for (auto c : counts) present += c;
  tm.start(); size_t erased = map.erase(query.begin(), query.end()); double t_erase = tm.stop();
  tm.start(); auto counts2 = map.count(query.begin(), query.end()); double t_count2 = tm.stop();
  size_t present2 = 0; for (auto c : counts2) present2 += c;

  std::printf("[%s] N=%zu distinct=%zu capacity=%zu queries=%zu\n", name.c_str(), input.size(), sz, size_t(map.capacity()), query.size());
  auto mark = [&](const char* op) { return measured == op ? '*' : ' '; };
  std::printf(" %cinsert  %9.4f s  %10.3f M/s  (%s)\n", mark("insert"), t_ins, input.size() / t_ins / 1e6, insert_mode == "iter" ? "insert" : "v_insert");
  std::printf(" %cfind    %9.4f s  %10.3f M/s  (found %zu; single-key loop as in the reference)\n", mark("find"), t_find1, query.size() / t_find1 / 1e6, result);
  std::printf("  find_b  %9.4f s  %10.3f M/s  (found %zu; batch form)\n", t_find, query.size() / t_find / 1e6, found.size());
  std::printf(" %ccount   %9.4f s  %10.3f M/s  (present %zu)\n", mark("count"), t_count, query.size() / t_count / 1e6, present);
  std::printf(" %cerase   %9.4f s  %10.3f M/s  (erased %zu)\n", mark("erase"), t_erase, query.size() / t_erase / 1e6, erased);
  std::printf(" %ccount2  %9.4f s  %10.3f M/s  (present %zu)\n", mark("count2"), t_count2, query.size() / t_count2 / 1e6, present2);
  if (found.size() != query.size() || result != query.size() || present != query.size() || present2 != 0 || map.size() != sz - erased) {
    std::printf("  SELF-CHECK FAILED\n");
    std::exit(2);
  }
}

}  // namespace

int main(int argc, char** argv) {
  std::string map = "robinhood", fname, insert_mode = "index", measured = "insert", alphabet = "dna";
  bool full = false, canonical = false;
  size_t N = 100000000, Q = 10, R = 10, reps = 1;
  float max_load = 0.8f, min_load = 0.35f;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto need = [&](const char* f) { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", f); std::exit(1); } return argv[++i]; };
    if (a == "-m") map = need("-m");
    else if (a == "-N") N = std::strtoull(need("-N"), nullptr, 10);
    else if (a == "-Q") Q = std::strtoull(need("-Q"), nullptr, 10);
    else if (a == "-R") R = std::strtoull(need("-R"), nullptr, 10);
    else if (a == "-F") fname = need("-F");
    else if (a == "-r") reps = std::strtoull(need("-r"), nullptr, 10);
    else if (a == "--max_load") max_load = std::strtof(need("--max_load"), nullptr);
    else if (a == "--min_load") min_load = std::strtof(need("--min_load"), nullptr);
    else if (a == "-I" || a == "--insert_mode") insert_mode = need("-I");
    else if (a == "-A" || a == "--alphabet") alphabet = need("-A");
    else if (a == "-f" || a == "--full") full = true;
    else if (a == "-c" || a == "--canonical") canonical = true;
    else if (a == "--measured_op") measured = need("--measured_op");
    else if (a == "--insert_prefetch" || a == "--query_prefetch") need(a.c_str());   // CPU software-prefetch distances: no meaning here
    else { std::fprintf(stderr, "usage: %s [-m robinhood|linearprobe] [-N n] [-Q query_frac] [-R repeat_rate] [-r repeats] [-F file] [--max_load f] [--min_load f]\n"
                                "          [-I iter|index|integrated|sort|shuffle] [-A dna] [-f] [-c] [--measured_op op] [--insert_prefetch n] [--query_prefetch n]\n", argv[0]); return 1; }
  }
  if (Q == 0 || R == 0) return 1;
  if (alphabet != "dna") { std::fprintf(stderr, "alphabet %s: only dna (one 64-bit word per k-mer) is implemented on this path\n", alphabet.c_str()); return 1; }
  if (insert_mode != "iter" && insert_mode != "index" && insert_mode != "integrated" && insert_mode != "sort" && insert_mode != "shuffle") {
    std::fprintf(stderr, "unknown insert mode %s\n", insert_mode.c_str()); return 1; }
  if ((insert_mode == "integrated" || insert_mode == "sort" || insert_mode == "shuffle") && map != "robinhood") {
    std::fprintf(stderr, "insert mode %s exists for the robinhood map only (BenchmarkHashTables.cpp:905-963)\n", insert_mode.c_str()); return 1; }
  std::vector<pair_t> input = fname.empty() ? generate_input(N, R, full, canonical) : load_input(fname);
  for (size_t r = 0; r < reps; ++r) {
    if (map == "robinhood") benchmark_hashmap<::fsc::hashmap_robinhood_doubling>("robinhood", input, Q, max_load, min_load, insert_mode, measured);
    else if (map == "linearprobe") benchmark_hashmap<::fsc::hashmap_linearprobe_doubling>("linearprobe", input, Q, max_load, min_load, insert_mode, measured);
    else { std::fprintf(stderr, "unknown map %s\n", map.c_str()); return 1; }
  }
  return 0;
}
