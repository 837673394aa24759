// Measures what the find path is priced against: the rate at which one MI355X serves RANDOM 64-byte sector reads from a table-sized
// buffer (2 GiB = 2^27 slots of 16 bytes, the configs[1] table), issued the way k_find issues them (four 16-byte loads of one aligned
// sector per query, several queries in flight per lane).  Not part of the library; build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ra scripts/random_access_roofline.hip && /tmp/ra
// Output: one line per variant with queries/s and sector bytes/s.  `profiles/README.md` quotes the numbers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {      // splitmix64 finaliser: cheap, good enough to defeat any locality
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// ITEMS independent sector reads in flight per lane; SECTOR = 16-byte loads per touch (4 = one 64-byte sector, 1 = one slot);
// CHAIN = dependent touches per query (2 mimics a probe that needs a second sector)
template <int ITEMS, int SECTOR, int CHAIN>
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ buf, uint64_t nslots, uint64_t nq, uint32_t* __restrict__ sink) {
  const uint64_t mask = nslots - 1;
  const uint64_t stride = (uint64_t)gridDim.x * 256 * ITEMS;
  uint32_t acc = 0;
  for (uint64_t b = (uint64_t)blockIdx.x * 256 * ITEMS + threadIdx.x; b < nq; b += stride) {
    uint64_t pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) pos[j] = mix(b + (uint64_t)j * 256) & mask & ~(uint64_t)(SECTOR - 1);
#pragma unroll
    for (int c = 0; c < CHAIN; ++c) {
      uint4 w[ITEMS][SECTOR];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j)
#pragma unroll
        for (int s = 0; s < SECTOR; ++s) w[j][s] = buf[pos[j] + s];
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        uint32_t x = 0;
#pragma unroll
        for (int s = 0; s < SECTOR; ++s) x ^= w[j][s].x ^ w[j][s].w;
        acc += x;
        pos[j] = (mix(pos[j] + x + c) & mask) & ~(uint64_t)(SECTOR - 1);      // the next touch depends on what was read
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;      // (keeps the loads alive)
}

template <int ITEMS, int SECTOR, int CHAIN>
static void run(const char* name, const uint4* buf, uint64_t nslots, uint64_t nq, uint32_t* sink, int wgs_per_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * wgs_per_cu;
  std::vector<float> ms;
  for (int r = 0; r < 6; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_touch<ITEMS, SECTOR, CHAIN>), dim3(grid), dim3(256), 0, 0, buf, nslots, nq, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float t; CHECK(hipEventElapsedTime(&t, e0, e1));
    if (r) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double t = ms[ms.size() / 2] * 1e-3;
  const double touches = (double)nq * CHAIN;
  printf("%-34s items %d x sector %3d B x chain %d, %2d WG/CU: %8.3f ms  %.3e queries/s  %.3e touches/s  %7.1f GB/s of touched bytes\n", name, ITEMS,
         SECTOR * 16, CHAIN, wgs_per_cu, t * 1e3, nq / t, touches / t, touches * SECTOR * 16 / t * 1e-9);
  fflush(stdout);
}

int main() {
  const uint64_t nslots = 1ull << 27;      // 2 GiB of 16-byte slots
  uint4* buf; uint32_t* sink;
  CHECK(hipMalloc(&buf, nslots * 16)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 0x5A, nslots * 16)); CHECK(hipMemset(sink, 0, 64));
  CHECK(hipDeviceSynchronize());
  const uint64_t nq = 100000000ull;
  for (int occ : {4, 8}) {
    run<4, 4, 1>("sector, independent", buf, nslots, nq, sink, occ);
    run<8, 4, 1>("sector, independent", buf, nslots, nq, sink, occ);
    run<4, 1, 1>("slot (16 B), independent", buf, nslots, nq, sink, occ);
    run<4, 4, 2>("sector, two dependent touches", buf, nslots, nq, sink, occ);
    run<4, 8, 1>("128-byte line, independent", buf, nslots, nq, sink, occ);
  }
  // the find workload's size: 10^7 queries (launch ramp and tail included)
  run<4, 4, 1>("sector, independent, 1e7 queries", buf, nslots, 10000000ull, sink, 8);
  run<4, 4, 2>("sector, two touches, 1e7 queries", buf, nslots, 10000000ull, sink, 8);
  return 0;
}
