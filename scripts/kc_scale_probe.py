"""where the time of ONE configs[4]-sized counting insert goes (1.67e9 k-mers drawn from 3e8 distinct ones, farm hash): library
per-kernel HIP-event times + wall clock, first call (allocations) and steady state"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_670_000_000
u = int(float(sys.argv[2])) if len(sys.argv) > 2 else 300_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
uni = torch.randint(-(1 << 62), 1 << 62, (u,), dtype=torch.int64, device=dev, generator=g) & ((1 << 62) - 1)
t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm", seed=43)
t.profile_enable(True)
for rep in range(3):
    idx = torch.randint(0, u, (n,), dtype=torch.int64, device=dev, generator=g)
    km = uni[idx]
    del idx
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    new = t.insert_reduce_plus(km)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    prof = t.profile(); t.profile_reset()
    ksum = sum(v[1] for v in prof.values())
    print("rep %d: %d k-mers, %d new, capacity %d: wall %.3f s, kernels %.3f s  ->  %.2e k-mers/s" % (rep, n, new, t.capacity(), wall, ksum / 1e3, n / wall), flush=True)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print("    %-18s %4d launches %10.3f ms" % (k, v[0], v[1]), flush=True)
    del km
