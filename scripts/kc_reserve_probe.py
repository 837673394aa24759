"""counting inserts of 4.55e8 k-mers (2.3e8 distinct of a 3e8 universe) per batch into a table PRE-SIZED by reserve() (the --hll-reserve
mode of benchmark/kmer_counter.py): per-kernel times"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
n, u = 455_000_000, 300_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
uni = torch.randint(-(1 << 62), 1 << 62, (u,), dtype=torch.int64, device=dev, generator=g) & ((1 << 62) - 1)
for pre in (0, 330_000_000):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="farm", seed=43)
    t.profile_enable(True)
    for rep in range(3):
        idx = torch.randint(0, u, (n,), dtype=torch.int64, device=dev, generator=g)
        km = uni[idx]
        del idx
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pre:
            t.reserve(pre)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        new = t.insert_reduce_plus(km)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        prof = t.profile(); t.profile_reset()
        print("reserve %d rep %d: %d new, capacity %d: reserve %.3f s insert %.3f s, kernels %.3f s" % (pre, rep, new, t.capacity(), t1 - t0, t2 - t1, sum(v[1] for v in prof.values()) / 1e3), flush=True)
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]:
            print("    %-18s %4d launches %10.3f ms" % (k, v[0], v[1]), flush=True)
        del km
    t.close()
