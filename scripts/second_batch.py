"""cost of inserting into a NON-empty table (general path: de-dup + re-layout of the whole table): two batches of 5e7 keys"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
if len(sys.argv) > 1 and sys.argv[1] == "w1":            # x5.5 multiplicity: the capacity prediction fails, the early vote must catch it
    kk, vv = W.w1_benchmark_hashtables(n, seed=23)
    dk = torch.from_numpy(kk.view(np.int64)).cuda(); dv = torch.from_numpy(vv.view(np.int32)).cuda()
else:
    dk = torch.from_numpy(W.distinct_u64(n, seed=1).view(np.int64)).cuda(); dv = torch.arange(n, device="cuda", dtype=torch.int32)
n1 = n // 2
if "8020" in sys.argv:      # bench.py's second_batch leg: 2*10^7 new keys + 10^6 repeats into 8*10^7 (capacity stays)
    n1 = 80_000_000
    dk = torch.cat([dk[:n1], dk[n1:n1 + 20_000_000], dk[:1_000_000]]); dv = torch.cat([dv[:n1], dv[n1:n1 + 20_000_000], dv[:1_000_000]])
    perm = torch.randperm(21_000_000, device="cuda") + n1
    dk[n1:] = dk[perm]; dv[n1:] = dv[perm]
for rep in range(3):
    t = (kh.hashmap_linearprobe_doubling if "lp" in sys.argv else kh.hashmap_robinhood_doubling)(128, 0.35, 0.8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t.insert(dk[:n1], dv[:n1])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    t.profile_enable(True)
    t.insert(dk[n1:], dv[n1:])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("first: %.2f ms  second (into the loaded table): %.2f ms  cap %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, t.capacity()), {k: round(v[1], 2) for k, v in t.profile().items() if v[1] > 0.05}, flush=True)
    t.close()
