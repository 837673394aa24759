"""single-key insert / update / erase on a table of 1e8 elements (host round trip included): in-place path vs re-layout"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
dk = torch.from_numpy(W.distinct_u64(n, seed=1).view(np.int64)).cuda(); dv = torch.arange(n, device="cuda", dtype=torch.int32)
t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
t.insert(dk, dv)
fresh = W.distinct_u64(200, seed=777)
one = np.zeros(1, dtype=np.uint32)
def timed(f, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("insert(k,v) new key      %8.1f us" % timed(lambda i: t.insert(fresh[i:i + 1], one), 100))
print("update(k,v) existing key %8.1f us" % timed(lambda i: t.update(fresh[i:i + 1], one), 100))
print("erase(k) batch form      %8.1f us" % timed(lambda i: t.erase(fresh[i:i + 1]), 100))
print("find(k)                  %8.1f us" % timed(lambda i: t.find(fresh[i:i + 1]), 100))
print("count(k)                 %8.1f us" % timed(lambda i: t.count(fresh[i:i + 1]), 100))
t.close()
