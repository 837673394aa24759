"""A/B: fused bulk build vs. general path on the bench workload (set KH_DISABLE_FUSED_BUILD=1 for the general path)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
for rep in range(4):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t.insert(dk, dv)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("insert %.3f ms" % (dt * 1e3), {k: round(v[1], 3) for k, v in t.profile().items()}, t.size(), t.capacity(), flush=True)
    t.close()
