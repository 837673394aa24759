"""host-side overhead of one bench step outside the insert/find calls: table create / destroy, profile readout"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
dk = torch.from_numpy(W.distinct_u64(n, seed=1).view(np.int64)).cuda(); dv = torch.arange(n, device="cuda", dtype=torch.int32)
dq = dk[:10_000_000].clone()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.profile_enable(True)
    t1 = time.perf_counter()
    t.insert(dk, dv)
    t2 = time.perf_counter()
    fk, fv = t.find(dq)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    p = t.profile()
    t4 = time.perf_counter()
    s, c = t.size(), t.capacity()
    t.close()
    t5 = time.perf_counter()
    print("create %.3f  insert %.3f  find %.3f  profile %.3f  close %.3f  total %.3f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)), flush=True)
