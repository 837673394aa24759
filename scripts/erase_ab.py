"""Robin Hood batch erase, 10^7 of 10^8 keys: per-kernel times of the streaming form (erase keys partitioned by chunk, dropped inside
the one-launch re-layout) and, with KH_DISABLE_STREAM_ERASE=1, of the mark + re-layout form"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n, nq = 100_000_000, 10_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
dq = dk[:nq].clone()
for rep in range(4):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.insert(dk, dv)
    t.profile_enable(True); t.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ne = t.erase(dq)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    p = t.profile()
    print("erase %d: wall %.3f ms, kernels %.3f ms  %s" % (ne, wall, sum(v[1] for v in p.values()), {k: round(v[1], 3) for k, v in sorted(p.items(), key=lambda kv: -kv[1][1])}), flush=True)
    if rep == 0:      # the table afterwards: the erased keys are gone, every other key still carries its value
        c = t.count(dk)
        assert int(c[:nq].sum()) == 0 and int(c[nq:].sum()) == n - nq and t.size() == n - nq
        v, f = t.find_values(dk[nq:])
        assert bool(f.all()) and bool((v == dv[nq:]).all())
    t.close()
