"""experiment helper: per-kernel times of one 107374184-key insert call of whatever library KH_LIB_SUFFIX selects (results are not checked:
used with experiment builds whose insert stops after the partition)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 107_374_184
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
for rep in range(5):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.profile_enable(True)
    try:
        t.insert(dk, dv)
    except Exception as e:
        print("insert raised:", str(e)[:100])
    torch.cuda.synchronize()
    p = t.profile()
    print({k: (v[0], round(v[1], 4)) for k, v in sorted(p.items(), key=lambda kv: -kv[1][1])}, flush=True)
    t.close()
