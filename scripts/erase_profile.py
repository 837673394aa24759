import sys
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n, nq = 100_000_000, 10_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
dq = dk[:nq].clone()
for rep in range(3):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.insert(dk, dv)
    t.profile_enable(True); t.profile_reset()
    t.erase(dq)
    torch.cuda.synchronize()
    print({k: round(v[1], 3) for k, v in t.profile().items()}, flush=True)
    t.profile_reset()
    t.rehash(1 << 28)
    torch.cuda.synchronize()
    print("rehash 2^27 -> 2^28:", {k: round(v[1], 3) for k, v in t.profile().items()}, flush=True)
    t.close()
