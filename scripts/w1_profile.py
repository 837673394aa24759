"""per-kernel profile of one W1-shaped insert (x5.5 multiplicity, 10^8 pairs: the reference benchmark's own default input)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
keys, vals = W.w1_benchmark_hashtables(n, seed=23)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
for name, cls in (("robinhood", kh.hashmap_robinhood_doubling), ("linearprobe", kh.hashmap_linearprobe_doubling)):
    for rep in range(3):
        t = cls(128, 0.35, 0.8)
        t.profile_enable(True)
        ni = t.insert(dk, dv); torch.cuda.synchronize()
        p = t.profile(); cap = t.capacity(); t.close()
    tot = sum(v[1] for v in p.values())
    print(name, "distinct", ni, "capacity", cap, "kernel sum %.3f ms" % tot)
    for k, v in sorted(p.items(), key=lambda kv: -kv[1][1]): print("   %-20s x%d %.3f ms" % (k, v[0], v[1]))
