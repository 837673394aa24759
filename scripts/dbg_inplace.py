import sys; sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
k = W.distinct_u64(20000, seed=3); v = np.arange(20000, dtype=np.uint32)
g = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
print(g.insert(k, v), g.capacity())
k2 = W.distinct_u64(50, seed=4); v2 = np.arange(50, dtype=np.uint32)
try:
    print("host", g.insert(k2, v2))
except Exception as e:
    print("ERR host", e)
k3 = W.distinct_u64(50, seed=5)
try:
    print("dev", g.insert(torch.from_numpy(k3.view(np.int64)).cuda(), torch.from_numpy(v2.view(np.int32)).cuda()))
except Exception as e:
    print("ERR dev", e)
print(g.size(), g.profile())
