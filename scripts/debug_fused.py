import sys, numpy as np
sys.path.insert(0, '.')
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 300_000
keys = W.distinct_u64(n, seed=77)
keys[::33] = keys[5::33][: len(keys[::33])]
vals = np.arange(n, dtype=np.uint32)
for kind, cls in (("rh", kh.hashmap_robinhood_doubling), ("lp", kh.hashmap_linearprobe_doubling)):
    for h in ("murmur3avx64", "murmur", "farm", "identity"):
        for rep in range(3):
            g = cls(128, 0.35, 0.8, hash=h, seed=43)
            g.profile_enable(True)
            g.insert(keys, vals)
            p = g.profile()
            print(kind, h, rep, "fused" if "k_dedup" not in p else "GENERAL", flush=True)
            g.close()
