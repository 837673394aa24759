"""experiment: how much of the partition passes' time is the cost of runs that start and end inside a 128-byte line?  Keys crafted (identity hash) so that every
(tile, digit) run of BOTH passes holds exactly 32 records = 384 bytes = three whole lines, against random keys under the same hash; per-kernel times of one insert"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
n = 107_374_184
dev = torch.device("cuda")
i = torch.arange(n, device=dev, dtype=torch.int64)
t = i // 8192; r = i % 8192
d1 = r % 256; j = r // 256
d2 = (j * 8 + (t % 8)) % 256
q = d1 * 256 + d2
# p = bit reversal of the 16-bit q
p = torch.zeros_like(q)
for b in range(16):
    p |= ((q >> b) & 1) << (15 - b)
bucket = (((i * -7046029254386353131) >> 40) & 2047)        # (a multiplicative hash of i: the homes inside a chunk are spread)
crafted = (p << 11) | bucket | (i << 27)
rnd = torch.randint(0, 2**62, (n,), device=dev, dtype=torch.int64)
vals = torch.arange(n, device=dev, dtype=torch.int32)
for name, keys in (("crafted (aligned 384-byte runs)", crafted), ("random", rnd), ("crafted", crafted), ("random", rnd)):
    for rep in range(3):
        tb = kh.hashmap_robinhood_doubling(128, 0.35, 0.8, hash="identity")
        tb.profile_enable(True)
        ni = tb.insert(keys, vals); torch.cuda.synchronize()
        p_ = tb.profile(); tb.close()
    print(name, ni, {k: (v[0], round(v[1], 4)) for k, v in sorted(p_.items(), key=lambda kv: -kv[1][1]) if v[1] > 0.01}, flush=True)
