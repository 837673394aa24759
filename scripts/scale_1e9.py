"""one table, one GPU, 10^9 distinct keys in ONE insert call (BASELINE config 4's total on a single MI355X): size-independent
checks (every key inserted once, capacity by the rule, all queries found, misses not found) + timing"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
t0 = time.time()
dk = torch.empty(n, dtype=torch.int64, device="cuda")
step = 100_000_000
for a in range(0, n, step):                      # generate in slices: bijective splitmix64 of a counter -> distinct keys
    b = min(n, a + step)
    dk[a:b] = torch.from_numpy(W.distinct_u64(b - a, seed=1, start=a).view(np.int64)).cuda()
dv = torch.arange(n, dtype=torch.int64, device="cuda").to(torch.int32)
print("generated %d keys in %.1f s" % (n, time.time() - t0), flush=True)
for rep in range(2):       # the first pass pays for the device allocations (tens of GB: ~1.4 s at 10^9 keys), the pool keeps them
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.profile_enable(True)
    torch.cuda.synchronize(); t1 = time.time()
    ni = t.insert(dk, dv)
    torch.cuda.synchronize(); t2 = time.time()
    if rep == 0:
        print("first insert (cold pool): %.1f ms" % ((t2 - t1) * 1e3), flush=True)
        t.close()
print("insert: %d new, size %d, capacity %d, %.1f ms (%.3g inserts/s)" % (ni, t.size(), t.capacity(), (t2 - t1) * 1e3, n / (t2 - t1)), flush=True)
print({k: round(v[1], 2) for k, v in t.profile().items() if v[1] > 1.0})
assert ni == n and t.size() == n
cap = 128
while int(np.float32(cap) * np.float32(0.8)) < n: cap *= 2
assert t.capacity() == cap, (t.capacity(), cap)
q = dk[:: max(1, n // 10_000_000)][:10_000_000].contiguous()
torch.cuda.synchronize(); t1 = time.time()
fk, fv = t.find(q)
torch.cuda.synchronize(); t2 = time.time()
assert fk.numel() == q.numel() and bool((fk == q).all())
idx = torch.arange(0, n, max(1, n // 10_000_000), device="cuda")[: q.numel()].to(torch.int32)
assert bool((fv == idx).all())
print("find %d hits: %.2f ms" % (q.numel(), (t2 - t1) * 1e3), flush=True)
miss = torch.from_numpy(W.distinct_u64(1_000_000, seed=1, start=n + 5).view(np.int64)).cuda()
assert int(t.count(miss).sum().item()) == 0
h = t.displacement_histogram()
print("max displacement", int(np.flatnonzero(h)[-1]), "mean %.3f" % (float((h * np.arange(128)).sum()) / n))
t.close()
print("OK")
