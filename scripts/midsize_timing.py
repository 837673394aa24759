"""batches of 10^2..10^6 keys into / out of a Robin Hood table of 10^8 elements (capacity 2^27): in place (regions owned by one
lane each) vs the whole-table re-layout (KH_DISABLE_INPLACE=1).  VERDICT r1 #5: 10^4 keys <= 0.15 ms."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
t.insert(torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals.view(np.int32)).cuda())
fresh = W.distinct_u64(4_000_000, seed=2)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
pos = 0
for m in (100, 1000, 10_000, 100_000, 1_000_000):
    ti, te = [], []
    for rep in range(4):
        k = torch.from_numpy(fresh[pos:pos + m].view(np.int64)).cuda(); v = torch.arange(m, dtype=torch.int32, device="cuda")
        a, ni = timed(lambda: t.insert(k, v))
        b, ne = timed(lambda: t.erase(k))
        assert ni == m and ne == m and t.size() == n
        if rep: ti.append(a); te.append(b)
    print("batch %8d: insert %8.3f ms (%7.2f M keys/s)   erase %8.3f ms (%7.2f M keys/s)" % (m, np.median(ti), m / np.median(ti) / 1e3, np.median(te), m / np.median(te) / 1e3), flush=True)
print(t.profile() if False else "capacity %d size %d" % (t.capacity(), t.size()))
for m in (100, 10_000, 100_000):
    t.profile_reset(); t.profile_enable(True)
    k = torch.from_numpy(fresh[3_000_000:3_000_000 + m].view(np.int64)).cuda(); v = torch.arange(m, dtype=torch.int32, device="cuda")
    t.insert(k, v); pi = t.profile(); t.profile_reset()
    t.erase(k); pe = t.profile(); t.profile_enable(False)
    print("batch %d insert kernels (launches, ms):" % m, {a: (b[0], round(b[1], 4)) for a, b in pi.items()})
    print("batch %d erase kernels:" % m, {a: (b[0], round(b[1], 4)) for a, b in pe.items()})
