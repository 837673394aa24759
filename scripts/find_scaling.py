"""count / find of 10^7 queries against tables of growing size: where does the probe time come from?  A table that fits the L2 /
Infinity Cache answers from on-die memory; if the time does not drop there, the kernel is bound by issue / address processing, not by HBM."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
nq = 10_000_000
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
for n in (100_000, 800_000, 6_000_000, 25_000_000, 100_000_000):
    keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    dk = torch.from_numpy(keys.view(np.int64)).cuda(); t.insert(dk, torch.from_numpy(vals.view(np.int32)).cuda())
    rng = np.random.default_rng(3)
    dq = dk[torch.from_numpy(rng.integers(0, n, nq)).cuda()].contiguous()
    dm = torch.from_numpy((W.distinct_u64(nq, seed=99) | np.uint64(1 << 63)).view(np.int64)).cuda()
    tc = min(timed(lambda: t.count(dq))[0] for _ in range(4)); tf = min(timed(lambda: t.find(dq))[0] for _ in range(4)); tm = min(timed(lambda: t.count(dm))[0] for _ in range(4))
    print("n %9d cap %10d (%7.1f MB) load %.2f: count hits %.3f ms, find %.3f ms, count misses %.3f ms" % (n, t.capacity(), t.capacity() * 16 / 1e6, n / t.capacity(), tc, tf, tm), flush=True)
    t.close()
