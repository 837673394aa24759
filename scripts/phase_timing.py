"""device-resident timing of the five benchmark_hashmap phases (insert, find, count, erase, count2; BenchmarkHashTables.cpp:1037-1186)
for both tables: median of 5 repeats, fresh table per repeat (SURVEY 8d timing protocol)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n, nq = 100_000_000, 10_000_000
wl = sys.argv[1] if len(sys.argv) > 1 else "w2"
if wl == "w1":
    keys, vals = W.w1_benchmark_hashtables(n, seed=23)
else:
    keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
dq = dk[:nq].clone()
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
for name, cls in (("robinhood", kh.hashmap_robinhood_doubling), ("linearprobe", kh.hashmap_linearprobe_doubling)):
    rows = []
    for rep in range(6):
        t = cls(128, 0.35, 0.8)
        ti, ni = timed(lambda: t.insert(dk, dv))
        tf, (fk, fv) = timed(lambda: t.find(dq))
        tc, c = timed(lambda: t.count(dq))
        te, ne = timed(lambda: t.erase(dq))
        tc2, c2 = timed(lambda: t.count(dq))
        assert fk.numel() == nq and int(c.sum()) == nq and int(c2.sum()) == 0
        if rep: rows.append((ti, tf, tc, te, tc2))
        cap = t.capacity(); t.close()
    med = np.median(np.array(rows), axis=0); mn = np.min(np.array(rows), axis=0)
    print("[%s %s] distinct %d erased %d capacity(after erase) %d" % (name, wl, ni, ne, cap))
    for lbl, m, lo, cnt in zip(("insert", "find", "count", "erase", "count2"), med, mn, (n, nq, nq, nq, nq)):
        print("  %-7s median %7.3f ms  min %7.3f ms  %8.2f G/s" % (lbl, m, lo, cnt / m / 1e6))
