"""with a KH_LIB_SUFFIX=_trace2 KH_EXTRA_FLAGS="-DKH_TRACE -DKH_TRACE_SRC=2" build: clock64 stamps at the phase boundaries of 512 chunks of k_build_fused<SRC 2>
(insert of 5*10^7 keys into a table holding 5*10^7)"""
import sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W, _capi
n = 100_000_000
dk = torch.from_numpy(W.distinct_u64(n, seed=1).view(np.int64)).cuda(); dv = torch.arange(n, device="cuda", dtype=torch.int32)
lib = _capi.lib() if hasattr(_capi, "lib") else _capi._lib
for r in range(2):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.insert(dk[: n // 2], dv[: n // 2]); t.insert(dk[n // 2:], dv[n // 2:]); torch.cuda.synchronize(); t.close()
buf = (ctypes.c_ulonglong * (512 * 12))()
print("rc", lib.kh_debug_trace(buf))
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 12)[:, :9].astype(np.int64)
d = np.diff(a, axis=1)
names = ["cursor, set zero, table stage (loads + compaction)", "records -> LDS, image init, fold", "home counts (hash)", "scan", "look-back", "placement", "(nodup check: none)", "write-out"]
for i, nm in enumerate(names): print("%-52s median %8.0f mean %8.0f" % (nm, np.median(d[:, i]), d[:, i].mean()))
print("total median", np.median(a[:, 8] - a[:, 0]))
