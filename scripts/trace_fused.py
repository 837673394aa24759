import sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W, _capi
n = 107374184
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
lib = _capi.lib() if hasattr(_capi, "lib") else _capi._lib
for r in range(3):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.insert(dk, dv); torch.cuda.synchronize(); t.close()
buf = (ctypes.c_ulonglong * (512 * 12))()
print("rc", lib.kh_debug_trace(buf))
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 12)[:, :(8 if "lean" in sys.argv[1:] else 9)].astype(np.int64)
d = np.diff(a, axis=1)
names = ["setup(cur load)", "load+stage", "home counts", "scan", "look-back", "placement", "dup check", "write-out"]
if "lean" in sys.argv[1:]:      # k_build_lean's stamps (the kernel the benchmark case runs)
    names = ["cursor+records->LDS, hash, counts", "barrier", "scan", "group starts, sort", "dup check", "look-back collect", "image, placement, write-out"]
print("clock64 ticks (100 MHz wall? / shader?) median, mean per phase over 512 chunks:")
for i, nm in enumerate(names): print("%-18s median %8.0f mean %8.0f" % (nm, np.median(d[:, i]), d[:, i].mean()))
print("total median", np.median(a[:, -1] - a[:, 0]))
