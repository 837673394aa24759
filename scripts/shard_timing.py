"""times kh_shard_permute (the device half of the multi-GPU exchange) on one GPU"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from kmerhash_amd import workloads as W
from kmerhash_amd.dist import GpuBackend
n = 100_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
be = GpuBackend(0)
for p in (2, 4, 8, 8, 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ok, ov, c = be.shard(dk, dv, p)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("shard p=%d  %.2f ms  (%.1f GB/s algorithmic)" % (p, dt * 1e3, n * 24 / dt / 1e9), c[:2], flush=True)
