"""ShardedTable.find / count over RCCL with ONE rank and forced collectives (self-exchange): wall time of 10^7 queries against a 10^8-key
table for 1 / 2 / 4 pieces, and the per-phase device times -- what the Python layer adds per piece"""
import os, sys, time
os.environ["KH_DIST_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, ".")
import numpy as np, torch
import torch.distributed as dist
from kmerhash_amd import dist as khd, workloads as W
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, nq = 100_000_000, 10_000_000
keys = W.distinct_u64(n, seed=1)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.arange(n, dtype=torch.int32, device="cuda")
dq = dk[:nq].clone()
st = khd.ShardedTable(khd.GpuBackend(0), timing=True)
st.insert(dk, dv, chunks=4)
for pieces in (1, 2, 4, 1, 2, 4):
    st.query_pieces = pieces
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pk, fv, ff = st.find(dq); st.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    assert int(ff.sum().item()) == nq
    ph = st.timings()
    print("find %d piece(s): %.3f ms   device ms over 5 reps: %s" % (pieces, best, {k: round(v, 2) for k, v in ph.items()}), flush=True)
dist.destroy_process_group()
