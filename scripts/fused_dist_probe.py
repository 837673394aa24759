"""why is k_build_fused slower inside the (rehearsed) multi-GPU insert than in the plain insert?  (1 GPU)"""
import os, sys
os.environ["KH_DIST_FORCE_COLLECTIVES"] = "1"
os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29544"})
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
import kmerhash_amd as kh
from kmerhash_amd import workloads as W, dist as khd

n = 100_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dev = torch.device("cuda", 0)
dk = torch.from_numpy(keys.view(np.int64)).to(dev); dv = torch.from_numpy(vals.view(np.int32)).to(dev)

def run(label, fn, reps=5):
    acc = {}
    for r in range(reps + 1):
        table, go = fn()
        if r: table.profile_enable(True)
        go()
        torch.cuda.synchronize()
        if r:
            for k, (c, ms) in table.profile().items():
                a = acc.setdefault(k, [0, 0.0]); a[0] += c; a[1] += ms
        table.close()
    print(label, {k: round(v[1] / reps, 3) for k, v in sorted(acc.items()) if v[1] / reps > 0.05}, flush=True)

def plain():
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    return t, lambda: t.insert(dk, dv)
run("A plain            ", plain)

dist.init_process_group("nccl", device_id=dev)
run("B plain after init ", plain)

def sharded():
    be = khd.GpuBackend(0, "rh", 128, 0.35, 0.8, "murmur3avx64", 43)
    t = khd.ShardedTable(be)
    return be.table, lambda: t.insert(dk, dv)
run("C sharded chunks=1 ", sharded)

def shard_only():
    be = khd.GpuBackend(0, "rh", 128, 0.35, 0.8, "murmur3avx64", 43)
    def go():
        ok, ov, sc = be.shard(dk, dv, 1)
        be.table.insert(ok, ov)
    return be.table, go
run("D shard + insert   ", shard_only)

hold = [torch.empty(150_000_000, dtype=torch.int64, device=dev) for _ in range(2)]
run("E plain, +2.4GB held", plain)
dist.destroy_process_group()
