import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n=100_000_000
keys=W.distinct_u64(n,seed=1); vals=np.arange(n,dtype=np.uint32)
dk=torch.from_numpy(keys.view(np.int64)).cuda(); dv=torch.from_numpy(vals.view(np.int32)).cuda()
for parts in (1,1,2,4,8):
    t=kh.hashmap_robinhood_doubling(128,0.35,0.8)
    torch.cuda.synchronize(); t0=time.perf_counter()
    b=[n*i//parts for i in range(parts+1)]
    for i in range(parts): t.insert(dk[b[i]:b[i+1]], dv[b[i]:b[i+1]])
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("parts",parts,"ms %.2f"%(dt*1e3), t.size(), t.capacity(), flush=True)
    t.close()
# reserve first then insert in parts
for parts in (4,):
    t=kh.hashmap_robinhood_doubling(128,0.35,0.8); t.reserve(n)
    torch.cuda.synchronize(); t0=time.perf_counter()
    b=[n*i//parts for i in range(parts+1)]
    for i in range(parts): t.insert(dk[b[i]:b[i+1]], dv[b[i]:b[i+1]])
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("reserved parts",parts,"ms %.2f"%(dt*1e3), t.size(), t.capacity(), flush=True)
