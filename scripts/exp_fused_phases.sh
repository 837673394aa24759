#!/bin/bash
# EXPERIMENT (timing only, results of the variants are wrong on purpose): which phase of k_build_fused costs what.
# Needs temporary `#if KH_EXP == n` cuts in the kernel (1: no look-back poll, 2: no fold, 3: no write-out, 4: return after the
# look-back); round-1 result at 1e8 keys: 1.45 / 1.35 / 1.12 / 1.18 / 1.10 ms.  KH_EXTRA_FLAGS is honoured by kmerhash_amd/build.py.
mkdir -p gpurun_out
for e in 0 1 2 3 4; do
  KH_EXTRA_FLAGS="-DKH_EXP=$e" python3 -m kmerhash_amd.build > /dev/null 2>&1 || exit 1
  python3 - "$e" <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import kmerhash_amd as kh
from kmerhash_amd import workloads as W
n = 100_000_000
keys = W.distinct_u64(n, seed=1); vals = np.arange(n, dtype=np.uint32)
dk = torch.from_numpy(keys.view(np.int64)).cuda(); dv = torch.from_numpy(vals.view(np.int32)).cuda()
acc = []
for r in range(4):
    t = kh.hashmap_robinhood_doubling(128, 0.35, 0.8)
    t.profile_enable(True)
    try:
        t.insert(dk, dv)
    except Exception as ex:
        pass
    torch.cuda.synchronize()
    p = t.profile()
    if r: acc.append(p.get("k_build_fused", (1, 0))[1])
    t.close()
print("KH_EXP", sys.argv[1], "k_build_fused ms", [round(a, 3) for a in acc], flush=True)
PY
done
python3 -m kmerhash_amd.build > /dev/null 2>&1
