#!/bin/bash
# EXPERIMENT (timing only; the variants produce wrong partitions on purpose): what costs what in k_part_scatter.
# Needs temporary `#if KH_EXP == n` cuts in the kernel: 11 reservation without the atomic, 12 return after the reservation,
# 13 return after LDS staging, 14 straight copy-out (pos = tile position: same bytes, no scatter).
mkdir -p gpurun_out
for e in 0 11 12 13 14; do
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
    if r: acc.append(p.get("k_part_scatter", (1, 0))[1] / 2)
    t.close()
print("KH_EXP", sys.argv[1], "k_part_scatter ms per launch", [round(a, 3) for a in acc], flush=True)
PY
done
python3 -m kmerhash_amd.build > /dev/null 2>&1
