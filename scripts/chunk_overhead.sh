#!/bin/bash
# per-kernel cost of the streamed (chunked) multi-GPU insert vs the one-shot insert, rehearsed on one GPU
# (KH_DIST_FORCE_COLLECTIVES=1: the collectives run with a single rank, i.e. as local copies)
mkdir -p gpurun_out
for c in 1 4 8; do
  KH_DIST_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 --chunks $c --no-cpu-baseline > gpurun_out/chunks_$c.json 2> gpurun_out/chunks_$c.err || exit 1
  python3 - "$c" <<'PY'
import json, sys
c = sys.argv[1]
d = json.loads(open("gpurun_out/chunks_%s.json" % c).read().strip().splitlines()[-1])
print("chunks", c, "insert_ms %.3f find_ms %.3f step %.3f" % (d["insert_ms"], d["find_ms"], d["ms_per_step"]))
print("   ", {k: v for k, v in d["kernels_ms_per_step"].items()})
PY
done
