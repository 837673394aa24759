#!/bin/bash
# usage (GPU box): bash scripts/ab_env.sh VAR=a VAR=b ...   -- bench.py (configs[1], no CPU legs, no extras) once per environment setting, interleaved
# three times on the same box; prints insert / find ms and the kernel sum (the difference is host-side gaps)
OUT=gpurun_out
mkdir -p $OUT
for round in 1 2 3; do
  for kv in "$@"; do
    env $kv timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 8 > $OUT/abenv_$round.json 2> $OUT/abenv_$round.err || { echo "$kv failed"; tail -3 $OUT/abenv_$round.err; }
    python3 - "$kv" $round <<PY
import json, sys
d = json.load(open("$OUT/abenv_$round.json"))
k = d["kernels_ms_per_step"]
ins = sum(v for n, v in k.items() if n != "k_find")
print("%-18s round %s: insert %.3f ms (kernels %.3f) find %.3f ms (kernel %.3f) step %.3f" % (sys.argv[1], sys.argv[2], d["insert_ms"], ins, d["find_ms"], k["k_find"], d["ms_per_step"]))
PY
  done
done
