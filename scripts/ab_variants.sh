#!/bin/bash
# usage (GPU box): bash scripts/ab_variants.sh <suffix> [<suffix> ...]   -- bench.py (configs[1], no CPU legs) once per prebuilt library
# variant libkmerhash_amd<suffix>.so ("" = the product), same box, interleaved twice; prints the per-kernel ms per step
OUT=gpurun_out
mkdir -p $OUT
for round in 1 2; do
  for sfx in "$@"; do
    s=$sfx; [ "$s" = "base" ] && s=""
    KH_LIB_SUFFIX=$s timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 8 > $OUT/ab_${sfx}_$round.json 2> $OUT/ab_${sfx}_$round.err || { echo "variant $sfx failed"; tail -3 $OUT/ab_${sfx}_$round.err; }
    python3 - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${sfx}_$round.json"))
    print("%-10s round $round: insert %.3f ms find %.3f ms  %s" % ("$sfx", d["insert_ms"], d["find_ms"], {k: v for k, v in d["kernels_ms_per_step"].items() if v > 0.05}))
except Exception as e:
    print("$sfx: no result", e)
PY
  done
done
