#!/bin/bash
# usage (GPU box): bash scripts/profile_kmer_counter.sh <tag> -- rocprofv3 kernel stats of the configs[4]-shaped driver on one GPU
TAG=${1:-kc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/benchmark/kmer_counter.py --reads 3400000 --batches 8 --cycle > $OUT/prof_$TAG.log 2>&1
echo "rc=$?"
python3 - "$OUT/prof_$TAG" <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms %.2f" % (tot / 1e6))
    for r in rows[:22]:
        print("%-60s calls %5s total %8.3f ms avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
