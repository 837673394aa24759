#!/bin/bash
# usage (GPU box): bash scripts/profile_l2.sh <tag>  -- one more PMC pass of the bench workload: L2 hits / misses and
# memory read requests per kernel (sectors per operation of the probing kernels).  Counter names differ between
# ROCm releases: the list is probed first and only the names that exist are requested.
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
# bench.py run as ONE rank directly (WORLD_SIZE set): under rocprofv3 the launcher form would start its rank as a child of a
# process whose GPU the profiler has already initialised, which the box forbids
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/pmc_list_$TAG.txt 2>&1
WANT=""
for c in TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA_RDREQ_sum TCC_REQ_sum; do
  if grep -q "$c" $OUT/pmc_list_$TAG.txt; then WANT="$WANT $c"; fi
done
echo "counters:$WANT"
for c in $WANT; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${c}_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_${c}_$TAG.log 2>&1 || echo "pass $c failed"
done
python3 - "$TAG" "$OUT" $WANT <<'PY'
import sys, glob, csv, collections
tag, out = sys.argv[1], sys.argv[2]
for c in sys.argv[3:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("%s/pmc_%s_%s/*/*counter_collection.csv" % (out, c, tag)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
        print("%-22s %-28s launches(x dims) %4d  total %.4g" % (c, k[:28], n, v))
PY
