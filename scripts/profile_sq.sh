#!/bin/bash
# usage (GPU box): bash scripts/profile_sq.sh <tag>  -- SQ counters of the bench workload per kernel: where do the wave cycles go
# (parked on s_waitcnt / barriers, issue stalls, issuing), how many vector / LDS instructions, LDS bank conflicts.
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES (MI355X guide, counter table).  Counters only, no trace domains.
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/pmc_list_$TAG.txt 2>&1
pass() {
  local name=$1; shift
  local WANT=""
  for c in "$@"; do if grep -q "$c" $OUT/pmc_list_$TAG.txt; then WANT="$WANT $c"; fi; done
  echo "pass $name:$WANT"
  timeout -k 10 300 rocprofv3 --pmc $WANT --output-format csv -d $OUT/sq_${name}_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq_${name}_$TAG.log 2>&1 || echo "pass $name failed"
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS && \
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES
python3 - "$TAG" "$OUT" <<'PY'
import sys, glob, csv, collections
tag, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("%s/sq_*_%s/*/*counter_collection.csv" % (out, tag)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in acc.values() for c in v})
top = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]
with open("%s/sq_summary_%s.txt" % (out, tag), "w") as fo:
    for k, v in top:
        line = "%-40s " % k[:40] + " ".join("%s=%.4g" % (c.replace("SQ_", ""), v.get(c, 0)) for c in names)
        print(line); fo.write(line + "\n")
PY
