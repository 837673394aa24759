#!/bin/bash
# GPU box: a subset of the parity suite (-k expression $2), then timing scripts and the bench line.  usage: gpu_step.sh <tag> <-k expr>
TAG=${1:-x}; KEXPR=${2:-"mid_size or insert_find_count_erase or random_operation"}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/gputest_$TAG.log 2>&1 && \
timeout -k 10 300 python3 scripts/midsize_timing.py > $OUT/midsize_$TAG.log 2>&1 && \
timeout -k 10 400 python3 scripts/phase_timing.py > $OUT/phases_$TAG.log 2>&1 && \
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rc=$?
tail -25 $OUT/gputest_$TAG.log; cat $OUT/midsize_$TAG.log; cat $OUT/phases_$TAG.log; python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1]); print(d['insert_ms'], d['find_ms'], d['kernels_ms_per_step'])"
echo "gpu_step rc=$rc"; exit $rc
