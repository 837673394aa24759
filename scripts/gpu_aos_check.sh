#!/bin/bash
# GPU box: the parity suite once with poisoned destination buffers (every slot of a re-layout's destination must be written),
# once plain, then the bench line.
TAG=${1:-r2b}
OUT=gpurun_out
mkdir -p $OUT
KH_DEBUG_POISON=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_poison_$TAG.log 2>&1 && \
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_$TAG.log 2>&1 && \
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rc=$?
tail -15 $OUT/gputest_poison_$TAG.log; tail -5 $OUT/gputest_$TAG.log; head -c 1500 $OUT/bench_$TAG.json; echo "aos_check rc=$rc"
exit $rc
