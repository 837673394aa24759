#!/bin/bash
# usage (GPU box, repo root): bash scripts/gpu_round_check.sh <tag>
# GPU test suite (it holds the forced-collectives runs of BOTH sharded layers over RCCL: tests/cpp/test_dist.cpp and
# tests/test_gpu_dist_nccl.py), then bench.py through its launcher (N=1; the JSON line carries the five phases for both tables, the
# host-inclusive pass, W1, the second batch and configs[2]), the loud failure of --gpus 2 on a one-GPU box, and the N>1 code path of
# bench.py rehearsed with a single RCCL rank (every collective executed as a self-exchange).  Steps are joined with && : nothing runs
# after a failed GPU step.
TAG=${1:-r3}
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_$TAG.log 2>&1 && \
timeout -k 10 400 python3 bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err && \
{ timeout -k 10 120 python3 bench.py --gpus 2 --no-cpu-baseline > $OUT/bench2_$TAG.json 2> $OUT/bench2_$TAG.err; echo "gpus2 rc=$?" > $OUT/bench2_$TAG.rc; } && \
KH_DIST_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --chunks 4 --keys 100000000 > $OUT/bench_fd_$TAG.json 2> $OUT/bench_fd_$TAG.err
rc=$?
tail -3 $OUT/gputest_$TAG.log; cat $OUT/bench2_$TAG.rc 2>/dev/null; head -c 600 $OUT/bench_$TAG.json 2>/dev/null; echo; echo "round_check $TAG rc=$rc"
exit $rc
