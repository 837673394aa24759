#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag>
# rocprofv3 kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench.py workload.
# The program after `--` is python3 itself (no env/bash hop: the profiler initialises the GPU before the program starts).
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
# bench.py run as ONE rank directly (WORLD_SIZE set): under rocprofv3 the launcher form would start its rank as a child of a
# process whose GPU the profiler has already initialised, which the box forbids
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
export TMPDIR=/tmp
cd /tmp
# (the headline legs alone: --no-extras; then ONE more kernel-trace pass with the informational legs -- five phases for both tables,
#  W1, second batch, configs[2] -- whose stats carry k_count / k_erase_fused / k_insert_fused / k_dedup ...: prof_<tag>_phases)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof_$TAG.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_fetch_$TAG.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_write_$TAG.log 2>&1 && \
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_phases -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_${TAG}_phases.log 2>&1
echo "profile_round $TAG rc=$?"
