#!/bin/bash
# usage (GPU box, repo root): bash scripts/profile_stream_kernels.sh <tag>
# HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes, as scripts/profile_round.sh) of the kernels only the informational legs of bench.py run:
# k_erase_stream (batch erase as an ordered stream), k_insert_stream (second batch into the loaded table), k_dedup, ...  ->  gpurun_out/pmc_{fetch,write}_<tag>_phases
TAG=${1:-r3f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_${TAG}_phases -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch_${TAG}_phases.log 2>&1 && \
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_${TAG}_phases -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write_${TAG}_phases.log 2>&1
echo "profile_stream_kernels $TAG rc=$?"
