#!/bin/bash
# usage: benchmark/run_dist.sh N [flags of benchmark_dist_hashtables...]
# starts N processes of the distributed benchmark driver, one per GPU (the reference is started by mpirun -np N); the RCCL
# communicator id travels through a file that rank 0 writes.
N=${1:?number of GPUs}; shift
DIR=$(cd "$(dirname "$0")" && pwd)
IDF=$(mktemp -u /tmp/khd_id.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
pids=()
for ((r = 0; r < N; r++)); do
  "$DIR/_benchmark_dist_hashtables" --nranks "$N" --rank "$r" --id-file "$IDF" "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
rm -f "$IDF"
exit $rc
