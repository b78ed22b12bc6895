#!/bin/bash
# Rehearses bench.py with several ranks sharing one GPU (gloo instead of RCCL, which refuses
# two ranks on one device).  Usage: scripts/rehearse_ranks.sh N [bench.py arguments...]
# Every rank runs under a time limit; on expiry its Python stack is dumped to its log.
N=${1:-2}; shift
mkdir -p gpurun_out
export BENCH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29533} WORLD_SIZE=$N
pids=()
for ((r = 0; r < N; ++r)); do
    RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 ${REHEARSE_LIMIT:-90} python3 -X faulthandler -u bench.py --gpus $N "$@" \
        > gpurun_out/rehearse_rank$r.log 2>&1 &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for ((r = 0; r < N; ++r)); do echo "== rank $r"; tail -n 25 gpurun_out/rehearse_rank$r.log; done
exit $rc
