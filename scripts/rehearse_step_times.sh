#!/bin/bash
# scripts/rank_step_times.py on N ranks sharing one GPU (gloo).
N=${1:-2}
mkdir -p gpurun_out
export BENCH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29544} WORLD_SIZE=$N
pids=()
for ((r = 0; r < N; ++r)); do
    RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 90 python3 -X faulthandler -u scripts/rank_step_times.py > gpurun_out/steps_rank$r.log 2>&1 &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
for ((r = 0; r < N; ++r)); do echo "== rank $r"; tail -n 8 gpurun_out/steps_rank$r.log; done
exit $rc
