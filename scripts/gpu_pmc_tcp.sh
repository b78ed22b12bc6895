# Vector-cache side of a workload's kernels: requests the L1s send to the L2, their latency, L2 hits:
#   bash scripts/gpu_pmc_tcp.sh <workload>     (on the GPU box; results under gpurun_out/tcp)
set -e
WL=${1:-8k_noise_q100_444}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tcp
mkdir -p $OUT
for PASS in "tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "tcp2 TCP_TOTAL_READ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "ta TA_BUSY_avr TA_ADDR_STALL_BY_TC_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"; do
  set -- $PASS
  NAME=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${WL}_$NAME -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/${WL}_$NAME.log 2>&1
  python3 - $OUT/${WL}_$NAME <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "jpeghip" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-34:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-36s %-36s %14.0f  (%d launches)" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
done
