# rocprofv3 kernel stats of one bench workload: bash scripts/gpu_prof.sh <workload> <tag>
set -e
WL=${1:-16k_420_q90}; TAG=${2:-prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/$TAG/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Name']:
        print(r['Name'][:70].ljust(72), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
