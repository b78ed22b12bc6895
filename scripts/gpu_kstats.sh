# rocprofv3 kernel statistics of one bench workload: bash scripts/gpu_kstats.sh <tag> <bench arguments...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 "$@" > gpurun_out/$TAG.log 2>&1
python3 - $TAG <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/%s/*/*kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Name']:
        print(r['Name'][:70].ljust(72), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
