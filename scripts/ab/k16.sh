# Kernel averages of a workload with the current library: bash scripts/ab/k16.sh <workload>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${1:-16k_420_q90}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k16 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/k16.log 2>&1 || { tail -5 gpurun_out/k16.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/k16/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Name'] and int(r['Calls'])>5: print('  ',r['Name'].split('(')[0][-36:].ljust(38), r['Calls'], round(float(r['AverageNs'])/1000,1))
PY
