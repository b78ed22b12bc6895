# Kernel times of the three forms of the 4:4:4 transform on the 4K workload:
# bash scripts/ab_444.sh   (on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for FORM in 3 1; do
  export JPEGENC_444_FORM=$FORM
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab444_$FORM -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload 4k_444_q90 > gpurun_out/ab444_$FORM.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/ab444_$FORM/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'transform' in r['Name']:
        print('$FORM', r['Name'][:60].ljust(62), r['Calls'], round(float(r['AverageNs'])/1e3,2),'us', 'min', round(float(r['MinNs'])/1e3,2))
PY
done
