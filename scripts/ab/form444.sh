# Transform kernel of the 4:4:4 workloads under the two forms (one wave / three waves per 64 block positions).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for WL in ${WLS:-4k_444_q90}; do
for f in 1 3; do
  export JPEGENC_444_FORM=$f
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f444_$f -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/f444.log 2>&1 || { tail -5 gpurun_out/f444.log; exit 1; }
  python3 - $f $WL <<'PY'
import csv,glob,sys,json
f=sorted(glob.glob('gpurun_out/f444_%s/*/*kernel_stats.csv' % sys.argv[1]))[-1]
for r in csv.DictReader(open(f)):
    if 'k_transform' in r['Name']: print(sys.argv[2], 'form', sys.argv[1], r['Name'].split('(')[0][-30:], round(float(r['AverageNs'])/1e3,2), 'us')
d=json.loads([l for l in open('gpurun_out/f444.log') if l.startswith('{')][-1]); print('   step', d['ms_per_step'], d['stage_ms'])
PY
done
done
