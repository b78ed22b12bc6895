# Kernel averages of the histogram pass under library variants: VARIANTS="a b" bash scripts/ab/hist.sh <workload>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so
for WL in "$@"; do
for v in ${VARIANTS:-cur}; do
  cp scripts/ab/$v.so $L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abh_$v -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abh.log 2>&1 || { tail -5 gpurun_out/abh.log; cp /tmp/keep.so $L; exit 1; }
  python3 - $v $WL <<'PY'
import csv,glob,sys,json
f=sorted(glob.glob('gpurun_out/abh_%s/*/*kernel_stats.csv' % sys.argv[1]))[-1]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'jpeghip' in n and 'histogram' in n:
        out.append('%s %.1f' % (n.split('(')[0].split('::')[-1][:22], float(r['AverageNs'])/1e3))
d=json.loads([l for l in open('gpurun_out/abh.log') if l.startswith('{')][-1])
print(sys.argv[2].ljust(20), sys.argv[1].ljust(8), ' | '.join(out), ' file ok' if d.get('file_equals_gpu_file', d.get('checks')) else '', 'step %.4f' % d['ms_per_step'])
PY
done
done
cp /tmp/keep.so $L
