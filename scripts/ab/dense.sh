# Kernel averages of the dense bit writer under library variants and grid sizes: bash scripts/ab/dense.sh <workload>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${1:-8k_noise_q100_444}
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so
run() {
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abd_$1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abd.log 2>&1 || { tail -5 gpurun_out/abd.log; cp /tmp/keep.so $L; exit 1; }
  python3 - $1 <<'PY'
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/abd_%s/*/*kernel_stats.csv' % sys.argv[1]))[-1]
out=[]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'jpeghip' in n and any(k in n for k in ('encode','merge','prepare','stuff')):
        out.append('%s %.1f' % (n.split('(')[0].split('::')[-1][:22], float(r['AverageNs'])/1e3))
print(sys.argv[1].ljust(12), ' | '.join(out))
PY
}
for v in ${VARIANTS:-w4 w3}; do
  cp scripts/ab/$v.so $L
  for grid in ${GRIDS:-1024}; do
    export JPEGENC_DENSE_GRID=$grid
    run ${v}_g$grid
  done
done
unset JPEGENC_DENSE_GRID
cp /tmp/keep.so $L
