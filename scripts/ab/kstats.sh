# Kernel averages (rocprofv3) of one workload per library variant: VARIANTS="a b" bash scripts/ab/kstats.sh <workload> <pattern>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${1:-8k_noise_q100_444}; PAT=${2:-encode}
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-old new}; do
  cp scripts/ab/$v.so $L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abks_$v -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abks.log 2>&1
  python3 - $v $PAT <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/abks_%s/*/*kernel_stats.csv' % sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if 'jpeghip' in r['Name'] and sys.argv[2] in r['Name']:
        print(sys.argv[1], r['Name'][:60].ljust(62), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
done
cp /tmp/keep.so $L
