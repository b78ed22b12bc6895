# A/B of builds of the library on one box (scripts/ab/<variant>.so; VARIANTS="old new" by default): bash scripts/ab/run.sh <workload> [rounds]
WL=${1:-4k_444_q90}; N=${2:-3}
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so
for i in $(seq $N); do
  for v in ${VARIANTS:-old new}; do
    cp scripts/ab/$v.so $L
    timeout -k 10 200 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $WL > gpurun_out/ab.log 2>&1 || { tail -3 gpurun_out/ab.log; cp /tmp/keep.so $L; exit 1; }
    python -c "import json; d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]); print('$WL $v', d['ms_per_step'], d['stage_ms'])"
  done
done
cp /tmp/keep.so $L
