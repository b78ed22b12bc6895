# The mixed workload with each form of the bit writer forced: bash scripts/gpu_mixed_tiles.sh
set -e
for t in 256 64 16 0; do
  JPEGENC_GROUP_TILE=$t timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload ${1:-4k2_mixed_q90_420} > gpurun_out/mixed_$t.log 2>&1 || { tail -5 gpurun_out/mixed_$t.log; exit 1; }
  python3 - $t <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/mixed_%s.log' % sys.argv[1]) if l.startswith('{')][-1])
print('tile', sys.argv[1], 'step', d['ms_per_step'], d['stage_ms'], 'runs/img', d['entropy_stage_runs_per_image'])
PY
done
