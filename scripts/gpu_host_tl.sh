# Host stations and kernel stages of a step, per workload: bash scripts/gpu_host_tl.sh
set -e
for a in "--workload 1080p_420_q90" "--workload 4k_444_q90" "--workload 16k_420_q90 --band-of 8" "--workload 16k_420_q90" "--workload 8k_noise_q100_444"; do
  timeout -k 10 200 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 $a > gpurun_out/htl.log 2>&1 || { tail -5 gpurun_out/htl.log; exit 1; }
  python3 - "$a" <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/htl.log') if l.startswith('{')][-1])
r=d['per_rank'][0]
print(sys.argv[1], 'step', d['ms_per_step'], 'kern', d['stage_ms'])
print('   host_us', r['host_us'])
PY
done
