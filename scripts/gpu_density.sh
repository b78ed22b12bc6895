# Entropy stage across the density range: bash scripts/gpu_density.sh
set -e
for wl in 4k2_noise_q50_420 4k2_noise_q75_420 4k2_noise_q75_444 4k2_noise_q90_420 8k_noise_q100_420; do
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $wl > gpurun_out/dens_$wl.log 2>&1 || { tail -5 gpurun_out/dens_$wl.log; exit 1; }
  python3 - $wl <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/dens_%s.log' % sys.argv[1]) if l.startswith('{')][-1])
blocks = d['config']['width']*d['config']['height']//64*(3 if d['config']['sampling']=='444' else 1.5)
print(sys.argv[1], 'step', d['ms_per_step'], d['stage_ms'], 'runs/img', d['entropy_stage_runs_per_image'], 'bits/block %.0f' % (d['config']['jpeg_bytes_rank0']*8/blocks))
PY
done
