# Bench lines of all workloads (no profiler, no CPU baseline): ms per step and stage times.
for WL in 16k_420_q90 4k_444_q90 1080p_420_q90 batch_1080p_420_q90 8k_noise_q100_420 8k_noise_q100_444; do
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', d['ms_per_step'], d['stage_ms'], (d.get('pipelined') or {}).get('ms_per_step'))" || exit 1
done
