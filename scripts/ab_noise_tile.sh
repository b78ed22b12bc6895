# Noise workloads: one-lane-per-block writer (0) against the group writer with small tiles: bash scripts/ab_noise_tile.sh  (GPU box)
for WL in 8k_noise_q100_420 8k_noise_q100_444; do
for T in 0 16 64; do
  JPEGENC_GROUP_TILE=$T timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL tile $T', d['ms_per_step'], d['stage_ms'])" || echo "$WL tile $T failed"
done
done
