# Stuffing pass: workgroups in the grid: bash scripts/ab_stuff_grid.sh  (GPU box)
for WL in 16k_420_q90 8k_noise_q100_444; do
for G in 1024 1280 1536 2048 2560; do
  JPEGENC_STUFF_GRID=$G timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL grid $G', d['ms_per_step'], d['stage_ms']['entropy'])"
done
done
