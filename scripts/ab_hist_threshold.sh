# Histogram pass: one lane per block against one lane per group, by image size: bash scripts/ab_hist_threshold.sh  (GPU box)
for WL in 1080p_420_q90 4k_444_q90; do
for T in 100000000 0; do
  JPEGENC_HIST_TILES_MIN_BLOCKS=$T timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $WL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL tiles_min $T', d['ms_per_step'], d['stage_ms'])"
done
done
for N in 16 32 64; do
for T in 100000000 0; do
  JPEGENC_HIST_TILES_MIN_BLOCKS=$T timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --band-of $N 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band-of-$N tiles_min $T', d['ms_per_step'], d['stage_ms'])"
done
done
