# Batch pass time by number of histogram parts and histogram form: bash scripts/ab_batch_parts.sh  (GPU box)
for REP in 1 2; do
for T in 1000000 1000; do
for P in 2 4 8; do
  JPEGENC_HIST_TILES_MIN_BLOCKS=$T JPEGENC_BATCH_PARTS=$P timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload batch_1080p_420_q90 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles_min $T parts $P', d['ms_per_step'], d['stage_ms'])" || exit 1
done
done
done
