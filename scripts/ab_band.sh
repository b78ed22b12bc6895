# One rank's band of the 16k image (no all-reduce) under other thresholds: bash scripts/ab_band.sh  (GPU box)
run() {
  timeout -k 10 100 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --band-of $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band-of $1 $2', d['ms_per_step'], d['stage_ms'])"
}
for N in 8 4 2; do
  run $N default
  JPEGENC_HIST_TILES_MIN_BLOCKS=500000 run $N hist_tiles_from_500k
  JPEGENC_SLOTS_MIN_BLOCKS=100000000 run $N placement_not_deferred
done
