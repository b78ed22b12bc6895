# Two planes for the blocks' halves against contiguous blocks, by size of a 4:2:0 band: bash scripts/ab_split_threshold.sh  (GPU box)
for N in 8 12 16 24 32; do
for T in 1000000000 0; do
  JPEGENC_SPLIT_MIN_BLOCKS=$T timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --band-of $N 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band-of-$N split_min $T', d['ms_per_step'], d['stage_ms'])"
done
done
