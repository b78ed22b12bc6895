# Bit writer: thresholds of the group form and of the deferred placement, by image size: bash scripts/ab_writer_threshold.sh  (GPU box)
run() {
  timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], d['stage_ms']['entropy'])"
}
for ARGS in "--workload 4k_444_q90" "--band-of 16" "--band-of 32" "--band-of 64" "--workload 1080p_420_q90"; do
  TAG="$ARGS default" run $ARGS
  TAG="$ARGS slots_from_0" JPEGENC_SLOTS_MIN_BLOCKS=0 run $ARGS
  TAG="$ARGS groups_from_0" JPEGENC_GROUP_MIN_BLOCKS=0 run $ARGS
  TAG="$ARGS groups_from_0+slots_from_0" JPEGENC_GROUP_MIN_BLOCKS=0 JPEGENC_SLOTS_MIN_BLOCKS=0 run $ARGS
  TAG="$ARGS blocks" JPEGENC_GROUP_MIN_BLOCKS=100000000 run $ARGS
done
