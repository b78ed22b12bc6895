for G in 768 1024 1280 1536 2048 3072; do
  JPEGENC_HIST_GRID=$G timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $G', d['ms_per_step'], d['stage_ms'])"
done
