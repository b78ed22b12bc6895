for T in 256 64; do
  JPEGENC_GROUP_TILE=$T timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload 4k_444_q90 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4k tile $T', d['ms_per_step'], d['stage_ms'])"
  JPEGENC_GROUP_TILE=$T timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --band-of 32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band-of-32 tile $T', d['ms_per_step'], d['stage_ms'])"
done
