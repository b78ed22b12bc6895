# A/B of the entropy kernels: bash scripts/ab_entropy.sh  (old = one lane per block, new = groups)
for wl in 16k_420_q90 4k_444_q90 batch_1080p_420_q90 8k_noise_q100_420 1080p_420_q90; do
for old in 1 0; do
JPEGENC_OLD_ENTROPY=$old timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $wl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl old=$old', d['ms_per_step'], d['stage_ms'])"
done; done
