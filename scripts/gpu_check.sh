# Parity tests, then the two headline benchmarks (no CPU baseline): quick loop for kernel work.
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
for wl in 16k_420_q90 4k_444_q90 8k_noise_q100_420 ${EXTRA_WL}; do
  timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl > gpurun_out/b_$wl.log 2>&1 || { tail -5 gpurun_out/b_$wl.log; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/b_$wl.log').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d['config']['jpeg_bytes_rank0'])"
done
