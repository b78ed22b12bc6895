set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
for v in 2 3; do
  JPEGHIP_VARIANT=$v timeout -k 10 200 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b16k_v$v.log 2>&1
  python -c "import json,sys; d=json.loads(open('gpurun_out/b16k_v$v.log').read().strip().splitlines()[-1]); print('16k v$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"
done
for v in 2 3 4; do
  JPEGHIP_VARIANT=$v timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload 4k_444_q90 > gpurun_out/b4k_v$v.log 2>&1
  python -c "import json,sys; d=json.loads(open('gpurun_out/b4k_v$v.log').read().strip().splitlines()[-1]); print('4k v$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"
done
