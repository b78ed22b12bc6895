# Whole GPU suite, then the benchmark workloads with their stage times and the runs of the entropy stage per image.
set -e
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/check_tests.log 2>&1 || { tail -40 gpurun_out/check_tests.log; exit 1; }
tail -2 gpurun_out/check_tests.log
for wl in ${WORKLOADS:-16k_420_q90 4k_444_q90 1080p_420_q90 8k_noise_q100_444 8k_noise_q100_420 4k2_mixed_q90_420 4k2_noise_q75_420 4k2_noise_q75_444}; do
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $wl > gpurun_out/check_$wl.log 2>&1 || { tail -5 gpurun_out/check_$wl.log; exit 1; }
  python3 - $wl <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/check_%s.log' % sys.argv[1]) if l.startswith('{')][-1])
print('%-20s step %.4f  %s  runs/img %s  frac %.3f' % (sys.argv[1], d['ms_per_step'], d['stage_ms'], d['entropy_stage_runs_per_image'], d['roofline']['frac']))
PY
done
