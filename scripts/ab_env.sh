# A/B of an environment switch over bench workloads, alternating: bash scripts/ab_env.sh VAR=VALUE workload...
set -e
SW=$1; shift
for wl in "$@"; do
  for rep in 1 2; do
    for mode in base switch; do
      if [ $mode = switch ]; then export $SW; else unset ${SW%%=*}; fi
      timeout -k 10 200 python -u bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipelined-steps 0 --workload $wl > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
      python3 - $wl $mode <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab.log') if l.startswith('{')][-1])
print('%-20s %-6s step %.4f  %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['stage_ms']))
PY
    done
  done
done
