# Kernel timeline of one step under rocprofv3: bash scripts/gpu_timeline.sh <tag> <bench arguments...>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --pipelined-steps 0 --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/$TAG.log 2>&1
python3 scripts/trace_timeline.py $TAG
tail -1 gpurun_out/$TAG.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'])"
