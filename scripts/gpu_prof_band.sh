set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_band -- python3 bench.py --pipelined-steps 0 --steps 10 --warmup 3 --no-cpu-baseline --workload 16k_420_q90 --band-of 8 > gpurun_out/prof_band.log 2>&1
python3 scripts/trace_timeline.py prof_band
