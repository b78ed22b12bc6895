# Kernel statistics, bench lines and counter passes of the three profiled workloads, one run each:
#   bash scripts/gpu_final_profiles.sh   (on the GPU box; results under gpurun_out/final)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
mkdir -p $OUT
for WL in 16k_420_q90 4k_444_q90 batch_1080p_420_q90; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload $WL > $OUT/stats_$WL.log 2>&1
  grep '^{' $OUT/stats_$WL.log | tail -1 > $OUT/bench_line_$WL.json
  for PASS in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    set -- $PASS
    NAME=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${WL}_$NAME -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/pmc_${WL}_$NAME.log 2>&1
  done
  echo "done $WL"
done
# plain bench lines (no profiler), with the CPU baseline for the headline workload
timeout -k 10 300 python3 bench.py > $OUT/bench_16k_default.json 2> $OUT/bench_16k_default.err
for WL in 4k_444_q90 batch_1080p_420_q90 1080p_420_q90 8k_noise_q100_420 8k_noise_q100_444; do
  timeout -k 10 300 python3 bench.py --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
# Kernel statistics and the two traffic passes for the remaining workloads.
for WL in 1080p_420_q90 8k_noise_q100_420 8k_noise_q100_444; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload $WL > $OUT/stats_$WL.log 2>&1
  grep '^{' $OUT/stats_$WL.log | tail -1 > $OUT/bench_line_$WL.json
  for PASS in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    set -- $PASS
    NAME=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${WL}_$NAME -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/pmc_${WL}_$NAME.log 2>&1
  done
  echo "done $WL"
done
echo all done
