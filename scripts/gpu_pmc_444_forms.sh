# Kernel statistics and the two SQ counter passes of the 4K 4:4:4 transform in both of its forms
# (JPEGENC_444_FORM=1: one wave per 64 block positions, the default at this size; 3: three waves):
#   bash scripts/gpu_pmc_444_forms.sh   (GPU box; results under gpurun_out/forms444)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/forms444
mkdir -p $OUT
for FORM in 1 3; do
  export JPEGENC_444_FORM=$FORM
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_form$FORM -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload 4k_444_q90 > $OUT/stats_form$FORM.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_form$FORM -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload 4k_444_q90 > $OUT/sq_form$FORM.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq2_form$FORM -- python3 bench.py --pipelined-steps 0 --steps 2 --warmup 1 --no-cpu-baseline --workload 4k_444_q90 > $OUT/sq2_form$FORM.log 2>&1
  echo "done form $FORM"
done
