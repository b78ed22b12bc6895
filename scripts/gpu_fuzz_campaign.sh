# A longer randomised comparison with the CPU checker than the test suite's, three times: the kernels every picture
# gets by its size, the kernels on groups and tiles for every size, the dense bit writer with tile numbers by XCD for
# every size.  bash scripts/gpu_fuzz_campaign.sh [seeds] [largest side]
set -e
export JPEGENC_FUZZ_SEEDS=${1:-120} JPEGENC_FUZZ_MAX_SIDE=${2:-700}
T="timeout -k 10 900 python -u -m pytest tests/test_gpu_fuzz.py -k random_configurations_byte_identical -x -q --timeout 300 --timeout-method thread"
$T > gpurun_out/fuzz_default.log 2>&1 || { tail -30 gpurun_out/fuzz_default.log; exit 1; }
echo "default: $(tail -1 gpurun_out/fuzz_default.log)"
JPEGENC_GROUP_MIN_BLOCKS=0 JPEGENC_SLOTS_MIN_BLOCKS=0 JPEGENC_HIST_TILES_MIN_BLOCKS=0 $T > gpurun_out/fuzz_groups.log 2>&1 || { tail -30 gpurun_out/fuzz_groups.log; exit 1; }
echo "groups: $(tail -1 gpurun_out/fuzz_groups.log)"
JPEGENC_WRITER=dense JPEGENC_DENSE_PER_XCD_FROM=0 JPEGENC_HIST_TILES_MIN_BLOCKS=0 $T > gpurun_out/fuzz_dense.log 2>&1 || { tail -30 gpurun_out/fuzz_dense.log; exit 1; }
echo "dense: $(tail -1 gpurun_out/fuzz_dense.log)"
