# The dense bit writer: hash tests of the noise workloads, whole GPU suite, kernel averages.
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_fullsize_hashes.py -x -q --timeout 200 --timeout-method thread -k "noise or mixed" > gpurun_out/dense_t1.log 2>&1 || { tail -30 gpurun_out/dense_t1.log; exit 1; }
tail -1 gpurun_out/dense_t1.log
cp jpeg-encoder_amd/lib/libjpegenc_hip.so scripts/ab/cur.so
VARIANTS=cur GRIDS="${GRIDS:-1024}" bash scripts/ab/dense.sh 8k_noise_q100_444
VARIANTS=cur GRIDS="${GRIDS:-1024}" bash scripts/ab/dense.sh 8k_noise_q100_420
WORKLOADS="8k_noise_q100_444 8k_noise_q100_420 4k2_noise_q75_444" bash scripts/gpu_check2.sh
