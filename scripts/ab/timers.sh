# One run of a workload with the instrumented library scripts/ab/timers.so (prints cycle counts per phase of the dense bit writer).
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so; cp scripts/ab/timers.so $L
timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipelined-steps 0 --workload ${1:-8k_noise_q100_444} > gpurun_out/timers.log 2>&1
cp /tmp/keep.so $L
grep "^wg" gpurun_out/timers.log | tail -6
