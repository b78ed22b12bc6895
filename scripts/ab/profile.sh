cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=jpeg-encoder_amd/lib/libjpegenc_hip.so
cp $L /tmp/keep.so
for v in old new; do
  cp scripts/ab/$v.so $L
  for WL in 1080p_420_q90 16k_420_q90; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abprof_${v}_$WL -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --pipelined-steps 0 --workload $WL > gpurun_out/abprof.log 2>&1
    echo "$v $WL $(grep -h 'k_stuff_bytes\|k_prepare' gpurun_out/abprof_${v}_$WL/*/*kernel_stats.csv | cut -d, -f1,3,4 | cut -c1-40,100-160 | tr '\n' ' ')"
  done
done
cp /tmp/keep.so $L
