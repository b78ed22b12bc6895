# Host-side AddressSanitizer build of the library, the host classes and the driver (the GPU
# code is not instrumented): bash scripts/asan/build.sh, then run scripts/asan/run.sh on the GPU box.
set -e
cd "$(dirname "$0")/../.."
OUT=scripts/asan/out
mkdir -p $OUT
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Iinclude -Wno-unused-function \
  -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -shared -o $OUT/libjpegenc_hip.so \
  jpeg-encoder_amd/csrc/capi.cpp jpeg-encoder_amd/csrc/sharded.cpp jpeg-encoder_amd/csrc/host_tables.cpp jpeg-encoder_amd/csrc/decode_host.cpp \
  jpeg-encoder_amd/csrc/kernels.hip jpeg-encoder_amd/csrc/entropy.hip jpeg-encoder_amd/csrc/decode_kernels.hip jpeg-encoder_amd/csrc/static_workspace.hip \
  -ldl
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++14 -ffp-contract=off -fPIC -fsanitize=address -fno-omit-frame-pointer -Ijpeg-encoder_amd/host -Iinclude \
  -shared -o $OUT/libjpegenc_host.so jpeg-encoder_amd/host/jpegenc_host.cpp -L$OUT -ljpegenc_hip -lz -Wl,-rpath,'$ORIGIN'
for t in encode_jpeg host_selftest; do
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++14 -ffp-contract=off -fsanitize=address -fno-omit-frame-pointer -Ijpeg-encoder_amd/host -Iinclude \
    -o $OUT/$t jpeg-encoder_amd/host/$t.cpp -L$OUT -ljpegenc_host -ljpegenc_hip -Wl,-rpath,'$ORIGIN'
done
ls -la $OUT
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -Iinclude -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
  -o $OUT/abi_exercise scripts/asan/abi_exercise.cpp -L$OUT -ljpegenc_hip -Wl,-rpath,'$ORIGIN'
