// Write a buffer of X MB, then read it: how fast is the read, by size?  (What the memory-side cache
// and the second-level caches keep of freshly written data.)  Also read-after-read.
//   hipcc --offload-arch=gfx950 -O3 -o mall_size mall_size.hip && ./mall_size
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void wr(uint4* __restrict__ dst, long long n16, unsigned v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ src, long long n16, unsigned* sink) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        const uint4 d = src[i];
        if ((d.x ^ d.y ^ d.z ^ d.w) == 0x12345678u) *sink = 1;
    }
}

int main() {
    uint4* buf;
    unsigned* sink;
    hipMalloc(&buf, 1ll << 30);
    hipMalloc(&sink, 4);
    hipEvent_t a, b, c, d;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventCreate(&c);
    hipEventCreate(&d);
    for (int mb : {8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 768}) {
        const long long n16 = (long long)mb * (1 << 20) / 16;
        const unsigned grid = (unsigned)(n16 / 256);
        float tw = 0, tr = 0, trr = 0;
        for (int it = 0; it < 6; ++it) {
            float ms;
            hipEventRecord(a);
            hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, n16, (unsigned)it);
            hipEventRecord(b);
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            hipEventRecord(c);
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            hipEventRecord(d);
            hipEventSynchronize(d);
            if (it) {
                hipEventElapsedTime(&ms, a, b); tw += ms;
                hipEventElapsedTime(&ms, b, c); tr += ms;
                hipEventElapsedTime(&ms, c, d); trr += ms;
            }
        }
        const double gb = mb * 1.048576e-3;
        printf("%4d MB: write %.1f us (%.0f GB/s), read after it %.1f us (%.0f GB/s), read again %.1f us (%.0f GB/s)\n", mb,
               tw / 5 * 1e3, gb / (tw / 5 * 1e-3), tr / 5 * 1e3, gb / (tr / 5 * 1e-3), trr / 5 * 1e3, gb / (trr / 5 * 1e-3));
    }
    return 0;
}
