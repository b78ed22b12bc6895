// Does the order in which a kernel reads a large buffer matter when another kernel has just
// written it?  The memory-side cache (256 MB) holds what was written last: a reader that starts
// at the end finds that part there, one that starts at the front finds none of it (its own reads
// have pushed it out by the time it gets there) and competes with the cache's write-backs.
//   hipcc --offload-arch=gfx950 -O3 -o mall_order mall_order.hip && ./mall_order
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void wr(uint4* __restrict__ dst, long long n16, unsigned v) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}

// Workgroup b reads 4 KB pieces b, b + grid, ... counted from the front (kBack = false) or from the end.
template <bool kBack>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ src, long long n16, unsigned* sink) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 d = src[kBack ? n16 - 1 - i : i];
        acc += d.x ^ d.y ^ d.z ^ d.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const long long bytes = 805306368ll, n16 = bytes / 16;
    uint4* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int grid : {2048, 8192, 65536, (int)(n16 / 256)}) {
        float fwd = 0, bwd = 0, again = 0;
        for (int it = 0; it < 6; ++it) {
            float ms;
            hipLaunchKernelGGL(wr, dim3(8192), dim3(256), 0, 0, buf, n16, (unsigned)it);
            hipEventRecord(a);
            hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (it) fwd += ms;
            hipEventRecord(a);   // a second forward pass right behind the first: what the first one left in the cache
            hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (it) again += ms;
            hipLaunchKernelGGL(wr, dim3(8192), dim3(256), 0, 0, buf, n16, (unsigned)it);
            hipEventRecord(a);
            hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n16, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (it) bwd += ms;
        }
        printf("grid %7d: read after write, front first %.1f us, end first %.1f us; end first after a front-first read %.1f us\n", grid,
               fwd / 5 * 1e3, bwd / 5 * 1e3, again / 5 * 1e3);
    }
    return 0;
}
