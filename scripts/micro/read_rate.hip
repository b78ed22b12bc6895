// How fast can a kernel that only reads stream 805 MB (the coefficients of a 16384 x 16384
// 4:2:0 image) from memory?  The floor under the histogram pass and the bit writer's first pass.
// Forms: 16 bytes per lane, contiguous per wave (1 KB per load instruction), kAhead loads in
// flight per lane; grid-stride with a persistent grid, or one pass per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o read_rate read_rate.hip && ./read_rate
#include <hip/hip_runtime.h>

#include <cstdio>

template <int kAhead>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ src, long long n16, unsigned* sink) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (kAhead - 1) * stride < n16; i += kAhead * stride) {
        uint4 d[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) d[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) acc += d[u].x ^ d[u].y ^ d[u].z ^ d[u].w;
    }
    for (; i < n16; i += stride) {
        const uint4 d = src[i];
        acc += d.x ^ d.y ^ d.z ^ d.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// One lane per 128-byte block (the one-lane-per-block kernels' pattern): eight 16-byte loads, 128 bytes apart from lane to lane.
__global__ __launch_bounds__(256) void rd_block(const uint4* __restrict__ src, long long blocks, unsigned* sink) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < blocks; b += stride) {
        uint4 d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = src[b * 8 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += d[u].x ^ d[u].y ^ d[u].z ^ d[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// Persistent grid, every workgroup reads a contiguous region of its own, 4 KB * kAhead per step.
template <int kAhead>
__global__ __launch_bounds__(256) void rd_region(const uint4* __restrict__ src, long long n16, unsigned* sink) {
    unsigned acc = 0;
    const long long per = (n16 / gridDim.x) & ~(long long)(256 * kAhead - 1);
    const uint4* p = src + per * blockIdx.x + threadIdx.x;
    for (long long i = 0; i < per; i += 256 * kAhead) {
        uint4 d[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) d[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) acc += d[u].x ^ d[u].y ^ d[u].z ^ d[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// Persistent grid, every wave reads a contiguous region of its own, 1 KB * kAhead per step.
template <int kAhead>
__global__ __launch_bounds__(256) void rd_wave_region(const uint4* __restrict__ src, long long n16, unsigned* sink) {
    unsigned acc = 0;
    const long long waves = (long long)gridDim.x * 4;
    const long long per = (n16 / waves) & ~(long long)(64 * kAhead - 1);
    const uint4* p = src + per * (blockIdx.x * 4 + (threadIdx.x >> 6)) + (threadIdx.x & 63);
    for (long long i = 0; i < per; i += 64 * kAhead) {
        uint4 d[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) d[u] = p[i + u * 64];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) acc += d[u].x ^ d[u].y ^ d[u].z ^ d[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <class F>
static float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20;
}

int main() {
    const long long bytes = 805306368ll, n16 = bytes / 16;
    uint4* src;
    unsigned* sink;
    hipMalloc(&src, bytes);
    hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes);
    for (int grid : {1024, 2048, 4096, 8192}) {
        const float t1 = timeit([&] { hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float t2 = timeit([&] { hipLaunchKernelGGL(rd<2>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float t4 = timeit([&] { hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float t8 = timeit([&] { hipLaunchKernelGGL(rd<8>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float tb = timeit([&] { hipLaunchKernelGGL(rd_block, dim3(grid), dim3(256), 0, 0, src, n16 / 8, sink); });
        const float r1 = timeit([&] { hipLaunchKernelGGL(rd_region<1>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float r4 = timeit([&] { hipLaunchKernelGGL(rd_region<4>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float w1 = timeit([&] { hipLaunchKernelGGL(rd_wave_region<1>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        const float w4 = timeit([&] { hipLaunchKernelGGL(rd_wave_region<4>, dim3(grid), dim3(256), 0, 0, src, n16, sink); });
        printf("grid %7d  wave-region1 %.1f us  wave-region4 %.1f us\n", grid, w1 * 1e3, w4 * 1e3);
        printf("grid %7d  region1 %.1f us  region4 %.1f us\n", grid, r1 * 1e3, r4 * 1e3);
        printf("grid %7d  ahead1 %.1f us %.0f GB/s | ahead2 %.1f us %.0f | ahead4 %.1f us %.0f | ahead8 %.1f us %.0f | per-block %.1f us %.0f\n",
               grid, t1 * 1e3, bytes / t1 / 1e6, t2 * 1e3, bytes / t2 / 1e6, t4 * 1e3, bytes / t4 / 1e6, t8 * 1e3, bytes / t8 / 1e6,
               tb * 1e3, bytes / tb / 1e6);
    }
    return 0;
}
