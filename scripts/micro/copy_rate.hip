// Copy-rate microbenchmark: what does the chip sustain when a kernel reads N bytes and
// writes N bytes (the traffic shape of the fused transform: 805 MB in, 805 MB out)?
// Build: hipcc -O3 --offload-arch=gfx950 copy_rate.hip -o copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <bool kNtLoad, bool kNtStore>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n)
{
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * 256 < n) v[j] = kNtLoad ? __builtin_nontemporal_load(src + i + j * 256) : src[i + j * 256];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * 256 < n) {
                if (kNtStore) __builtin_nontemporal_store(v[j], dst + i + j * 256);
                else dst[i + j * 256] = v[j];
            }
    }
}

template <bool kNtLoad, bool kNtStore>
static float run(const u32x4* src, u32x4* dst, long n, int grid)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy<kNtLoad, kNtStore>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_copy<kNtLoad, kNtStore>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const size_t bytes = 805306368;   // 16384 x 16384 x 3
    u32x4 *src, *dst;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) return 1;
    (void)hipMemset(src, 1, bytes);
    const long n = (long)(bytes / 16);
    const int grids[] = {1024, 2048, 4096, 16384, 49152};
    for (int grid : grids) {
        const float t[4] = {run<false, false>(src, dst, n, grid), run<true, false>(src, dst, n, grid),
                            run<false, true>(src, dst, n, grid), run<true, true>(src, dst, n, grid)};
        printf("grid %6d:", grid);
        const char* names[4] = {"plain", "nt-load", "nt-store", "nt-both"};
        for (int i = 0; i < 4; ++i) printf("  %s %.3f ms %.2f TB/s", names[i], t[i], 2.0 * bytes / t[i] * 1e-9);
        printf("\n");
    }
    return 0;
}
