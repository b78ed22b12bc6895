// Read-rate microbenchmark: how fast can one wave64 per 64 blocks of 128 bytes pull its blocks
// out of HBM, depending on which lane reads which 16-byte piece?
//   A  lane b reads the 8 pieces of block b (lane stride 128 B, every instruction touches 64 lines)
//   B  instruction j reads pieces 64j..64j+63 of the 8 KiB group in order (fully coalesced)
//   C  instruction j, lane (g,i) reads piece i of block 8g+j (8 whole lines per instruction)
//   D  as C, but the piece index comes from lane bits 3-5 (neighbouring lanes in different lines)
//   E  as C, piece index from lane bits 0,4,5 (lane pairs read 32 contiguous bytes)
//   F  as C, piece index from lane bits 0,1,5 (lane quads read 64 contiguous bytes)
// each with ordinary and with non-temporal loads.
// Build: hipcc -O3 --offload-arch=gfx950 load_patterns.hip -o load_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int kPattern, bool kNt>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, long groups, u32* __restrict__ sink)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nWaves = (long)gridDim.x * 4;
    u32 acc = 0;
    for (long g = wave; g < groups; g += nWaves) {
        const u32x4* base = src + g * 512;
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int idx;
            if (kPattern == 0) idx = lane * 8 + j;
            else if (kPattern == 1) idx = j * 64 + lane;
            else if (kPattern == 2) idx = ((lane >> 3) * 8 + j) * 8 + (lane & 7);
            else if (kPattern == 3) idx = ((lane & 7) * 8 + j) * 8 + (lane >> 3);
            else if (kPattern == 4) idx = (((lane >> 1) & 7) * 8 + j) * 8 + ((lane & 1) | (((lane >> 4) & 3) << 1));
            else idx = (((lane >> 2) & 7) * 8 + j) * 8 + ((lane & 3) | (((lane >> 5) & 1) << 2));
            v[j] = kNt ? __builtin_nontemporal_load(base + idx) : base[idx];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        // stand-in for the walk: some dependent ALU work per pass
#pragma unroll 1
        for (int k = 0; k < WORK; ++k) acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int kPattern, bool kNt>
static float run(const u32x4* src, long groups, u32* sink, int grid)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<kPattern, kNt>), dim3(grid), dim3(256), 0, 0, src, groups, sink);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<kPattern, kNt>), dim3(grid), dim3(256), 0, 0, src, groups, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const long blocks = 6291456;               // 16384 x 16384 4:2:0
    const long groups = blocks / 64;
    const size_t bytes = (size_t)blocks * 128;
    u32x4* src; u32* sink;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(src, 1, bytes);
    const int grids[] = {256, 512, 1024, 2048, 24576};
    for (int grid : grids) {
        const float t[12] = {run<0, false>(src, groups, sink, grid), run<0, true>(src, groups, sink, grid),
                             run<1, false>(src, groups, sink, grid), run<1, true>(src, groups, sink, grid),
                             run<2, false>(src, groups, sink, grid), run<2, true>(src, groups, sink, grid),
                             run<3, false>(src, groups, sink, grid), run<3, true>(src, groups, sink, grid),
                             run<4, false>(src, groups, sink, grid), run<4, true>(src, groups, sink, grid),
                             run<5, false>(src, groups, sink, grid), run<5, true>(src, groups, sink, grid)};
        printf("work %d grid %5d (ms, plain/nt):", WORK, grid);
        const char* names = "ABCDEF";
        for (int i = 0; i < 6; ++i) printf("  %c %.3f/%.3f", names[i], t[2 * i], t[2 * i + 1]);
        printf("\n");
    }
    return 0;
}
