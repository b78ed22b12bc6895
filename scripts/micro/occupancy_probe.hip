// How many waves of a kernel the hardware keeps on a SIMD at once, measured rather than
// computed: every workgroup notes the constant-rate clock at its start and end and the place
// it ran (XCC, SE, CU, SIMD); the host counts how many lifetimes overlap on each SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

template <int kRegs>
__global__ __launch_bounds__(64) void probe(unsigned long long* out, float* sink, int spin) {
    float v[kRegs];
#pragma unroll
    for (int i = 0; i < kRegs; ++i) v[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < spin; ++k) {
#pragma unroll
        for (int i = 0; i < kRegs; ++i) v[i] = v[i] * 1.0001f + v[(i + 1) % kRegs];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < kRegs; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long hw = ((unsigned long long)__builtin_amdgcn_s_getreg((20 & 63) | (31 << 11)) << 32) |
                                      __builtin_amdgcn_s_getreg((4 & 63) | (31 << 11));
        unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 3;
        o[0] = t0; o[1] = t1; o[2] = hw;
    }
}

template <int kRegs>
void run(int threads, int blocks, int spin) {
    const int waves = blocks * (threads / 64);
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, (size_t)waves * 24);
    hipMalloc(&sink, 4);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)probe<kRegs>);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<kRegs>, dim3(blocks), dim3(threads), 0, 0, d, sink, spin);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)waves * 3);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    // peak overlap per SIMD
    std::map<std::tuple<int, int, int>, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int w = 0; w < waves; ++w) {
        const unsigned long long hw = h[3 * w + 2];
        const int xcc = (int)((hw >> 32) & 0xf), id = (int)(hw & 0xffffffff);
        const int simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        auto& e = ev[std::make_tuple(xcc, (se << 5) | (sh << 4) | cu, simd)];
        e.push_back({h[3 * w], +1});
        e.push_back({h[3 * w + 1], -1});
    }
    int peakMax = 0;
    long long peakSum = 0;
    for (auto& kv : ev) {
        auto& e = kv.second;
        std::sort(e.begin(), e.end());
        int cur = 0, peak = 0;
        for (auto& x : e) { cur += x.second; peak = std::max(peak, cur); }
        peakMax = std::max(peakMax, peak);
        peakSum += peak;
    }
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, h[3 * w]); tmax = std::max(tmax, h[3 * w + 1]); }
    printf("regs asked %3d  compiled vgprs %3d  threads %3d  blocks %5d : SIMDs used %4zu  peak waves per SIMD max %d mean %.2f  span %.1f us\n",
           kRegs, fa.numRegs, threads, blocks, ev.size(), peakMax, (double)peakSum / ev.size(), (tmax - tmin) / 100.0);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int threads : {64}) {
        const int blocks = 2048 * 4 * 64 / threads;
        run<80>(threads, blocks, 50);
        run<120>(threads, blocks, 40);
        run<150>(threads, blocks, 30);
        run<200>(threads, blocks, 20);
        run<240>(threads, blocks, 20);
    }
    return 0;
}
