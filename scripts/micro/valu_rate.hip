// Issue rate of the VALU instructions the transform kernels are made of, per SIMD and per
// resident wave count: cycles per wave-instruction for v_fma_f32, v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32 and v_cvt_f32_ubyte0, as independent streams (8 accumulators) and as one
// dependent chain.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int kOp, bool kChain>
__global__ __launch_bounds__(64) void rate(unsigned long long* out, float* sink, int iters) {
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f,
       a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned u = threadIdx.x * 2654435761u;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (kOp == 0) {        // v_fma_f32
            if (kChain) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0.x) : "v"(m.x), "v"(c.x));) }
            else {
                asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                             : "v"(m.x), "v"(c.x));
            }
        } else if (kOp == 1) {  // v_pk_fma_f32
            if (kChain) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));) }
            else {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));
            }
        } else if (kOp == 2) {  // v_pk_mul_f32
            if (kChain) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m));) }
            else {
                asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\n"
                             "v_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m));
            }
        } else if (kOp == 3) {  // v_pk_add_f32
            if (kChain) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));) }
            else {
                asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(c));
            }
        } else if (kOp == 4) {  // v_cvt_f32_ubyte0 (independent only)
            asm volatile("v_cvt_f32_ubyte0 %0, %8\nv_cvt_f32_ubyte1 %1, %8\nv_cvt_f32_ubyte2 %2, %8\nv_cvt_f32_ubyte3 %3, %8\n"
                         "v_cvt_f32_ubyte0 %4, %8\nv_cvt_f32_ubyte1 %5, %8\nv_cvt_f32_ubyte2 %6, %8\nv_cvt_f32_ubyte3 %7, %8"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                         : "v"(u));
        } else if (kOp == 5) {  // v_add_f32
            if (kChain) { REP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0.x) : "v"(c.x));) }
            else {
                asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8"
                             : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                             : "v"(c.x));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 12345.678f) sink[0] = s.x;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int kOp, bool kChain>
void run(const char* name, int wavesPerSimd) {
    const int blocks = 1024 * wavesPerSimd, iters = 2000;
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, blocks * 16);
    hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate<kOp, kChain>), dim3(blocks), dim3(64), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
    }
    unsigned long long* h = new unsigned long long[2 * blocks];
    hipMemcpy(h, d, blocks * 16, hipMemcpyDeviceToHost);
    double sum = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { sum += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double perWave = sum / blocks / (iters * 8.0);
    printf("%-16s %-11s waves/SIMD %d : %.2f counts per instruction per wave, %.2f per SIMD; counter runs at %.0f MHz; %.2f ns per instruction per SIMD\n", name,
           kChain ? "dependent" : "independent", wavesPerSimd, perWave, perWave / wavesPerSimd, sum / real * 100.0, real / blocks / (iters * 8.0) * 10.0 / wavesPerSimd);
    delete[] h;
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0, false>("v_fma_f32", w);
        run<0, true>("v_fma_f32", w);
        run<5, false>("v_add_f32", w);
        run<1, false>("v_pk_fma_f32", w);
        run<1, true>("v_pk_fma_f32", w);
        run<2, false>("v_pk_mul_f32", w);
        run<3, false>("v_pk_add_f32", w);
        run<3, true>("v_pk_add_f32", w);
        run<4, false>("v_cvt_f32_ubyteN", w);
    }
    return 0;
}
