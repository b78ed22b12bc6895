// Inverse kernels of the verification decoder (gfx950): dequantization and inverse DCT,
// chroma replication and YCbCr -> RGB, squared error between two images.  They mirror
// the inverse operations the reference carries (DCT::directIDCT, DCT.cpp:46-79;
// convertYCbCrToRGBAVXImpl, SIMD256.h:188-216; the byte rounding of ImageLoader::SavePPM,
// ImageLoader.cpp:168-172) and are built with -ffp-contract=off like the forward path.
#include <hip/hip_runtime.h>

#include "decode.h"

namespace jpeghip {

using u32 = uint32_t;
using u64 = unsigned long long;

__device__ constexpr int kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// One thread per block.  F = coefficient * divisor (natural order), F[0][0] += 1024, then
// DCT::directIDCT: out[x][y] = (2/N) * sum_i sum_j F[i][j]*ci*cj*cos_x(i)*cos_y(j) with the
// products taken left to right and the sum in (i, j) order; finally / 255 to undo the
// scaling in front of the forward transform.
__global__ __launch_bounds__(64) void k_inverse_blocks(DGeom g, const int16_t* __restrict__ coef,
                                                       const float* __restrict__ q, const float* __restrict__ cosTab,
                                                       float* __restrict__ planes) {
    __shared__ float ct[64];    // ct[x * 8 + i] = cos((2x + 1) * i * pi / 16)
    __shared__ float qs[192];
    ct[threadIdx.x] = cosTab[threadIdx.x];
    for (int i = threadIdx.x; i < 192; i += 64) qs[i] = q[i];
    __syncthreads();
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    if (b >= g.totalBlocks) return;
    const long long m = b / g.bpm;
    const int k = (int)(b - m * g.bpm);
    const int c = k < g.off1 ? 0 : (k < g.off2 ? 1 : 2);
    const int kk = k - (c == 0 ? 0 : (c == 1 ? g.off1 : g.off2));
    const int by = kk / g.hS[c], bx = kk - by * g.hS[c];
    const int my = (int)(m / g.mcusPerRow), mx = (int)(m - (long long)my * g.mcusPerRow);

    float F[64];
    const int16_t* src = coef + b * 64;
#pragma unroll
    for (int z = 0; z < 64; ++z) F[kNatural[z]] = (float)src[z] * qs[c * 64 + z];
    F[0] += 1024.0f;
    const float sqrtHalf = 0.70710678118654752440f;   // float(1.0 / c_sqrt(2))
    const float halfN = 0.25f;
    const int X0 = (mx * g.hS[c] + bx) * 8, Y0 = (my * g.vS[c] + by) * 8;
    float* plane = planes + g.planeOffset[c];
    for (int x = 0; x < 8; ++x) {        // x: row of the block, as in the reference
        for (int y = 0; y < 8; ++y) {
            float temp = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float ci = i == 0 ? sqrtHalf : 1.0f;
                const float cosi = ct[x * 8 + i];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float cj = j == 0 ? sqrtHalf : 1.0f;
                    const float cosj = ct[y * 8 + j];
                    temp += F[i * 8 + j] * ci * cj * cosi * cosj;
                }
            }
            plane[(long long)(Y0 + x) * g.planeW[c] + X0 + y] = (halfN * temp) / 255.0f;
        }
    }
}

// One thread per pixel of the real image.
__global__ __launch_bounds__(256) void k_planes_to_rgb(DGeom g, const float* __restrict__ planes,
                                                       uint8_t* __restrict__ rgb, long long strideBytes) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.width) return;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        v[c] = planes[g.planeOffset[c] + (long long)(y / g.hf[c]) * g.planeW[c] + x / g.wf[c]];
    // convertYCbCrToRGBAVXImpl: rows of M applied to (Y + 0, Cb - 0.5, Cr - 0.5); the terms
    // with a zero factor are left out (they add +-0).
    const float cb = v[1] + -0.5f, cr = v[2] + -0.5f;
    const float r = v[0] + 1.4021f * cr;
    const float gg = (v[0] + -0.3442f * cb) + -0.7142f * cr;
    const float bl = v[0] + 1.7720f * cb;
    const float out[3] = {r, gg, bl};
    uint8_t* dst = rgb + (long long)y * strideBytes + 3LL * x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int t = (int)(out[c] * 255.0f + 0.5f);   // clamp(0, int(v * 255 + 0.5f), 255)
        t = t < 0 ? 0 : (t > 255 ? 255 : t);
        dst[c] = (uint8_t)t;
    }
}

__global__ __launch_bounds__(256) void k_squared_error(const uint8_t* __restrict__ a, long long strideA,
                                                       const uint8_t* __restrict__ b, long long strideB, int rowBytes,
                                                       int height, u64* __restrict__ sum) {
    u64 acc = 0;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const uint8_t* ra = a + (long long)y * strideA;
        const uint8_t* rb = b + (long long)y * strideB;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < rowBytes; i += gridDim.x * 256) {
            const int d = (int)ra[i] - (int)rb[i];
            acc += (u64)(d * d);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc != 0) atomicAdd(sum, acc);
}

__global__ void k_clear_sum(u64* sum) { *sum = 0; }

hipError_t launch_inverse_blocks(const DGeom& g, const int16_t* d_coef, const float* d_q, const float* d_cos,
                                 float* d_planes, hipStream_t s) {
    hipLaunchKernelGGL(k_inverse_blocks, dim3((unsigned)((g.totalBlocks + 63) / 64)), dim3(64), 0, s, g, d_coef, d_q, d_cos,
                       d_planes);
    return hipGetLastError();
}

hipError_t launch_planes_to_rgb(const DGeom& g, const float* d_planes, uint8_t* d_rgb, long long strideBytes,
                                hipStream_t s) {
    hipLaunchKernelGGL(k_planes_to_rgb, dim3((unsigned)((g.width + 255) / 256), (unsigned)g.height), dim3(256), 0, s, g,
                       d_planes, d_rgb, strideBytes);
    return hipGetLastError();
}

hipError_t launch_squared_error(const uint8_t* d_a, long long strideA, const uint8_t* d_b, long long strideB, int width,
                                int height, u64* d_sum, hipStream_t s) {
    hipLaunchKernelGGL(k_clear_sum, dim3(1), dim3(1), 0, s, d_sum);
    const int rowBytes = width * 3;
    int gx = (rowBytes + 255) / 256;
    if (gx > 64) gx = 64;
    int gy = height < 1024 ? height : 1024;
    hipLaunchKernelGGL(k_squared_error, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, d_a, strideA, d_b, strideB,
                       rowBytes, height, d_sum);
    return hipGetLastError();
}

// ---- synthetic images of the benchmark and the tests ----------------------------------------
// Integer-only, the same bytes on every machine: mix32 is a 32-bit avalanche hash; "noise" is
// byte(x, y, c) = mix32(seed ^ (3 * (y * W + x) + c)) >> 24, "natural" a set of triangle-wave
// colour ramps plus a 4-bit hashed dither.  One thread per byte, rows y0.. of `count` images whose
// seeds are seed, seed + 1, ...
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void k_synthetic_images(unsigned char* __restrict__ out, int kind, int width, int rows,
                                                          int y0, unsigned seed, long long strideBytes,
                                                          long long imageStrideBytes) {
    const int xb = blockIdx.x * 256 + threadIdx.x;   // byte inside the row
    if (xb >= width * 3) return;
    const unsigned y = (unsigned)(y0 + blockIdx.y), x = (unsigned)xb / 3u, c = (unsigned)xb - 3u * x;
    const unsigned idx = (y * (unsigned)width + x) * 3u + c;
    const unsigned h = mix32((seed + blockIdx.z) ^ idx);
    unsigned v;
    if (kind == 0) {
        v = h >> 24;
    } else {
        const unsigned t = (x * (3u + c) + y * (5u - c)) >> 3;
        const int lo = (int)(t & 255u);
        const int tri = ((t >> 8) & 1u) ? 255 - lo : lo;
        int s = tri + (int)(h >> 28) - 8;
        s = s < 0 ? 0 : (s > 255 ? 255 : s);
        v = (unsigned)s;
    }
    out[(long long)blockIdx.z * imageStrideBytes + (long long)blockIdx.y * strideBytes + xb] = (unsigned char)v;
}

hipError_t launch_synthetic_images(unsigned char* d_out, int kind, int width, int rows, int y0, unsigned seed,
                                   long long strideBytes, long long imageStrideBytes, int count, hipStream_t s) {
    // the grid's y and z extents are limited to 65535
    for (int r0 = 0; r0 < rows; r0 += 32768)
        for (int c0 = 0; c0 < count; c0 += 32768) {
            const int nr = rows - r0 < 32768 ? rows - r0 : 32768, nc = count - c0 < 32768 ? count - c0 : 32768;
            hipLaunchKernelGGL(k_synthetic_images, dim3((unsigned)((width * 3 + 255) / 256), (unsigned)nr, (unsigned)nc),
                               dim3(256), 0, s, d_out + (long long)c0 * imageStrideBytes + (long long)r0 * strideBytes, kind,
                               width, nr, y0 + r0, seed + (unsigned)c0, strideBytes, imageStrideBytes);
        }
    return hipGetLastError();
}

}  // namespace jpeghip
