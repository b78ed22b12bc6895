// Device kernels of the MI355X (gfx950) JPEG encode hot path.
//
// Build with -ffp-contract=off: the transform must round every multiply and add on its
// own, exactly like the reference's AVX path (SIMD256.h:155-186, 337-441).  Division is
// the compiler's correctly rounded fp32 division, rounding is v_rndne_f32 (ties to even).
#include "kernels.h"

#include "jpegenc_hip.h"

namespace jpeghip {

typedef unsigned long long u64;
typedef unsigned int u32;

// ZigZag.h:10-19
__device__ constexpr int kZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// AAN factors exactly as the reference's constexpr cosine recursion yields them
// (const_math.h:16-41, SIMD256.h:335-352); several differ from cosf() by 1-2 ulp.
#define AAN_A1 0x1.6a09e6p-1f
#define AAN_A2 0x1.1517a8p-1f
#define AAN_A4 0x1.4e7aeap+0f
#define AAN_A5 0x1.87de2cp-2f
#define AAN_S0 0x1.6a09e6p-2f
#define AAN_S1 0x1.0503ecp-2f
#define AAN_S2 0x1.1517a8p-2f
#define AAN_S3 0x1.33e37ap-2f
#define AAN_S4 0x1.6a09e6p-2f
#define AAN_S5 0x1.ccc9aep-2f
#define AAN_S6 0x1.4e7ae8p-1f
#define AAN_S7 0x1.480da4p+0f

// Two floats that travel through the same instructions (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32: two IEEE single operations per lane and instruction, each rounded like its
// scalar counterpart).  The transform kernels issue VALU instructions back to back, so
// halving their number is what makes them faster.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float x) { return (f2){x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// One 8-point AAN pass, operation order of SIMD256.h:354-415.  T: float, or f2 for two
// independent passes at once.
template <class T>
__device__ __forceinline__ void aan1d(T& x0, T& x1, T& x2, T& x3, T& x4, T& x5, T& x6, T& x7) {
    const T t0 = x0 + x7, t1 = x1 + x6, t2 = x2 + x5, t3 = x3 + x4;
    const T t4 = x3 - x4, t5 = x2 - x5, t6 = x1 - x6, t7 = x0 - x7;
    const T r0 = t0 + t3, r1 = t1 + t2, r2 = t1 - t2, r3 = t0 - t3;
    const T r4 = t4 + t5, r5 = t5 + t6, r6 = t6 + t7, r7 = t7;
    const T u0 = r0 + r1, u1 = r0 - r1;
    T u2 = r2 + r3;
    const T z = r6 - r4;
    u2 = u2 * AAN_A1;
    const T za5 = z * AAN_A5;
    const T u4 = (r4 * AAN_A2) - za5;
    const T u5m = r5 * AAN_A1;
    const T u6 = (r6 * AAN_A4) - za5;
    const T y2 = u2 + r3, y6 = r3 - u2;
    const T u5 = u5m + r7, u7 = r7 - u5m;
    x0 = u0 * AAN_S0;
    x4 = u1 * AAN_S4;
    x2 = y2 * AAN_S2;
    x6 = y6 * AAN_S6;
    x5 = (u4 + u7) * AAN_S5;
    x1 = (u5 + u6) * AAN_S1;
    x7 = (u5 - u6) * AAN_S7;
    x3 = (u7 - u4) * AAN_S3;
}

// Rows, then columns, then DC - 1024 (SIMD256.h:426-441).  v is row-major.
__device__ __forceinline__ void aan2d(float (&v)[64]) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
        aan1d(v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7]);
#pragma unroll
    for (int c = 0; c < 8; ++c) aan1d(v[c], v[8 + c], v[16 + c], v[24 + c], v[32 + c], v[40 + c], v[48 + c], v[56 + c]);
    v[0] = v[0] - 1024.0f;
}

// The same on pairs.  In: a[(r / 2) * 8 + c] = (v[r][c], v[r + 1][c]) for even r, i.e. two
// rows side by side, so that a row pass transforms both.  Out: b[r * 4 + c / 2] = (F[r][c],
// F[r][c + 1]), two columns side by side after a column pass that transforms both.  Every
// value sees exactly the operations of aan2d.
__device__ __forceinline__ void aan2d_pairs(f2 (&a)[32], f2 (&b)[32]) {
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
        aan1d(a[8 * rp], a[8 * rp + 1], a[8 * rp + 2], a[8 * rp + 3], a[8 * rp + 4], a[8 * rp + 5], a[8 * rp + 6],
              a[8 * rp + 7]);
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const f2 p1 = a[8 * rp + 2 * cp], p2 = a[8 * rp + 2 * cp + 1];
            b[(2 * rp) * 4 + cp] = (f2){p1.x, p2.x};
            b[(2 * rp + 1) * 4 + cp] = (f2){p1.y, p2.y};
        }
#pragma unroll
    for (int cp = 0; cp < 4; ++cp)
        aan1d(b[cp], b[4 + cp], b[8 + cp], b[12 + cp], b[16 + cp], b[20 + cp], b[24 + cp], b[28 + cp]);
    b[0].x = b[0].x - 1024.0f;
}

// One colour channel of one pixel (SIMD256.h:155-186).  The reference also adds the
// matrix's "0 * 1" term; adding +0.0f can only turn a -0.0 into +0.0, which no later
// step can observe in the integer coefficients, so it is left out.
__device__ __forceinline__ float channel_value(int c, bool convert, float r, float g, float b) {
    if (!convert) return c == 0 ? r : (c == 1 ? g : b);
    const float m0 = c == 0 ? 0.2990f : (c == 1 ? -0.1687f : 0.5000f);
    const float m1 = c == 0 ? 0.5870f : (c == 1 ? -0.3312f : -0.4186f);
    const float m2 = c == 0 ? 0.1140f : (c == 1 ? 0.5000f : -0.0813f);
    float t = m0 * r;
    t = t + m1 * g;
    t = t + m2 * b;
    if (c != 0) t = t + 0.5f;
    return t;
}

__device__ __forceinline__ void load_pixel(const void* rgb, long long stride, int fmt, int x, int y, float& r, float& g,
                                           float& b) {
    if (fmt == JPEG_HIP_RGB8) {   // ImageLoader.cpp:190-193: byte / 255.0f
        const unsigned char* p = (const unsigned char*)rgb + (long long)y * stride + (long long)x * 3;
        r = (float)p[0] / 255.0f;
        g = (float)p[1] / 255.0f;
        b = (float)p[2] / 255.0f;
    } else {
        const float* p = (const float*)((const char*)rgb + (long long)y * stride) + (long long)x * 3;
        r = p[0];
        g = p[1];
        b = p[2];
    }
}

// MCU m of the pass -> MCU position inside its image and that image's pixels.
__device__ __forceinline__ long long locate_mcu(const KGeom& g, long long m, int& mx, int& my) {
    long long ml = m, offset = 0;
    if (g.batch > 1) {
        const long long img = m / g.mcusPerImage;
        ml = m - img * g.mcusPerImage;
        offset = img * g.imageStrideBytes;
    }
    my = (int)(ml / g.mcusPerRow);
    mx = (int)(ml - (long long)my * g.mcusPerRow);
    return offset;
}

// Which channel a block of an MCU belongs to, and its position inside the MCU
// (Encoder::createBlocks, Encoder.cpp:202-223; scan order of Encoder.cpp:500-514).
__device__ __forceinline__ void block_in_mcu(const KGeom& g, int k, int& c, int& bx, int& by) {
    c = k < g.off1 ? 0 : (k < g.off2 ? 1 : 2);
    const int kk = k - (c == 0 ? 0 : (c == 1 ? g.off1 : g.off2));
    by = kk / g.hF[c];
    bx = kk - by * g.hF[c];
}

// ----------------------------------------------------------------------------------------
// Generic fused transform: any sampling scheme, any image size.  One thread per 8x8
// block in scan order; samples are gathered (with edge replication, Image.cpp:63-139, and
// on-the-fly width-then-height reduction, Encoder.cpp:40-155) into LDS, then transformed
// in registers.
// ----------------------------------------------------------------------------------------
constexpr int kGenThreads = 64;

// Workgroup 0 of a transform kernel clears the histograms the following pass will count
// into (one set of uint64[4][256]), which saves that pass a launch of its own.
__device__ __forceinline__ void clear_histograms(u64* __restrict__ hist) {
    if (hist != nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) hist[i] = 0;
}

__global__ __launch_bounds__(kGenThreads) void k_transform_generic(KGeom g, const void* __restrict__ rgb,
                                                                   const float* __restrict__ qtab,
                                                                   int16_t* __restrict__ coef, u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    __shared__ float samples[kGenThreads * 65];
    __shared__ float q[3 * 64];
    for (int i = threadIdx.x; i < 192; i += kGenThreads) q[i] = qtab[i];
    __syncthreads();

    const long long b = (long long)blockIdx.x * kGenThreads + threadIdx.x;
    if (b >= g.totalBlocks) return;
    const long long m = b / g.bpm;
    const int k = (int)(b - m * g.bpm);
    int c, bx, by;
    block_in_mcu(g, k, c, bx, by);
    int mx, my;
    const unsigned char* pix = (const unsigned char*)rgb + locate_mcu(g, m, mx, my);
    const int X0 = (mx * g.hF[c] + bx) * 8, Y0 = (my * g.vF[c] + by) * 8;
    const int wf = g.wf[c], hf = g.hf[c];
    const int nw = g.wm[c] == JPEG_HIP_AVERAGE ? wf : 1;
    const int nh = g.hm[c] == JPEG_HIP_AVERAGE ? hf : 1;
    const bool convert = (g.stages & JPEG_HIP_STAGE_YCBCR) != 0;
    float* mine = samples + threadIdx.x * 65;

    for (int i = 0; i < 64; ++i) {
        const int sx0 = (X0 + (i & 7)) * wf, sy0 = (Y0 + (i >> 3)) * hf;
        float vsum = 0.0f, hval = 0.0f;
        for (int dy = 0; dy < nh; ++dy) {
            const int sy = min(sy0 + dy, g.rowsAvail - 1);
            float hsum = 0.0f;
            for (int dx = 0; dx < nw; ++dx) {
                const int sx = min(sx0 + dx, g.W - 1);
                float r, gg, bb;
                load_pixel(pix, g.strideBytes, g.sampleFormat, sx, sy, r, gg, bb);
                float v = channel_value(c, convert, r, gg, bb);
                for (int o = 0; o < g.nGains; ++o)   // Encoder::multiplyColorChannelBy, Encoder.cpp:35-38
                    if (g.gains[o].channel == c && !g.gains[o].afterSubsampling) v *= g.gains[o].value;
                hsum = dx == 0 ? v : hsum + v;
            }
            hval = nw > 1 ? hsum / (float)wf : hsum;
            vsum = dy == 0 ? hval : vsum + hval;
        }
        float reduced = nh > 1 ? vsum / (float)hf : hval;
        for (int o = 0; o < g.nGains; ++o)
            if (g.gains[o].channel == c && g.gains[o].afterSubsampling) reduced *= g.gains[o].value;
        mine[i] = reduced;
    }

    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = mine[i] * 255.0f;   // SIMD256.h:484-494
    aan2d(v);
    const float* qc = q + 64 * c;
    u32 packed[32];
#pragma unroll
    for (int z = 0; z < 64; z += 2) {
        const int lo = (int)rintf(v[kZZ[z]] / qc[kZZ[z]]);
        const int hi = (int)rintf(v[kZZ[z + 1]] / qc[kZZ[z + 1]]);
        packed[z >> 1] = ((u32)lo & 0xffffu) | ((u32)hi << 16);
    }
    uint4* dst = (uint4*)(coef + b * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = make_uint4(packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]);
}

// ----------------------------------------------------------------------------------------
// Fast fused transforms for byte pixels (4:2:0 and 4:4:4).
//
// Both divisions of the path are replaced by sequences that are proven to give the same
// correctly rounded quotient:
//   byte / 255.0f   == fma(x, hi, x * lo) with hi + lo = 1/255 to 48 bits (all 256 bytes)
//   value / q[i]    == fma(x, rhi, x * rlo) with rhi + rlo = 1/q to 48 bits, for every
//                      integer divisor 1..255 and every float x: the quotient can never
//                      sit closer than 2^-33 (relative) to a rounding boundary, the
//                      two-term product is accurate to 2^-47.
// (oracle/tools/verify_exact_division.c checks both exhaustively).  Rounding to the
// nearest-even integer is done by adding 1.5 * 2^23, which leaves the integer in the low
// mantissa bits.
// ----------------------------------------------------------------------------------------
#define INV255_HI 0x1.010102p-8f
#define INV255_LO -0x1.fdfdfep-33f
#define ROUND_MAGIC 12582912.0f

__device__ __forceinline__ float unorm8(u32 w, int k) {
    const float x = (float)((w >> (8 * k)) & 0xffu);
    return __builtin_fmaf(x, INV255_HI, x * INV255_LO);
}

// The three channels of pixel i of an 8-pixel row held in six dwords.
__device__ __forceinline__ void pixel_ycbcr(const u32 (&raw)[6], int i, float& y, float& cb, float& cr) {
    const float r = unorm8(raw[(3 * i) >> 2], (3 * i) & 3);
    const float g = unorm8(raw[(3 * i + 1) >> 2], (3 * i + 1) & 3);
    const float b = unorm8(raw[(3 * i + 2) >> 2], (3 * i + 2) & 3);
    float t = 0.2990f * r;
    t = t + 0.5870f * g;
    y = t + 0.1140f * b;
    t = -0.1687f * r;
    t = t + -0.3312f * g;
    t = t + 0.5000f * b;
    cb = t + 0.5f;
    t = 0.5000f * r;
    t = t + -0.4186f * g;
    t = t + -0.0813f * b;
    cr = t + 0.5f;
}

// Pixel i of two rows at once: (row a, row b) pairs through the same arithmetic as pixel_ycbcr.
__device__ __forceinline__ f2 unorm8_pair(u32 wa, u32 wb, int k) {
    const f2 x = {(float)((wa >> (8 * k)) & 0xffu), (float)((wb >> (8 * k)) & 0xffu)};
    return fma2(x, splat(INV255_HI), x * INV255_LO);
}

__device__ __forceinline__ void pixel_ycbcr_pair(const u32 (&ra)[6], const u32 (&rb)[6], int i, f2& y, f2& cb, f2& cr) {
    const f2 r = unorm8_pair(ra[(3 * i) >> 2], rb[(3 * i) >> 2], (3 * i) & 3);
    const f2 g = unorm8_pair(ra[(3 * i + 1) >> 2], rb[(3 * i + 1) >> 2], (3 * i + 1) & 3);
    const f2 b = unorm8_pair(ra[(3 * i + 2) >> 2], rb[(3 * i + 2) >> 2], (3 * i + 2) & 3);
    f2 t = r * 0.2990f;
    t = t + g * 0.5870f;
    y = t + b * 0.1140f;
    t = r * -0.1687f;
    t = t + g * -0.3312f;
    t = t + b * 0.5000f;
    cb = t + 0.5f;
    t = r * 0.5000f;
    t = t + g * -0.4186f;
    t = t + b * -0.0813f;
    cr = t + 0.5f;
}

// x255 (SIMD256.h:484-494), transform, quantize, zig-zag: row-pair values in, the block's
// 32 dwords of int16 pairs out.
__device__ __forceinline__ void transform_quantize_pairs(f2 (&a)[32], const float* __restrict__ rhi,
                                                         const float* __restrict__ rlo, u32 (&p)[32]) {
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = a[i] * 255.0f;
    f2 b[32];
    aan2d_pairs(a, b);
    float q[64];
#pragma unroll
    for (int i = 0; i < 32; ++i) {   // natural order, two neighbouring columns at a time
        const f2 hi = {rhi[2 * i], rhi[2 * i + 1]}, lo = {rlo[2 * i], rlo[2 * i + 1]};
        const f2 t = fma2(b[i], hi, b[i] * lo) + ROUND_MAGIC;
        q[2 * i] = t.x;
        q[2 * i + 1] = t.y;
    }
#pragma unroll
    for (int z = 0; z < 64; z += 2)
        p[z >> 1] = __builtin_amdgcn_perm(__float_as_uint(q[kZZ[z + 1]]), __float_as_uint(q[kZZ[z]]), 0x05040100u);
}

// Row of 8 pixels starting at (x0, y): one aligned 24-byte read inside the image,
// byte-wise edge replication (Image.cpp:63-139) outside.
__device__ __forceinline__ void load_row8(const unsigned char* __restrict__ rgb, long long stride, int x0, int y,
                                          bool interior, int W, int rows, u32 (&raw)[6]) {
    if (interior) {
        const uint3* p = (const uint3*)(rgb + (long long)y * stride + (long long)x0 * 3);
        const uint3 a = p[0], b = p[1];
        raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = b.x; raw[4] = b.y; raw[5] = b.z;
    } else {
        const unsigned char* row = rgb + (long long)min(y, rows - 1) * stride;
#pragma unroll
        for (int d = 0; d < 6; ++d) raw[d] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned char* px = row + (long long)min(x0 + i, W - 1) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int bidx = 3 * i + ch;
                raw[bidx >> 2] |= (u32)px[ch] << ((bidx & 3) * 8);
            }
        }
    }
}

// x255, transform, quantize with the reciprocal pairs of channel table `tab`, zig-zag,
// and store the block's 128 bytes.
__device__ __forceinline__ void finish_block(float (&v)[64], const float* __restrict__ rhi,
                                             const float* __restrict__ rlo, int16_t* __restrict__ dstBlock) {
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = v[i] * 255.0f;   // SIMD256.h:484-494
    aan2d(v);
    u32 packed[32];
#pragma unroll
    for (int z = 0; z < 64; z += 2) {
        const int a = kZZ[z], b = kZZ[z + 1];
        const float qa = __builtin_fmaf(v[a], rhi[a], v[a] * rlo[a]) + ROUND_MAGIC;
        const float qb = __builtin_fmaf(v[b], rhi[b], v[b] * rlo[b]) + ROUND_MAGIC;
        packed[z >> 1] = __builtin_amdgcn_perm(__float_as_uint(qb), __float_as_uint(qa), 0x05040100u);
    }
    uint4* dst = (uint4*)dstBlock;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = make_uint4(packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]);
}

// Same as finish_block for row-pair values (see aan2d_pairs), and the wave's 64 blocks go through LDS so that every store
// instruction writes whole 128-byte lines (8 lanes per block) instead of 64 scattered
// 16-byte pieces.  `stage` is this wave's area of kRounds-th of 64 x 144 bytes (the 16
// bytes of padding per block keep the 16-byte LDS accesses conflict free); with kRounds
// = 2 the two half waves take turns, which halves the LDS footprint.  blockStride:
// distance in blocks between the blocks of consecutive lanes in the coefficient buffer;
// lanes at or beyond `count` hold no block.
constexpr int kStagePitch = 144;

template <int kRounds>
__device__ __forceinline__ void finish_block_staged(f2 (&a)[32], const float* __restrict__ rhi,
                                                    const float* __restrict__ rlo, unsigned char* stage,
                                                    int16_t* __restrict__ waveBase, int blockStride, int count) {
    const int lane = threadIdx.x & 63;
    constexpr int kPerRound = 64 / kRounds;
    u32 p[32];
    transform_quantize_pairs(a, rhi, rlo, p);
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
        if (kRounds == 1 || lane / kPerRound == round) {
            uint4* mine = (uint4*)(stage + (lane % kPerRound) * kStagePitch);
#pragma unroll
            for (int i = 0; i < 8; ++i) mine[i] = make_uint4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
        }
        // Same wave wrote and reads: only the LDS counter has to drain, no barrier.
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8 / kRounds; ++k) {
            const int local = 8 * k + (lane >> 3);
            const int blk = round * kPerRound + local;
            if (blk < count) {
                const uint4 t = *(const uint4*)(stage + local * kStagePitch + (lane & 7) * 16);
                *(uint4*)((unsigned char*)waveBase + (long long)blk * blockStride * 128 + (lane & 7) * 16) = t;
            }
        }
        if (kRounds > 1) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Luma variant for the 4:2:0 kernel: lane l holds block (l & 3) of the wave's MCU (l >> 2);
// an MCU's four luma blocks are contiguous (512 bytes), MCUs are six blocks apart.  Two
// half-wave rounds through a 32-block stage.
__device__ __forceinline__ void finish_block_staged_y(f2 (&a)[32], const float* __restrict__ rhi,
                                                      const float* __restrict__ rlo, unsigned char* stage,
                                                      int16_t* __restrict__ waveBase, int mcuCount) {
    const int lane = threadIdx.x & 63;
    u32 p[32];
    transform_quantize_pairs(a, rhi, rlo, p);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if ((lane >> 5) == round) {
            uint4* mine = (uint4*)(stage + (lane & 31) * kStagePitch);
#pragma unroll
            for (int i = 0; i < 8; ++i) mine[i] = make_uint4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int local = 8 * k + (lane >> 3);          // block within the round, 0..31
            const int blk = round * 32 + local;             // the lane that produced it
            const int mcu = blk >> 2;
            if (mcu < mcuCount) {
                const uint4 t = *(const uint4*)(stage + local * kStagePitch + (lane & 7) * 16);
                *(uint4*)((unsigned char*)waveBase + ((long long)mcu * 6 + (blk & 3)) * 128 + (lane & 7) * 16) = t;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// 4:2:0 (SamplingScheme.h:63-66): four threads per 16x16 MCU, one per luma block.  Each
// thread converts its 64 pixels once, keeps Y in registers and reduces Cb/Cr 2x2 on the
// fly -- horizontal pairs first, then the two rows, as the in-place passes of
// Encoder.cpp:58-64 and 118-126 do -- into a 4x4 corner of the MCU's chroma blocks in
// LDS.  After the barrier the first two waves transform the 128 chroma blocks.
constexpr int k420Mcus = 64;
constexpr int kChromaPitch = 68;   // floats per block in LDS: keeps 16-byte rows, spreads banks

__global__ __launch_bounds__(256) void k_transform_420(KGeom g, const unsigned char* __restrict__ rgb,
                                                       const float* __restrict__ qtab, int16_t* __restrict__ coef,
                                                       u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    __shared__ __attribute__((aligned(16))) float chroma[2 * k420Mcus * kChromaPitch];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 32 * kStagePitch];
    const float* rhi = qtab + 192;
    const float* rlo = qtab + 384;
    const int tid = threadIdx.x, q = tid & 3, ml = tid >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long mBase = (long long)blockIdx.x * k420Mcus;
    const long long left = g.nMcu - mBase;
    const int mcuCount = left < k420Mcus ? (int)left : k420Mcus;   // MCUs of this workgroup that exist
    const long long m = mBase + (ml < mcuCount ? ml : mcuCount - 1);   // idle lanes redo the last MCU, store nothing
    {
        int mx, my;
        rgb += locate_mcu(g, m, mx, my);
        const int x0 = mx * 16 + (q & 1) * 8, y0 = my * 16 + (q >> 1) * 8;
        const bool interior = x0 + 8 <= g.W && y0 + 8 <= g.rowsAvail;
        f2 a[32];   // a[rp * 8 + i] = Y of pixel i in rows 2rp and 2rp + 1
        // Chroma blocks in LDS hold their rows in pairs as well: value (r, c) at
        // (r / 2) * 16 + c * 2 + (r & 1), so the chroma waves read row-pair values directly.
        float* cbDst = chroma + ml * kChromaPitch + (q >> 1) * 32 + (q & 1) * 8;
        float* crDst = cbDst + k420Mcus * kChromaPitch;
        // All sixteen row loads are issued before the first conversion, so one memory
        // latency is paid per thread instead of one per row.
        u32 raw[8][6];
#pragma unroll
        for (int r = 0; r < 8; ++r) load_row8(rgb, g.strideBytes, x0, y0 + r, interior, g.W, g.rowsAvail, raw[r]);
        __builtin_amdgcn_sched_barrier(0);
        float ecb[4], ecr[4];   // chroma row of the even row pair, waiting for its odd partner
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            f2 cb[8], cr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pixel_ycbcr_pair(raw[2 * rp], raw[2 * rp + 1], i, a[8 * rp + i], cb[i], cr[i]);
            float ocb[4], ocr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (a + b) / 2.0f across, both rows at once (SIMD256.h:260-277), then
                // (row0 + row1) / 2.0f down (SIMD256.h:295-304).
                const f2 hcb = (cb[2 * j] + cb[2 * j + 1]) * 0.5f;
                const f2 hcr = (cr[2 * j] + cr[2 * j + 1]) * 0.5f;
                ocb[j] = (hcb.x + hcb.y) * 0.5f;
                ocr[j] = (hcr.x + hcr.y) * 0.5f;
            }
            if ((rp & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ecb[j] = ocb[j];
                    ecr[j] = ocr[j];
                }
            } else {   // chroma rows (q >> 1) * 4 + rp - 1 and + rp form a pair
                float* db = cbDst + (rp >> 1) * 16;
                float* dr = crDst + (rp >> 1) * 16;
                *(float4*)db = make_float4(ecb[0], ocb[0], ecb[1], ocb[1]);
                *(float4*)(db + 4) = make_float4(ecb[2], ocb[2], ecb[3], ocb[3]);
                *(float4*)dr = make_float4(ecr[0], ocr[0], ecr[1], ocr[1]);
                *(float4*)(dr + 4) = make_float4(ecr[2], ocr[2], ecr[3], ocr[3]);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the row pairs in order: bounds live registers
        }
        // The wave's 64 luma blocks are 16 MCUs x 4: consecutive lanes, consecutive blocks
        // inside an MCU, 6 blocks from MCU to MCU.  Lane l's block sits at
        // (l >> 2) * 6 + (l & 3) blocks from the wave's first; staged stores take a uniform
        // stride, so the four blocks of an MCU are addressed through their MCU.
        const int waveMcus = mcuCount - wave * 16;   // MCUs of this wave that exist (may be <= 0)
        finish_block_staged_y(a, rhi, rlo, stage + wave * 32 * kStagePitch, coef + (mBase + wave * 16) * 6 * 64,
                              waveMcus);
    }
    __syncthreads();
    if (tid < 2 * k420Mcus) {
        const int ch = wave;   // wave 0: Cb, wave 1: Cr
        const float4* src = (const float4*)(chroma + (ch * k420Mcus + (tid & 63)) * kChromaPitch);
        f2 a[32];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 t = src[i];
            a[2 * i] = (f2){t.x, t.y};
            a[2 * i + 1] = (f2){t.z, t.w};
        }
        finish_block_staged<2>(a, rhi + 64 * (1 + ch), rlo + 64 * (1 + ch), stage + wave * 32 * kStagePitch,
                               coef + (mBase * 6 + 4 + ch) * 64, 6, mcuCount);
    }
}

// 4:4:4 (SamplingScheme.h:48-51): three waves per workgroup, one per channel, each thread
// one 8x8 block.  The three waves of a workgroup read the same 64 x 192 pixel bytes (two of
// them from L1/L2), convert them for their own channel only and transform from registers:
// no value is shared between channels, so nothing has to be kept or exchanged, the channel
// is wave-uniform (scalar table loads, no divergence) and a 4K image still fills the chip.
constexpr int k444Positions = 64;

__global__ __launch_bounds__(3 * k444Positions) void k_transform_444(KGeom g, const unsigned char* __restrict__ rgb,
                                                                     const float* __restrict__ qtab,
                                                                     int16_t* __restrict__ coef, u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* rhi = qtab + 192 + 64 * c;
    const float* rlo = qtab + 384 + 64 * c;
    // Matrix row of this wave's channel (SIMD256.h:170-175) in scalar registers.  Y's bias
    // of +0.0f is added like the others: Y is never -0.0, so the sum is unchanged.
    const float m0 = c == 0 ? 0.2990f : (c == 1 ? -0.1687f : 0.5000f);
    const float m1 = c == 0 ? 0.5870f : (c == 1 ? -0.3312f : -0.4186f);
    const float m2 = c == 0 ? 0.1140f : (c == 1 ? 0.5000f : -0.0813f);
    const float bias = c == 0 ? 0.0f : 0.5f;
    __shared__ __attribute__((aligned(16))) unsigned char stage[3 * 32 * kStagePitch];
    const long long m0w = (long long)blockIdx.x * k444Positions;
    const long long m = m0w + (threadIdx.x & 63);
    const long long left = g.nMcu - m0w;
    const int count = left < 64 ? (int)left : 64;
    const long long mc = m < g.nMcu ? m : g.nMcu - 1;   // idle lanes redo the last block, store nothing
    int mx, my;
    rgb += locate_mcu(g, mc, mx, my);
    const int x0 = mx * 8, y0 = my * 8;
    const bool interior = x0 + 8 <= g.W && y0 + 8 <= g.rowsAvail;
    u32 raw[8][6];
#pragma unroll
    for (int r = 0; r < 8; ++r) load_row8(rgb, g.strideBytes, x0, y0 + r, interior, g.W, g.rowsAvail, raw[r]);
    __builtin_amdgcn_sched_barrier(0);
    f2 a[32];   // a[rp * 8 + i]: pixel i of rows 2rp and 2rp + 1
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f2 rr = unorm8_pair(raw[2 * rp][(3 * i) >> 2], raw[2 * rp + 1][(3 * i) >> 2], (3 * i) & 3);
            const f2 gg = unorm8_pair(raw[2 * rp][(3 * i + 1) >> 2], raw[2 * rp + 1][(3 * i + 1) >> 2], (3 * i + 1) & 3);
            const f2 bb = unorm8_pair(raw[2 * rp][(3 * i + 2) >> 2], raw[2 * rp + 1][(3 * i + 2) >> 2], (3 * i + 2) & 3);
            f2 t = rr * m0;
            t = t + gg * m1;
            t = t + bb * m2;
            a[8 * rp + i] = t + bias;
        }
        __builtin_amdgcn_sched_barrier(0);   // row pair by row pair: bounds the live registers
    }
    finish_block_staged<2>(a, rhi, rlo, stage + c * 32 * kStagePitch, coef + (m0w * 3 + c) * 64, 3, count);
}

// Pixel filters in front of the transform.  One thread per pixel; in place when src == dst.
// filter < 0 only converts to float.
__global__ __launch_bounds__(256) void k_pixel_filter(const void* __restrict__ src, int fmt, long long srcStride,
                                                      float* __restrict__ dst, int width, int filter, int channel,
                                                      float value) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width) return;
    float r, g, b;
    load_pixel(src, srcStride, fmt, x, y, r, g, b);
    float o[3] = {r, g, b};
    if (filter == JPEG_HIP_FILTER_SEPIA) {   // applySepiaFilterAVXImpl, SIMD256.h:218-244; its "+ 0 * 1" adds +0.0f
        float t = 0.3930f * r;
        t = t + 0.7690f * g;
        t = t + 0.1890f * b;
        o[0] = t + 0.0f;
        t = 0.3490f * r;
        t = t + 0.6860f * g;
        t = t + 0.1680f * b;
        o[1] = t + 0.0f;
        t = 0.2720f * r;
        t = t + 0.5340f * g;
        t = t + 0.1310f * b;
        o[2] = t + 0.0f;
    } else if (filter == JPEG_HIP_FILTER_MULTIPLY) {   // multiplyAVX, SIMD256.h:246-258
        o[channel] = o[channel] * value;
    }
    float* p = dst + ((long long)y * width + x) * 3;
    p[0] = o[0];
    p[1] = o[1];
    p[2] = o[2];
}

// Rows of bytes that start anywhere -> rows that start on 16-byte boundaries.  A thread makes
// 16 bytes of a row from the five aligned dwords that hold them (v_alignbyte); dwords past
// the last source byte are not touched.  Bytes of a destination row behind rowBytes are padding.
// Row y belongs to image y / rowsPerImage (images srcImageStride apart, packed in the copy).
__global__ __launch_bounds__(256) void k_realign_rows(const unsigned char* __restrict__ src, long long srcStride,
                                                      long long srcImageStride, int rowsPerImage,
                                                      unsigned char* __restrict__ dst, long long dstStride, int rowBytes,
                                                      int rows) {
    const int unit = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (16 * unit >= rowBytes) return;
    const int img = y / rowsPerImage, last = (rows - 1) / rowsPerImage;
    const unsigned char* s = src + img * srcImageStride + (long long)(y - img * rowsPerImage) * srcStride + 16 * (long long)unit;
    const unsigned char* end = src + last * srcImageStride + (long long)(rows - 1 - last * rowsPerImage) * srcStride + rowBytes;
    const int shift = (int)((uintptr_t)s & 3);
    const u32* a = (const u32*)(s - shift);
    u32 w[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = (const unsigned char*)(a + i) < end ? a[i] : 0u;
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(w[1], w[0], shift);
    o.y = __builtin_amdgcn_alignbyte(w[2], w[1], shift);
    o.z = __builtin_amdgcn_alignbyte(w[3], w[2], shift);
    o.w = __builtin_amdgcn_alignbyte(w[4], w[3], shift);
    *(uint4*)(dst + (long long)y * dstStride + 16 * (long long)unit) = o;
}

hipError_t launch_realign_rows(const void* d_src, long long srcStride, long long srcImageStride, int rowsPerImage,
                               void* d_dst, long long dstStride, int rowBytes, int rows, hipStream_t s) {
    const int units = (rowBytes + 15) / 16;
    hipLaunchKernelGGL(k_realign_rows, dim3((unsigned)((units + 255) / 256), (unsigned)rows), dim3(256), 0, s,
                       (const unsigned char*)d_src, srcStride, srcImageStride, rowsPerImage, (unsigned char*)d_dst,
                       dstStride, rowBytes, rows);
    return hipGetLastError();
}

// Three float planes (the reference's ImageData, Image.h:30-61) -> interleaved float pixels.
// One thread per pixel; the planes lie planeFloats apart, rows packed.
__global__ __launch_bounds__(256) void k_interleave_planes(const float* __restrict__ planes, long long planeFloats,
                                                           float* __restrict__ dst, long long pixels) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels) return;
    float* p = dst + i * 3;
    p[0] = planes[i];
    p[1] = planes[planeFloats + i];
    p[2] = planes[2 * planeFloats + i];
}

hipError_t launch_interleave_planes(const float* d_planes, long long planeFloats, float* d_dst, long long pixels,
                                    hipStream_t s) {
    hipLaunchKernelGGL(k_interleave_planes, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, s, d_planes, planeFloats,
                       d_dst, pixels);
    return hipGetLastError();
}

hipError_t launch_pixel_filter(const void* d_src, int sampleFormat, long long srcStride, float* d_dst, int width, int rows,
                               int filter, int channel, float value, hipStream_t s) {
    hipLaunchKernelGGL(k_pixel_filter, dim3((unsigned)((width + 255) / 256), (unsigned)rows), dim3(256), 0, s, d_src,
                       sampleFormat, srcStride, d_dst, width, filter, channel, value);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------
// Row-wise transform for the remaining schemes (4:2:2, 4:1:1, the averaging and the
// height-reducing variants) on byte pixels: one thread per 8x8 block like the generic
// kernel, but the source pixels come in as 24-byte rows (load_row8, which also does the
// edge replication), samples go straight to registers, and a wave's 64 blocks leave
// through the store stage in whole lines.  The reduction follows Encoder.cpp:40-155: across
// first, then down; "Average" sums left to right (top to bottom) and divides by the factor,
// "Subsampling" keeps the first pixel of each group.
// ----------------------------------------------------------------------------------------
template <int WF, int NH>
__device__ __forceinline__ void gather_block_rows(const KGeom& g, const unsigned char* __restrict__ pix, int sx0, int sy0,
                                                  int hf, int nw, bool interior, int c, f2 (&a)[32]) {
    // For groups of up to four source pixels: as many sample rows at a time as 48 registers of
    // source pixels hold, their loads all issued before the first is used.
    constexpr int kRows = 8 / (WF * NH) > 0 ? 8 / (WF * NH) : 1;
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += kRows) {
        u32 raw[kRows][NH][WF][6];
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr)
#pragma unroll
            for (int dy = 0; dy < NH; ++dy)
#pragma unroll
                for (int sgm = 0; sgm < WF; ++sgm)
                    load_row8(pix, g.strideBytes, sx0 + 8 * sgm, sy0 + (r0 + rr) * hf + dy, interior, g.W, g.rowsAvail,
                              raw[rr][dy][sgm]);
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) {
            const int r = r0 + rr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float vsum = 0.0f;
#pragma unroll
                for (int dy = 0; dy < NH; ++dy) {
                    float hsum = 0.0f;
#pragma unroll
                    for (int dx = 0; dx < WF; ++dx) {
                        if (dx < nw) {
                            const int px = i * WF + dx, sgm = px >> 3, j = px & 7;
                            const float rd = unorm8(raw[rr][dy][sgm][(3 * j) >> 2], (3 * j) & 3);
                            const float gn = unorm8(raw[rr][dy][sgm][(3 * j + 1) >> 2], (3 * j + 1) & 3);
                            const float bl = unorm8(raw[rr][dy][sgm][(3 * j + 2) >> 2], (3 * j + 2) & 3);
                            const float v = channel_value(c, true, rd, gn, bl);
                            hsum = dx == 0 ? v : hsum + v;
                        }
                    }
                    const float hval = nw > 1 ? hsum / (float)WF : hsum;
                    vsum = dy == 0 ? hval : vsum + hval;
                }
                const float val = NH > 1 ? vsum / (float)NH : vsum;
                if (r & 1) a[(r >> 1) * 8 + i].y = val;
                else a[(r >> 1) * 8 + i].x = val;
            }
        }
    }
}

// The deep groups (eight or more source pixels per sample): one sample row at a time, the rows
// of a group in a loop that is not unrolled, which keeps the code within the instruction cache.
template <int WF>
__device__ __forceinline__ void gather_block_rows_deep(const KGeom& g, const unsigned char* __restrict__ pix, int sx0,
                                                       int sy0, int hf, int nw, int nh, bool interior, int c, f2 (&a)[32]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float vsum[8];
#pragma unroll 1
        for (int dy = 0; dy < nh; ++dy) {
            u32 raw[WF][6];
#pragma unroll
            for (int sgm = 0; sgm < WF; ++sgm)
                load_row8(pix, g.strideBytes, sx0 + 8 * sgm, sy0 + r * hf + dy, interior, g.W, g.rowsAvail, raw[sgm]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float hsum = 0.0f;
#pragma unroll
                for (int dx = 0; dx < WF; ++dx) {
                    if (dx < nw) {
                        const int px = i * WF + dx, sgm = px >> 3, j = px & 7;
                        const float rd = unorm8(raw[sgm][(3 * j) >> 2], (3 * j) & 3);
                        const float gn = unorm8(raw[sgm][(3 * j + 1) >> 2], (3 * j + 1) & 3);
                        const float bl = unorm8(raw[sgm][(3 * j + 2) >> 2], (3 * j + 2) & 3);
                        const float v = channel_value(c, true, rd, gn, bl);
                        hsum = dx == 0 ? v : hsum + v;
                    }
                }
                const float hval = nw > 1 ? hsum / (float)WF : hsum;
                vsum[i] = dy == 0 ? hval : vsum[i] + hval;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float val = nh > 1 ? vsum[i] / (float)nh : vsum[i];
            if (r & 1) a[(r >> 1) * 8 + i].y = val;
            else a[(r >> 1) * 8 + i].x = val;
        }
    }
}

template <int WF>
__device__ __forceinline__ void gather_block_rows(const KGeom& g, const unsigned char* __restrict__ pix, int sx0, int sy0,
                                                  int hf, int nw, int nh, bool interior, int c, f2 (&a)[32]) {
    if (nh == 1) gather_block_rows<WF, 1>(g, pix, sx0, sy0, hf, nw, interior, c, a);
    else if (WF * 2 <= 4 && nh == 2) gather_block_rows<WF, (WF * 2 <= 4 ? 2 : 1)>(g, pix, sx0, sy0, hf, nw, interior, c, a);
    else if (WF == 1 && nh == 4) gather_block_rows<WF, (WF == 1 ? 4 : 1)>(g, pix, sx0, sy0, hf, nw, interior, c, a);
    else gather_block_rows_deep<WF>(g, pix, sx0, sy0, hf, nw, nh, interior, c, a);
}

// Float pixels (Image::setRawPixelData, values already in 0..1): eight pixels are 96 bytes,
// six 16-byte reads when rows start on 16-byte boundaries, single floats otherwise and at
// the image edge.
__device__ __forceinline__ void load_row8f(const unsigned char* __restrict__ rgb, long long stride, int x0, int y,
                                           bool fast, int W, int rows, float (&raw)[24]) {
    if (fast) {
        const float4* p = (const float4*)(rgb + (long long)y * stride + (long long)x0 * 12);
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float4 v = p[d];
            raw[4 * d] = v.x; raw[4 * d + 1] = v.y; raw[4 * d + 2] = v.z; raw[4 * d + 3] = v.w;
        }
    } else {
        const unsigned char* row = rgb + (long long)min(y, rows - 1) * stride;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* px = (const float*)(row + (long long)min(x0 + i, W - 1) * 12);
            raw[3 * i] = px[0]; raw[3 * i + 1] = px[1]; raw[3 * i + 2] = px[2];
        }
    }
}

template <int WF>
__device__ __forceinline__ void gather_block_rows_f32(const KGeom& g, const unsigned char* __restrict__ pix, int sx0,
                                                      int sy0, int hf, int nw, int nh, bool fast, int c, bool convert,
                                                      f2 (&a)[32]) {
    constexpr int kPer = 8 / WF;   // samples that one 8-pixel segment yields
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float vsum[8];
#pragma unroll 1
        for (int dy = 0; dy < nh; ++dy) {
#pragma unroll
            for (int sgm = 0; sgm < WF; ++sgm) {
                float raw[24];
                load_row8f(pix, g.strideBytes, sx0 + 8 * sgm, sy0 + r * hf + dy, fast, g.W, g.rowsAvail, raw);
#pragma unroll
                for (int ii = 0; ii < kPer; ++ii) {
                    const int i = sgm * kPer + ii;
                    float hsum = 0.0f;
#pragma unroll
                    for (int dx = 0; dx < WF; ++dx) {
                        if (dx < nw) {
                            const int j = ii * WF + dx;
                            const float v = channel_value(c, convert, raw[3 * j], raw[3 * j + 1], raw[3 * j + 2]);
                            hsum = dx == 0 ? v : hsum + v;
                        }
                    }
                    const float hval = nw > 1 ? hsum / (float)WF : hsum;
                    vsum[i] = dy == 0 ? hval : vsum[i] + hval;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float val = nh > 1 ? vsum[i] / (float)nh : vsum[i];
            if (r & 1) a[(r >> 1) * 8 + i].y = val;
            else a[(r >> 1) * 8 + i].x = val;
        }
    }
}

// A workgroup takes 64 MCUs, wave j block j of each of them: channel and factors are the
// same for a whole wave (no divergence between luma and chroma lanes), and lane l's block
// lies bpm blocks behind lane l - 1's, which is what the store stage expects.
constexpr int kRowsMaxBpm = 8;

template <bool kFloat>
__global__ __launch_bounds__(64 * kRowsMaxBpm) void k_transform_rows(KGeom g, const unsigned char* __restrict__ rgb,
                                                                     const float* __restrict__ qtab,
                                                                     int16_t* __restrict__ coef, u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    __shared__ __attribute__((aligned(16))) unsigned char stage[kRowsMaxBpm * 32 * kStagePitch];
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // block inside the MCU
    const int lane = threadIdx.x & 63;
    const long long mBase = (long long)blockIdx.x * 64;
    const long long left = g.nMcu - mBase;
    const int count = left < 64 ? (int)left : 64;
    const long long m = mBase + (lane < count ? lane : count - 1);   // idle lanes redo the last MCU, store nothing
    int c, bx, by;
    block_in_mcu(g, k, c, bx, by);
    int mx, my;
    const unsigned char* pix = rgb + locate_mcu(g, m, mx, my);
    const int wf = g.wf[c], hf = g.hf[c];
    const int nw = g.wm[c] == JPEG_HIP_AVERAGE ? wf : 1;
    const int nh = g.hm[c] == JPEG_HIP_AVERAGE ? hf : 1;
    const int sx0 = (mx * g.hF[c] + bx) * 8 * wf, sy0 = (my * g.vF[c] + by) * 8 * hf;
    // With "Subsampling" only the first row and column of each group are read, but a block
    // that reaches past the image in the unread ones is treated as an edge block all the same.
    const bool interior = sx0 + 8 * wf <= g.W && sy0 + 8 * hf <= g.rowsAvail;
    f2 a[32];
    if (kFloat) {
        const bool convert = (g.stages & JPEG_HIP_STAGE_YCBCR) != 0;
        const bool fast = interior && (((uintptr_t)rgb | (uintptr_t)g.strideBytes | (uintptr_t)g.imageStrideBytes) & 15) == 0;
        if (wf == 1) gather_block_rows_f32<1>(g, pix, sx0, sy0, hf, nw, nh, fast, c, convert, a);
        else if (wf == 2) gather_block_rows_f32<2>(g, pix, sx0, sy0, hf, nw, nh, fast, c, convert, a);
        else gather_block_rows_f32<4>(g, pix, sx0, sy0, hf, nw, nh, fast, c, convert, a);
    } else {
        if (wf == 1) gather_block_rows<1>(g, pix, sx0, sy0, hf, nw, nh, interior, c, a);
        else if (wf == 2) gather_block_rows<2>(g, pix, sx0, sy0, hf, nw, nh, interior, c, a);
        else gather_block_rows<4>(g, pix, sx0, sy0, hf, nw, nh, interior, c, a);
    }
    finish_block_staged<2>(a, qtab + 192 + 64 * c, qtab + 384 + 64 * c, stage + k * 32 * kStagePitch,
                           coef + (mBase * g.bpm + k) * 64, g.bpm, count);
}

// The row-wise kernel takes factors 1, 2 and 4 and up to eight blocks per MCU (every scheme
// of the reference).
static bool rows_kernel_takes(const KGeom& g) {
    for (int c = 0; c < 3; ++c)
        if (g.wf[c] == 3 || g.hf[c] == 3) return false;
    return g.bpm <= kRowsMaxBpm;
}

static bool scheme_is(const KGeom& g, int wf, int hf, int method) {
    if (g.wf[0] != 1 || g.hf[0] != 1) return false;
    for (int c = 1; c < 3; ++c) {
        if (g.wf[c] != wf || g.hf[c] != hf) return false;
        if (wf > 1 && g.wm[c] != method) return false;
        if (hf > 1 && g.hm[c] != method) return false;
    }
    return true;
}

hipError_t launch_transform(const KGeom& g, const void* d_rgb, const float* d_qtab, int16_t* d_coef, hipStream_t s,
                            unsigned long long* d_clearHist) {
    const bool bytesAligned = g.sampleFormat == JPEG_HIP_RGB8 && g.stages == JPEG_HIP_STAGE_ALL && g.nGains == 0 &&
                              (g.strideBytes & 3) == 0 && ((uintptr_t)d_rgb & 3) == 0 &&
                              (g.batch == 1 || (g.imageStrideBytes & 3) == 0);
    if (bytesAligned && scheme_is(g, 2, 2, JPEG_HIP_AVERAGE)) {
        const long long grid = (g.nMcu + k420Mcus - 1) / k420Mcus;
        hipLaunchKernelGGL(k_transform_420, dim3((unsigned)grid), dim3(256), 0, s, g, (const unsigned char*)d_rgb, d_qtab,
                           d_coef, d_clearHist);
    } else if (bytesAligned && scheme_is(g, 1, 1, JPEG_HIP_SUBSAMPLING)) {
        const long long grid = (g.nMcu + k444Positions - 1) / k444Positions;
        hipLaunchKernelGGL(k_transform_444, dim3((unsigned)grid), dim3(3 * k444Positions), 0, s, g,
                           (const unsigned char*)d_rgb, d_qtab, d_coef, d_clearHist);
    } else if (bytesAligned && rows_kernel_takes(g)) {
        const long long grid = (g.nMcu + 63) / 64;
        hipLaunchKernelGGL(k_transform_rows<false>, dim3((unsigned)grid), dim3(64 * g.bpm), 0, s, g,
                           (const unsigned char*)d_rgb, d_qtab, d_coef, d_clearHist);
    } else if (g.sampleFormat == JPEG_HIP_RGBF32 && g.nGains == 0 && rows_kernel_takes(g)) {   // any stages, any scheme
        const long long grid = (g.nMcu + 63) / 64;
        hipLaunchKernelGGL(k_transform_rows<true>, dim3((unsigned)grid), dim3(64 * g.bpm), 0, s, g,
                           (const unsigned char*)d_rgb, d_qtab, d_coef, d_clearHist);
    } else {
        const long long grid = (g.totalBlocks + kGenThreads - 1) / kGenThreads;
        hipLaunchKernelGGL(k_transform_generic, dim3((unsigned)grid), dim3(kGenThreads), 0, s, g, d_rgb, d_qtab, d_coef,
                           d_clearHist);
    }
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------
// Entropy stage, shared block walk.  A visitor sees the DC symbol, every AC symbol with
// its amplitude bits, zero-run-length symbols and the end-of-block symbol, in scan order
// (Encoder::calculateDCValues / calculateACValues, Encoder.cpp:260-352).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int bit_category(int v) {   // Encoder::lookupBitCategory, Encoder.cpp:354-366
    const int a = v < 0 ? -v : v;
    return 32 - __clz(a);   // __clz(0) == 32
}

// Number of leading bits equal to the sign bit (v_ffbh_i32; the compiler has no builtin for
// it and expands __builtin_clrsb into several instructions).  Not for 0 and -1.
__device__ __forceinline__ int leading_sign_bits(int x) {
    int r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ int coef_at(const u32 (&w)[32], int i) {
    return (int)(short)(w[i >> 1] >> ((i & 1) * 16));
}

// Where a block sits in its image.  The block index inside one image fits 32 bits for every
// legal JPEG (65535 x 65535 pixels are 2 * 10^8 blocks), which keeps the divisions cheap.
struct BlockRole {
    u32 mcu;       // MCU inside the image
    int k;         // block inside the MCU
    int c;         // channel
    int back;      // blocks between this one and the previous block of its channel, 1..bpm
    bool resets;   // first block of its channel in a restart interval: the DC predictor is 0
};

__device__ __forceinline__ BlockRole block_role(const KGeom& g, u32 bl) {
    BlockRole r;
    const u32 bpm = (u32)g.bpm;
    r.mcu = bl / bpm;
    r.k = (int)(bl - r.mcu * bpm);
    r.c = r.k < g.off1 ? 0 : (r.k < g.off2 ? 1 : 2);
    const int first = r.c == 0 ? 0 : (r.c == 1 ? g.off1 : g.off2);
    const int last = (r.c == 0 ? g.off1 : (r.c == 1 ? g.off2 : g.bpm)) - 1;
    r.back = r.k > first ? 1 : g.bpm - (last - first);
    // One interval per image (no restarts, or a batch): only MCU 0 starts one.
    const bool atStart = g.intervals == g.batch ? r.mcu == 0 : r.mcu % (u32)g.mcusPerInterval == 0;
    r.resets = r.k == first && atStart;
    return r;
}

// Block whose DC term the lane has to fetch from memory as predictor of block b: lanes hold
// consecutive blocks (lane l: block b0 + l), the predecessor is at most one MCU back, so all
// but the first `back` lanes of a wave get it from a neighbouring lane.  Lanes that need no
// fetch name their own block, which keeps the address valid and the line in cache.
__device__ __forceinline__ long long dc_predictor_source(const BlockRole& r, long long b) {
    const int lane = threadIdx.x & 63;
    return (lane >= r.back || r.resets) ? b : b - r.back;
}

// DC predictor (previous block of the same channel in scan order, 0 at the start of a
// restart interval and of the band) from the lane's own DC term, its neighbours' and the
// term fetched from dc_predictor_source().  Call with all lanes of the wave.
__device__ __forceinline__ int dc_predictor_lanes(const BlockRole& r, int myDc, int fetchedDc, bool active) {
    const int lane = threadIdx.x & 63;
    const int fromLane = __shfl(myDc, (lane - r.back) & 63, 64);
    if (!active || r.resets) return 0;
    return lane >= r.back ? fromLane : fetchedDc;
}

template <class Visitor>
__device__ __forceinline__ void walk_block(const u32 (&w)[32], int pred, Visitor& vis) {
    const int diff = coef_at(w, 0) - pred;
    const int dcat = bit_category(diff);
    vis.dc(dcat, (u32)(diff < 0 ? diff - 1 : diff) & ((1u << dcat) - 1u));
    int last = 0;   // index of the last non-zero coefficient seen (the DC term counts)
    // Eight coefficients (one 16-byte load) at a time: groups that are all zero -- most of
    // the high-frequency half of a typical block -- are skipped.  A zero coefficient costs
    // nothing but the test: the run in front of a non-zero one is the distance to `last`.
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const u32 any = (grp == 0 ? (w[0] & 0xffff0000u) : w[4 * grp]) | w[4 * grp + 1] | w[4 * grp + 2] | w[4 * grp + 3];
        if (any == 0) continue;
#pragma unroll
        for (int i = 8 * grp + (grp == 0 ? 1 : 0); i < 8 * grp + 8; ++i) {
            const int v = coef_at(w, i);
            if (v != 0) {
                int run = i - 1 - last;
                last = i;
                if (run > 15) {
                    vis.zrl(run >> 4);
                    run &= 15;
                }
                // The amplitude bits are those of v, or of v - 1 (the one's complement of |v|)
                // for a negative v; either way the category is the number of bits left
                // after the redundant sign bits (never 0 or -1 here: v is not 0).
                const int amp = v + (v >> 31);
                const int cat = 32 - leading_sign_bits(amp);
                vis.ac((run << 4) | cat, cat, (u32)amp & ((1u << cat) - 1u));
            }
        }
    }
    if (last != 63) vis.eob();
}

__device__ __forceinline__ void load_block(const int16_t* coef, long long b, u32 (&w)[32]) {
    const uint4* src = (const uint4*)(coef + b * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 t = src[i];
        w[4 * i] = t.x;
        w[4 * i + 1] = t.y;
        w[4 * i + 2] = t.z;
        w[4 * i + 3] = t.w;
    }
}

constexpr int kEntThreads = 256;

// ---- histograms -------------------------------------------------------------------------
struct HistVisitor {
    u32* dcHist;
    u32* acHist;
    __device__ __forceinline__ void dc(int cat, u32) { atomicAdd(&dcHist[cat], 1u); }
    __device__ __forceinline__ void ac(int sym, int, u32) { atomicAdd(&acHist[sym], 1u); }
    __device__ __forceinline__ void zrl(int n) { atomicAdd(&acHist[0xF0], (u32)n); }
    __device__ __forceinline__ void eob() { atomicAdd(&acHist[0], 1u); }
};

// blockIdx.y is the image of the batch; a workgroup never mixes images, so its LDS counts
// belong to one histogram set.
__global__ __launch_bounds__(kEntThreads) void k_histogram(KGeom g, const int16_t* __restrict__ coef,
                                                           u64* __restrict__ hist, HistogramExport ex) {
    __shared__ u32 lh[4 * 256];   // DC-Y, DC-C, AC-Y, AC-C (Encoder.cpp:368-416)
    __shared__ bool lastOne;
    for (int i = threadIdx.x; i < 1024; i += kEntThreads) lh[i] = 0;
    __syncthreads();
    const int16_t* image = coef + (long long)blockIdx.y * g.blocksPerImage * 64;
    const u32 blocks = (u32)g.blocksPerImage;
    const u32 stride = gridDim.x * kEntThreads;
    for (u32 bl = blockIdx.x * kEntThreads + threadIdx.x; __any(bl < blocks); bl += stride) {
        const bool active = bl < blocks;
        const u32 cur = active ? bl : blocks - 1;   // lanes past the end stay in step for the shuffle
        const BlockRole role = block_role(g, cur);
        u32 w[32];
        load_block(image, cur, w);
        const int fetched = image[dc_predictor_source(role, cur) * 64];
        const int pred = dc_predictor_lanes(role, coef_at(w, 0), fetched, active);
        if (active) {
            const int t = role.c == 0 ? 0 : 1;
            HistVisitor vis{lh + 256 * t, lh + 512 + 256 * t};
            walk_block(w, pred, vis);
        }
    }
    __syncthreads();
    u64* mine = hist + (long long)blockIdx.y * 1024;
    if (ex.host == nullptr) {
        for (int i = threadIdx.x; i < 1024; i += kEntThreads)
            if (lh[i] != 0) atomicAdd(&mine[i], (u64)lh[i]);
        return;
    }
    // The workgroup that finishes last hands the complete histograms to the host.  No
    // fences (a release at device scope writes the whole L2 back, once per workgroup): the
    // additions return their old values, so they have been performed when the workgroup
    // counts itself off, and the last one reads the sums where the additions were made.
    // The counter is never reset: the launcher tells which value the last one will see.
    u64 old[1024 / kEntThreads];
#pragma unroll
    for (int k = 0; k < 1024 / kEntThreads; ++k) {   // all in flight before the first is waited for
        const u32 v = lh[k * kEntThreads + threadIdx.x];
        old[k] = v != 0 ? atomicAdd(&mine[k * kEntThreads + threadIdx.x], (u64)v) : 0;
    }
    u64 seen = 0;
#pragma unroll
    for (int k = 0; k < 1024 / kEntThreads; ++k) seen |= old[k];
    // `seen` is there to be waited for, in front of the barrier: a store that depends on it
    // (under a condition that never holds) cannot move behind.
    if (seen == ~0ull) lh[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) lastOne = atomicAdd(ex.counter, 1ull) + 1 == ex.expected;
    __syncthreads();
    if (!lastOne) return;
    for (int i = threadIdx.x; i < 1024; i += kEntThreads)
        ex.host[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Zeroes n 8-byte words.  A kernel rather than hipMemsetAsync: a launch costs the host a
// few microseconds, a memset call several times that.
__global__ __launch_bounds__(256) void k_zero_words(u64* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0;
}

hipError_t launch_clear_histograms(unsigned long long* d_hist, int images, hipStream_t s) {
    const long long words = 1024LL * images;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((words + 255) / 256 > 1024 ? 1024 : (words + 255) / 256)), dim3(256), 0,
                       s, d_hist, words);
    return hipGetLastError();
}

hipError_t launch_histogram(const KGeom& g, const int16_t* d_coef, u64* d_hist, hipStream_t s, bool clear,
                            HistogramExport* ex) {
    if (clear) {
        const hipError_t e = launch_clear_histograms(d_hist, g.batch, s);
        if (e != hipSuccess) return e;
    }
    // Every workgroup takes the same number of 256-block passes.
    const long long passes = (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    // A workgroup ends with a flush of up to 1024 counters: a mid-sized image is better off
    // with two passes per workgroup than with twice the workgroups (4K 4:4:4: 40 -> 26 us),
    // a small one needs every workgroup it can get.
    long long cap = g.batch > 1 ? (4096 / g.batch > 8 ? 4096 / g.batch : 8) : 2048;
    if (g.batch == 1 && passes / 2 < cap) cap = passes / 2 > 256 ? passes / 2 : 256;
    const long long rounds = (passes + cap - 1) / cap;
    long long grid = (passes + rounds - 1) / rounds;
    if (grid < 1) grid = 1;
    HistogramExport none = {nullptr, 0, nullptr};
    if (ex != nullptr && g.batch == 1) ex->expected += (unsigned long long)grid;   // what the counter reads when all are done
    hipLaunchKernelGGL(k_histogram, dim3((unsigned)grid, (unsigned)g.batch), dim3(kEntThreads), 0, s, g, d_coef, d_hist,
                       ex != nullptr && g.batch == 1 ? *ex : none);
    return hipGetLastError();
}

// ---- bit lengths -------------------------------------------------------------------------
// Bits a block occupies in the stream under the given tables.
struct LengthVisitor {
    const u32* dcTab;
    const u32* acTab;
    u32 bits;
    __device__ __forceinline__ void dc(int cat, u32) { bits += (dcTab[cat] >> 16) + cat; }
    __device__ __forceinline__ void ac(int sym, int cat, u32) { bits += (acTab[sym] >> 16) + cat; }
    __device__ __forceinline__ void zrl(int n) { bits += n * (acTab[0xF0] >> 16); }
    __device__ __forceinline__ void eob() { bits += acTab[0] >> 16; }
};

// ---- workgroup scan ----------------------------------------------------------------------
constexpr int kScanThreads = 256;

__device__ __forceinline__ u64 wave_inclusive_scan(u64 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread workgroup; returns the
// exclusive prefix and the workgroup total.  lds: 4 values.
__device__ __forceinline__ u64 block_exclusive_scan(u64 v, u64* lds, u64& total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    u64 base = 0;
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < kScanThreads / 64; ++i) {
        if (i < wave) base += lds[i];
        sum += lds[i];
    }
    total = sum;
    return base + inc - v;
}

// ---- restart intervals -------------------------------------------------------------------
// The unstuffed stream holds, per interval, an RSTn marker (every interval but the image's
// first) and the interval's bit string padded to a whole byte; the band that ends the
// image appends EOI.  Markers travel inside the stream so that the stuffing pass needs no
// interval bookkeeping; it recognises them by position (is_marker_ff).
__device__ __forceinline__ int interval_marker_bytes(const KGeom& g, long long k) {
    return g.batch == 1 && g.firstInterval + k > 0 ? 2 : 0;
}

// Does EOI follow interval k?  The last interval of a band that ends its image; in a batch
// every interval is a whole image.
__device__ __forceinline__ bool interval_has_eoi(const KGeom& g, long long k) {
    return g.writeEoi && (g.batch > 1 || k == g.intervals - 1);
}

// ---- bit writer ------------------------------------------------------------------------
// One thread per block appends its codes MSB first.  A workgroup's 256 blocks cover one
// contiguous bit range; it is assembled in LDS (native word order, ds_or for the words two
// blocks share) and copied out byte swapped with coalesced stores, so that memory holds the
// big-endian bit string BitBuffer::pushBits builds (BitBuffer.cpp:128-204).  Only the
// range's first and last word are merged into the zeroed buffer with atomicOr.  Ranges too
// long for the LDS stage are written straight to memory.
constexpr int kEmitStageShort = 2048;   // stage words of a tile of short blocks (8 bytes per block on average)
constexpr int kEmitStageLong = 6144;    // ... of long blocks (up to 96 bytes per block on average)

// Where a lane's bits go: the LDS stage (native order) or the global stream (swapped).
struct BitTarget {
    u32* words;
    long long bias;   // index of the stream word that words[0] stands for
    bool swap;
    __device__ __forceinline__ void or_word(long long index, u32 v) const {
        if (v != 0) atomicOr(&words[index - bias], swap ? __builtin_bswap32(v) : v);
    }
    // ORs up to 32 bits, given left aligned in v, into the stream at bit position pos.
    __device__ __forceinline__ void or32(u64 pos, u32 v) const {
        const long long d = (long long)(pos >> 5);
        const int sh = (int)(pos & 31);
        or_word(d, v >> sh);
        if (sh != 0) or_word(d + 1, v << (32 - sh));
    }
};

// First walk: the block's bit string, starting at bit 0, into a few words of LDS private to
// the lane, and its length.  A block longer than the private words is only measured.
constexpr int kPrivWords = 8;
constexpr int kPrivPitch = 9;   // odd, so that the lanes' words spread over the banks

struct PrivateVisitor {
    const u32* dcTab;
    const u32* acTab;
    u32* words;
    u64 acc;    // pending bits, right aligned
    int n;      // number of pending bits (< 32 between calls)
    int word;   // words completed
    __device__ __forceinline__ void put(u32 bits, int len) {
        acc = (acc << len) | bits;
        n += len;
        if (n >= 32) {
            if (word < kPrivWords) words[word] = (u32)(acc >> (n - 32));
            ++word;
            n -= 32;
        }
    }
    __device__ __forceinline__ void dc(int cat, u32 amp) {
        const u32 e = dcTab[cat];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void ac(int sym, int cat, u32 amp) {
        const u32 e = acTab[sym];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void zrl(int cnt) {
        const u32 e = acTab[0xF0];
        for (int i = 0; i < cnt; ++i) put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void eob() {
        const u32 e = acTab[0];
        put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ u32 finish() {   // returns the length in bits
        if (n > 0 && word < kPrivWords) words[word] = (u32)(acc << (32 - n));
        return (u32)(word * 32 + n);
    }
};

// Second walk, for the blocks that did not fit their private words: straight to the target.
struct EmitVisitor {
    const u32* dcTab;
    const u32* acTab;
    BitTarget out;
    long long word;   // index of the word being filled
    u64 acc;          // pending bits, right aligned
    int n;            // number of pending bits (< 32 between calls)
    bool first;
    __device__ __forceinline__ void flush_word() {
        const u32 w = (u32)(acc >> (n - 32));
        if (first) {
            out.or_word(word, w);
            first = false;
        } else {
            out.words[word - out.bias] = out.swap ? __builtin_bswap32(w) : w;
        }
        ++word;
        n -= 32;
    }
    __device__ __forceinline__ void put(u32 bits, int len) {
        acc = (acc << len) | bits;
        n += len;
        if (n >= 32) flush_word();
    }
    // Code and amplitude go in together: at most 16 + 11 bits on top of fewer than 32 pending.
    __device__ __forceinline__ void dc(int cat, u32 amp) {
        const u32 e = dcTab[cat];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void ac(int sym, int cat, u32 amp) {
        const u32 e = acTab[sym];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void zrl(int cnt) {
        const u32 e = acTab[0xF0];
        for (int i = 0; i < cnt; ++i) put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void eob() {
        const u32 e = acTab[0];
        put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void finish() {
        if (n > 0) out.or_word(word, (u32)(acc << (32 - n)));
    }
};

// ---- single-pass bit writer --------------------------------------------------------------
// Bit lengths, stream positions and the bit writer in one pass over the coefficients.  A
// tile is 256 consecutive blocks of one image.  The stream position after a block is a
// function of the position before it: P + bits, or, behind the last block of a restart
// interval, roundup8(P + bits) + the marker that follows (the next interval's RSTn, or EOI).
// A run of blocks composes to "P + h" (no interval ends inside) or "roundup8(P + h) + m",
// an associative summary, so tiles chain with a decoupled look-back: every tile publishes
// its summary as soon as its lengths are known, then its first wave reads the summaries of
// its predecessors back to the nearest tile whose absolute position is already known.
// Tiles take their numbers from a counter, so every predecessor of a running tile is
// running or finished and the wait always ends.
struct PosSummary {
    u32 x;   // h, bit 31: an interval ends inside
    u32 y;   // m
};

__device__ __forceinline__ PosSummary pos_combine(PosSummary a, PosSummary b) {   // a precedes b
    PosSummary r;
    if ((a.x >> 31) == 0) {
        r.x = a.x + b.x;
        r.y = b.y;
    } else {
        const u32 bh = b.x & 0x7fffffffu;
        r.x = a.x;
        r.y = (b.x >> 31) ? ((a.y + bh + 7u) & ~7u) + b.y : a.y + bh;
    }
    return r;
}

__device__ __forceinline__ u64 pos_apply(PosSummary s, u64 p) {
    const u32 h = s.x & 0x7fffffffu;
    return (s.x >> 31) ? ((p + h + 7) & ~7ull) + s.y : p + h;
}

constexpr u64 kTileAggregate = 1ull << 62, kTilePosition = 2ull << 62, kTilePayload = (1ull << 62) - 1;
constexpr int kLookBackSpins = 1 << 18;

__device__ __forceinline__ PosSummary shfl_down_summary(PosSummary v, int d, int lane) {
    PosSummary o;
    o.x = __shfl_down(v.x, d, 64);
    o.y = __shfl_down(v.y, d, 64);
    if (lane + d >= 64) o.x = o.y = 0;
    return o;
}

// Absolute bit position at which tile `tile` starts; p0 is the position of the stream's
// first block.  Called by all 64 lanes of one wave.  Returns false if a predecessor never
// reported (cannot happen with ordered tile numbers; bounded so a fault cannot hang the GPU).
__device__ __forceinline__ bool look_back(const u64* state, long long tile, u64 p0, u64& position) {
    const int lane = threadIdx.x & 63;
    PosSummary acc = {0u, 0u};
    int spins = 0;
    for (long long base = tile - 1;; base -= 64) {
        const long long idx = base - lane;
        u64 st = idx < 0 ? (kTilePosition | p0) : __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__ballot((st >> 62) == 0) != 0) {
            if (++spins > kLookBackSpins) return false;
            __builtin_amdgcn_s_sleep(2);
            if ((st >> 62) == 0) st = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u64 known = __ballot((st >> 62) == 2);
        const int nearest = known ? __builtin_ctzll(known) : 64;   // lanes below it hold summaries only
        PosSummary v = {0u, 0u};
        if (lane < nearest) {
            v.x = (u32)((st & kTilePayload) >> 30);
            v.y = (u32)(st & 0x3fffffffu);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v = pos_combine(shfl_down_summary(v, d, lane), v);   // older tiles first
        PosSummary window;
        window.x = __shfl(v.x, 0, 64);
        window.y = __shfl(v.y, 0, 64);
        acc = pos_combine(window, acc);
        if (known) {
            const u64 p = __shfl(st & kTilePayload, nearest, 64);
            position = pos_apply(acc, p);
            return true;
        }
    }
}

// kPrivate: blocks are short on average (photographic content), so the first walk keeps
// each block's bits in a few private LDS words and the second walk is a copy; otherwise the
// first walk only measures and every block is walked twice.
template <bool kPrivate>
__global__ __launch_bounds__(kEntThreads) void k_encode_bits(KGeom g, const int16_t* __restrict__ coef,
                                                             const u32* __restrict__ tables, u64* __restrict__ state,
                                                             u64* __restrict__ ivOff, u64* __restrict__ result,
                                                             u32* __restrict__ words, u64* __restrict__ hostResult) {
    __shared__ u32 tab[4 * 256];
    constexpr int kStageWords = kPrivate ? kEmitStageShort : kEmitStageLong;
    __shared__ u32 stage[kStageWords];
    __shared__ u64 span[2];
    __shared__ PosSummary waveSum[kEntThreads / 64];
    __shared__ u32 priv[kPrivate ? kEntThreads * kPrivPitch : 1];
    __shared__ u64 tileStart;
    __shared__ long long tileNo;
    const long long tilesPerImage = (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    const long long tiles = tilesPerImage * g.batch;
    u64* counter = state + tiles;
    if (threadIdx.x == 0) tileNo = (long long)atomicAdd((unsigned long long*)counter, 1ull);
    __syncthreads();
    // Tile and block numbers fit 32 bits (see BlockRole); the tile number is the same for
    // the whole workgroup.
    const u32 tile = __builtin_amdgcn_readfirstlane((u32)tileNo);
    const u32 image = g.batch > 1 ? tile / (u32)tilesPerImage : 0u;
    const u32 bl = (tile - image * (u32)tilesPerImage) * kEntThreads + threadIdx.x;
    const long long b = (long long)image * g.blocksPerImage + bl;
    const bool active = bl < (u32)g.blocksPerImage;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long blocksPerInterval = g.mcusPerInterval * g.bpm;
    const BlockRole role = block_role(g, active ? bl : (u32)g.blocksPerImage - 1);

    u32 w[32];
    int fetchedDc = 0;
    if (active) {   // in flight while the tables are staged
        load_block(coef, b, w);
        fetchedDc = coef[dc_predictor_source(role, b) * 64];
    }
    const u32* myTables = tables + image * g.tableStride;
    for (int i = threadIdx.x; i < 1024; i += kEntThreads) tab[i] = myTables[i];
    __syncthreads();

    long long iv = 0;
    bool firstOfInterval = false, lastOfInterval = false;
    PosSummary mine = {0u, 0u};
    u32 bits = 0;
    const int c = role.c;
    const int pred = dc_predictor_lanes(role, active ? coef_at(w, 0) : 0, fetchedDc, active);
    if (active) {
        iv = g.intervals == g.batch ? (long long)image : (long long)(role.mcu / (u32)g.mcusPerInterval);
        const long long ivFirstBlock = iv * blocksPerInterval;
        long long ivEnd = ivFirstBlock + blocksPerInterval;
        if (ivEnd > g.totalBlocks) ivEnd = g.totalBlocks;
        firstOfInterval = b == ivFirstBlock;
        lastOfInterval = b == ivEnd - 1;
        const int t = c == 0 ? 0 : 1;
        if (kPrivate) {
            PrivateVisitor pv{tab + 256 * t, tab + 512 + 256 * t, priv + threadIdx.x * kPrivPitch, 0, 0, 0};
            walk_block(w, pred, pv);
            bits = pv.finish();
        } else {
            LengthVisitor len{tab + 256 * t, tab + 512 + 256 * t, 0u};
            walk_block(w, pred, len);
            bits = len.bits;
        }
        mine.x = bits;
        if (lastOfInterval) {
            mine.x |= 0x80000000u;
            mine.y = iv + 1 < g.intervals ? 8u * (u32)interval_marker_bytes(g, iv + 1) : 0u;
            if (interval_has_eoi(g, iv)) mine.y += 16u;
        }
    }
    // Inclusive scan of the summaries over the workgroup.
    PosSummary inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        PosSummary o;
        o.x = __shfl_up(inc.x, d, 64);
        o.y = __shfl_up(inc.y, d, 64);
        if (lane >= d) inc = pos_combine(o, inc);
    }
    if (lane == 63) waveSum[wave] = inc;
    __syncthreads();
    PosSummary before = {0u, 0u}, whole = {0u, 0u};
#pragma unroll
    for (int i = 0; i < kEntThreads / 64; ++i) {
        if (i < wave) before = pos_combine(before, waveSum[i]);
        whole = pos_combine(whole, waveSum[i]);
    }
    PosSummary excl;
    excl.x = __shfl_up(inc.x, 1, 64);
    excl.y = __shfl_up(inc.y, 1, 64);
    if (lane == 0) excl.x = excl.y = 0;
    excl = pos_combine(before, excl);

    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    if (wave == 0) {
        if (tile > 0 && lane == 0)
            __hip_atomic_store(state + tile, kTileAggregate | ((u64)whole.x << 30) | whole.y, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        u64 start = p0;
        bool ok = true;
        if (tile > 0) ok = look_back(state, tile, p0, start);
        if (lane == 0) {
            if (!ok) {
                start = p0;
                result[1] = 2;
                if (hostResult) hostResult[2] = 2;
            }
            __hip_atomic_store(state + tile, kTilePosition | pos_apply(whole, start), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            tileStart = start;
        }
    }
    __syncthreads();

    u64 pos = 0;
    int padBits = 0;
    u32 leadMarker = 0;
    bool tailEoi = false;
    if (active) {
        pos = pos_apply(excl, tileStart);
        u64 begin = pos;
        u64 end = pos + bits;
        if (firstOfInterval) {
            const int mk = interval_marker_bytes(g, iv);
            if (mk) {
                leadMarker = 0xffd0u + (u32)((g.firstInterval + iv - 1) & 7);
                begin -= 16;
            }
            ivOff[iv] = begin >> 3;
        }
        if (lastOfInterval) {   // pad the interval to a byte boundary (BitBuffer.cpp:12-16)
            padBits = (int)((8 - (end & 7)) & 7);
            end += padBits;
            if (interval_has_eoi(g, iv)) {
                tailEoi = true;
                end += 16;
            }
            if (b == g.totalBlocks - 1) {
                const u64 bytes = end >> 3;
                ivOff[g.intervals] = bytes;
                result[2] = bytes;
                result[3] = (bytes + kStuffChunk - 1) / kStuffChunk;
            }
        }
        if (threadIdx.x == 0) span[0] = begin;
        if (threadIdx.x == kEntThreads - 1 || bl == g.blocksPerImage - 1) span[1] = end;
        pos = begin;
    }
    __syncthreads();
    const u64 wordBase = span[0] >> 5;
    const u64 nWords = ((span[1] + 31) >> 5) - wordBase;
    const bool staged = nWords <= (u64)kStageWords;
    if (staged) {
        for (u32 i = threadIdx.x; i < (u32)nWords; i += kEntThreads) stage[i] = 0;
        __syncthreads();
    }
    if (active) {
        const BitTarget out{staged ? stage : words, staged ? (long long)wordBase : 0, !staged};
        const u64 start = pos + (leadMarker ? 16 : 0);   // first bit of the block itself
        if (leadMarker) out.or32(pos, leadMarker << 16);
        if (kPrivate && bits <= (u32)(kPrivWords * 32)) {
            const u32* mine32 = priv + threadIdx.x * kPrivPitch;
#pragma unroll
            for (int k = 0; k < kPrivWords; ++k)
                if ((u32)(32 * k) < bits) out.or32(start + 32 * k, mine32[k]);
        } else {
            const int t = c == 0 ? 0 : 1;
            EmitVisitor vis{tab + 256 * t, tab + 512 + 256 * t, out, (long long)(start >> 5), 0, (int)(start & 31), true};
            walk_block(w, pred, vis);
            vis.finish();
        }
        if (padBits != 0 && g.padOnes) out.or32(start + bits, ((1u << padBits) - 1u) << (32 - padBits));
        if (tailEoi) out.or32(start + bits + padBits, 0xffd9u << 16);
    }
    if (staged) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < (u32)nWords; i += kEntThreads) {
            const u32 v = __builtin_bswap32(stage[i]);
            if (i == 0 || i == (u32)nWords - 1) {
                if (v != 0) atomicOr(&words[wordBase + i], v);
            } else {
                words[wordBase + i] = v;
            }
        }
    }
}

// ---- byte stuffing -----------------------------------------------------------------------
__device__ __forceinline__ int count_ff(u32 w) {
    // A byte is 0xFF iff all its bits are set: fold the 8 bits of every byte into bit 0.
    u32 t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    return __popc(t & 0x01010101u);
}

// Is the 0xFF at byte position i (followed by `next`) the first byte of a marker that the
// bit writer placed in the stream?  Only reached for FF D0..D7 / FF D9 pairs, i.e. rarely.
__device__ __forceinline__ bool is_marker_ff(const KGeom& g, const u64* __restrict__ ivOff, u64 bytes, u64 i,
                                             uint8_t next) {
    (void)bytes;
    if (next != 0xd9 && (next < 0xd0 || next > 0xd7)) return false;
    long long lo = 0, hi = g.intervals - 1;   // last interval starting at or before i
    while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ivOff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    if (next == 0xd9) return interval_has_eoi(g, lo) && i + 2 == ivOff[lo + 1];
    return ivOff[lo] == i && interval_marker_bytes(g, lo) != 0;
}

__device__ __forceinline__ u32 ff_bits(u32 w) {   // bit k: byte k of w is 0xFF
    u32 t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    // The flags sit at bits 0, 8, 16, 24; the product brings them to bits 24..27 (no two of
    // the sixteen partial products meet, so nothing carries).
    return ((t & 0x01010101u) * 0x01020408u) >> 24;
}

// Loads 16 stream bytes (plus one of lookahead) and returns a 16-bit mask of the data bytes
// equal to 0xFF, i.e. those that need a 0x00 behind them (BitBuffer.cpp:156-161, 171-176,
// 190-195).  Only bytes of the stream count.
__device__ __forceinline__ u32 stuff_mask(const KGeom& g, const uint8_t* __restrict__ buf,
                                          const u64* __restrict__ ivOff, u64 bytes, long long i0, uint4& q) {
    q = *(const uint4*)(buf + i0);
    u32 ff = ff_bits(q.x) | (ff_bits(q.y) << 4) | (ff_bits(q.z) << 8) | (ff_bits(q.w) << 12);
    const long long left = (long long)bytes - i0;
    if (left < 16) ff &= left > 0 ? (1u << (int)left) - 1u : 0u;
    if (ff == 0) return 0;
    // One round per 0xFF byte of the piece (one or two as a rule) instead of one per byte.
    const u64 lo = (u64)q.x | ((u64)q.y << 32), hi = (u64)q.z | ((u64)q.w << 32);
    const u32 last = (u32)buf[i0 + 16];
    u32 mask = ff;
    for (u32 todo = ff; todo != 0; todo &= todo - 1) {
        const int j = __builtin_ctz(todo), n = j + 1;
        const uint8_t nxt = (uint8_t)(n < 8 ? lo >> (8 * n) : (n < 16 ? hi >> (8 * (n - 8)) : (u64)last));
        const bool maybe = (nxt >= 0xd0 && nxt <= 0xd7) || nxt == 0xd9;
        if (maybe && is_marker_ff(g, ivOff, bytes, (u64)(i0 + j), nxt)) mask &= ~(1u << j);
    }
    return mask;
}

// One wave per 4 KiB chunk: four 16-byte pieces per lane in flight, a shuffle reduction, no
// barriers.
constexpr int kChunkGroup = 64;   // chunks whose counts are also kept as one sum

__global__ __launch_bounds__(256) void k_count_ff(KGeom g, const uint8_t* __restrict__ buf,
                                                  const u64* __restrict__ ivOff, const u64* __restrict__ result,
                                                  u32* __restrict__ ffCount) {
    const u64 bytes = result[2];
    const long long chunks = (long long)((bytes + kStuffChunk - 1) / kStuffChunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long ch = (long long)blockIdx.x * 4 + wave; ch < chunks; ch += (long long)gridDim.x * 4) {
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kStuffChunk / 1024; ++k) {
            const long long i0 = ch * kStuffChunk + k * 1024 + lane * 16;
            if ((u64)i0 < bytes) {
                uint4 q;
                cnt += __popc(stuff_mask(g, buf, ivOff, bytes, i0, q));
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if (lane == 0) ffCount[ch] = (u32)cnt;
    }
}

// For long streams: the counts of kChunkGroup consecutive chunks added up, one wave per sum.
__global__ __launch_bounds__(256) void k_group_sums(const u32* __restrict__ ffCount, const u64* __restrict__ result,
                                                    u64* __restrict__ groupSum) {
    static_assert(kChunkGroup == 64, "one lane per chunk of a group");
    const long long chunks = (long long)((result[2] + kStuffChunk - 1) / kStuffChunk);
    const int lane = threadIdx.x & 63;
    const long long grp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp * kChunkGroup >= chunks) return;
    const long long c = grp * kChunkGroup + lane;
    u64 sum = c < chunks ? ffCount[c] : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) groupSum[grp] = sum;
}

// Stuffing bytes in front of chunk ch, added up by the 256 threads of a workgroup from the
// per-chunk counts; cheap enough once per workgroup that the stream needs no device-wide
// scan kernel.  Long streams (groupSum set) have the counts of whole groups added up already.
__device__ __forceinline__ u64 chunk_prefix(const u64* __restrict__ groupSum, const u32* __restrict__ ffCount,
                                            long long ch, u64* lds) {
    u64 v = 0;
    if (groupSum != nullptr) {
        const long long grp = ch / kChunkGroup;
        for (long long i = threadIdx.x; i < grp; i += 256) v += groupSum[i];
        const long long c = grp * kChunkGroup + threadIdx.x;
        if (threadIdx.x < kChunkGroup && c < ch) v += ffCount[c];
    } else {
        // Eight loads in flight per thread: one after the other they would cost a memory
        // latency each.
        for (long long i0 = threadIdx.x; i0 < ch; i0 += 8 * 256) {
            u32 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = i0 + k * 256 < ch ? ffCount[i0 + k * 256] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
    }
    u64 total;
    block_exclusive_scan(v, lds, total);
    __syncthreads();   // lds is free again
    return total;
}

// Copies the stream to the output with the stuffing bytes inserted.  A workgroup's 4 KiB
// chunk becomes one contiguous output range, which is laid out in LDS at the output's own
// 4-byte phase and stored as dwords.  In a batch the stream of image k goes behind header k
// in slot k; a chunk that holds the end of one image and the start of the next is written
// byte by byte instead.
constexpr int kStuffStageBytes = 2 * kStuffChunk + 16;

// Interval (= image of a batch) that contains stream byte i.
__device__ __forceinline__ long long interval_of(const KGeom& g, const u64* __restrict__ ivOff, u64 i) {
    long long lo = 0, hi = g.intervals - 1;
    while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ivOff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Output position of stream byte i of image k, given the stuffing bytes in front of it.
__device__ __forceinline__ long long batch_position(const BatchOutput& bo, const u64* __restrict__ ivOff, long long k,
                                                    u64 i, u64 ffBefore) {
    return k * bo.slotBytes + (long long)bo.hdrLen[k] + (long long)(i - ivOff[k]) + (long long)(ffBefore - bo.ffStart[k]);
}

template <bool kBatch>
__global__ __launch_bounds__(256) void k_stuff_bytes(KGeom g, const uint8_t* __restrict__ buf,
                                                     const u64* __restrict__ ivOff, const u32* __restrict__ ffCount,
                                                     const u64* __restrict__ groupSum, u64* __restrict__ result,
                                                     uint8_t* __restrict__ out, long long prefixLen,
                                                     long long outCapacity, BatchOutput bo, u64* __restrict__ hostResult) {
    __shared__ u64 lds[4];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStuffStageBytes];
    const u64 bytes = result[2];
    const long long chunks = (long long)((bytes + kStuffChunk - 1) / kStuffChunk);
    // A workgroup takes a run of consecutive chunks: the stuffing bytes in front of the
    // first come from the counting pass, those in front of the others from its own work.
    // The runs differ by one chunk at most, so that every workgroup of the grid has work.
    const long long chFirst = chunks * blockIdx.x / gridDim.x;
    const long long chEnd = chunks * (blockIdx.x + 1) / gridDim.x;
    if (chFirst >= chEnd) return;
    u64 ffBefore = chunk_prefix(groupSum, ffCount, chFirst, lds);
    for (long long ch = chFirst; ch < chEnd; ++ch) {
        const long long c0 = ch * kStuffChunk;
        const long long i0 = c0 + (long long)threadIdx.x * 16;
        uint4 q = make_uint4(0, 0, 0, 0);
        int valid = 0;
        u32 mask = 0;
        if ((u64)i0 < bytes) {
            valid = (int)(bytes - (u64)i0 < 16 ? bytes - (u64)i0 : 16);
            mask = stuff_mask(g, buf, ivOff, bytes, i0, q);
        }
        u64 total;
        const u64 local = block_exclusive_scan((u64)(valid + __popc(mask)), lds, total);
        // Output range of the chunk: one piece, except for a batch chunk that spans images.
        long long base, limit;
        bool onePiece = true;
        if (kBatch) {
            const long long k = interval_of(g, ivOff, (u64)c0);
            const u64 cEnd = (u64)c0 + kStuffChunk < bytes ? (u64)c0 + kStuffChunk : bytes;
            onePiece = cEnd <= ivOff[k + 1];
            base = batch_position(bo, ivOff, k, (u64)c0, ffBefore);
            limit = (k + 1) * bo.slotBytes;
        } else {
            base = prefixLen + c0 + (long long)ffBefore;
            limit = outCapacity;
        }
        if (onePiece) {
            if (base + (long long)total > limit) {
                if (threadIdx.x == 0) {
                    result[1] = 1;
                    if (hostResult) hostResult[1] = 1;
                }
            } else {
                const int phase = (int)((uintptr_t)(out + base) & 3);
                if (valid > 0) {
                    uint8_t* dst = stage + phase + (int)local;
                    const u32 ws[4] = {q.x, q.y, q.z, q.w};
                    if (mask == 0 && valid == 16) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) dst[j] = (uint8_t)(ws[j >> 2] >> ((j & 3) * 8));
                    } else {
                        int o = 0;
                        for (int j = 0; j < valid; ++j) {
                            dst[o++] = (uint8_t)(ws[j >> 2] >> ((j & 3) * 8));
                            if ((mask >> j) & 1) dst[o++] = 0x00;
                        }
                    }
                }
                __syncthreads();
                const int nTot = phase + (int)total;
                uint8_t* dst = out + base - phase;   // 4-byte aligned
                for (int d = threadIdx.x; d * 4 < nTot; d += 256) {
                    const int lo4 = d * 4;
                    if (lo4 >= phase && lo4 + 4 <= nTot) {
                        *(u32*)(dst + lo4) = *(const u32*)(stage + lo4);
                    } else {
                        for (int bb = lo4 < phase ? phase : lo4; bb < lo4 + 4 && bb < nTot; ++bb) dst[bb] = stage[bb];
                    }
                }
            }
        } else if (valid > 0) {
            // Batch chunk across an image boundary: every byte finds its own slot.
            const u32 ws[4] = {q.x, q.y, q.z, q.w};
            long long k = interval_of(g, ivOff, (u64)i0);
            u64 ff = ffBefore + (local - (u64)((long long)i0 - c0));   // stuffing bytes in front of byte i0
            bool overflow = false;
            for (int j = 0; j < valid; ++j) {
                const u64 i = (u64)(i0 + j);
                while (i >= ivOff[k + 1]) ++k;
                const long long pos = batch_position(bo, ivOff, k, i, ff);
                const int extra = (mask >> j) & 1;
                if (pos + 1 + extra <= (k + 1) * bo.slotBytes) {
                    out[pos] = (uint8_t)(ws[j >> 2] >> ((j & 3) * 8));
                    if (extra) out[pos + 1] = 0x00;
                } else {
                    overflow = true;
                }
                ff += extra;
            }
            if (overflow) result[1] = 1;
        }
        if (!kBatch && ch == chunks - 1 && threadIdx.x == 0) {
            result[0] = (u64)(base + (long long)total);
            if (hostResult) hostResult[0] = (u64)(base + (long long)total);
        }
        const u64 chunkBytes = bytes - (u64)c0 < (u64)kStuffChunk ? bytes - (u64)c0 : (u64)kStuffChunk;
        ffBefore += total - chunkBytes;   // what the chunk wrote beyond its own bytes
        __syncthreads();
    }
}

// Batch: stuffing bytes in front of each image's stream (and, for k = batch, in total).
__global__ __launch_bounds__(256) void k_image_ff_start(KGeom g, const uint8_t* __restrict__ buf,
                                                        const u64* __restrict__ ivOff, const u32* __restrict__ ffCount,
                                                        const u64* __restrict__ groupSum,
                                                        const u64* __restrict__ result, u64* __restrict__ ffStart) {
    __shared__ u64 lds[4];
    const u64 bytes = result[2];
    const long long k = blockIdx.x;
    const u64 pos = ivOff[k];
    const long long ch = (long long)(pos / kStuffChunk);
    const long long i0 = ch * kStuffChunk + (long long)threadIdx.x * 16;
    int cnt = 0;
    if ((u64)i0 < pos) {
        uint4 q;
        u32 mask = stuff_mask(g, buf, ivOff, bytes, i0, q);
        const u64 span = pos - (u64)i0;
        if (span < 16) mask &= (1u << span) - 1u;
        cnt = __popc(mask);
    }
    u64 total;
    block_exclusive_scan((u64)cnt, lds, total);
    __syncthreads();
    const u64 before = chunk_prefix(groupSum, ffCount, ch, lds);
    if (threadIdx.x == 0) ffStart[k] = before + total;
}

// Batch: header k to the front of slot k.
__global__ __launch_bounds__(256) void k_copy_headers(BatchOutput bo) {
    const long long k = blockIdx.x;
    const uint8_t* src = bo.hdr + k * bo.hdrPitch;
    uint8_t* dst = bo.out + k * bo.slotBytes;
    const u32 n = bo.hdrLen[k];
    for (u32 i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// Everything the single-pass bit writer and the stuffing pass expect to find cleared, in one
// launch: the reachable part of the unstuffed stream, the tile states with their counter,
// the result record.  Also moves the file header from its staging place to the output.
__global__ __launch_bounds__(256) void k_prepare_entropy(uint4* __restrict__ stream, long long stream16,
                                                         u64* __restrict__ tileState, long long tileWords,
                                                         u64* __restrict__ result, const uint8_t* __restrict__ hdr,
                                                         int hdrLen, uint8_t* __restrict__ out,
                                                         const u32* __restrict__ pending, u32* __restrict__ tables) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    if (pending != nullptr) {
        // Tables and header still sit in pinned host memory: bring them over (the device
        // copy serves later launches) and take the header for the output from there.
        const int words = 1024 + (hdrLen + 3) / 4;
        for (long long i = tid; i < words; i += step) tables[i] = pending[i];
        hdr = (const uint8_t*)(pending + 1024);
    }
    for (long long i = tid; i < stream16; i += step) stream[i] = make_uint4(0, 0, 0, 0);
    for (long long i = tid; i < tileWords; i += step) tileState[i] = 0;
    if (tid < 4) result[tid] = 0;
    for (long long i = tid; i < hdrLen; i += step) out[i] = hdr[i];
}

static hipError_t run_entropy_stage(const KGeom& g, const int16_t* d_coef, const u32* d_tables,
                                    const EntropyWorkspace& ws, uint8_t* d_out, long long prefixLen,
                                    long long outCapacity, const BatchOutput* bo, hipStream_t s) {
    const long long tilesPerImage = (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    const long long tileCount = tilesPerImage * g.batch;
    // The bit writer ORs into the stream, so the part it can reach starts out as zeros.
    long long zero = ws.streamBound < 0 ? ws.unstuffedCap : (ws.streamBound + kStuffChunk + 63) / 16 * 16;
    if (zero > ws.unstuffedCap) zero = ws.unstuffedCap;
    long long prepGrid = zero / 16 / (256 * 4);
    prepGrid = prepGrid < 1 ? 1 : (prepGrid > 2048 ? 2048 : prepGrid);
    // Chunks the stream can have at most: from the bound that was cleared.
    long long chunkBound = zero / kStuffChunk + 1;
    if (chunkBound > ws.maxChunks) chunkBound = ws.maxChunks;
    hipLaunchKernelGGL(k_prepare_entropy, dim3((unsigned)prepGrid), dim3(256), 0, s, (uint4*)ws.unstuffed, zero / 16,
                       ws.tileState, tileCount + 1, ws.result, ws.header, bo ? 0 : ws.headerLen, d_out,
                       bo ? nullptr : ws.pendingTables, const_cast<u32*>(d_tables));
    // Long blocks (noise, q=100) do not fit the private words; unknown lengths count as short.
    const bool shortBlocks = ws.avgBlockBits <= 128;
    if (shortBlocks)
        hipLaunchKernelGGL(k_encode_bits<true>, dim3((unsigned)tileCount), dim3(kEntThreads), 0, s, g, d_coef, d_tables,
                           ws.tileState, ws.ivOff, ws.result, (u32*)ws.unstuffed, bo ? nullptr : ws.hostResult);
    else
        hipLaunchKernelGGL(k_encode_bits<false>, dim3((unsigned)tileCount), dim3(kEntThreads), 0, s, g, d_coef, d_tables,
                           ws.tileState, ws.ivOff, ws.result, (u32*)ws.unstuffed, bo ? nullptr : ws.hostResult);
    long long chunkGrid = chunkBound < 4096 ? chunkBound : 4096;
    if (chunkGrid < 1) chunkGrid = 1;
    long long countGrid = (chunkBound + 3) / 4;
    countGrid = countGrid < 1 ? 1 : (countGrid > 4096 ? 4096 : countGrid);
    hipLaunchKernelGGL(k_count_ff, dim3((unsigned)countGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff, ws.result,
                       ws.ffCount);
    // Up to this many chunks a workgroup adds the counts in front of its first chunk up
    // directly (64 loads per thread at most); beyond, group sums come first.
    const u64* groupSum = nullptr;
    if (chunkBound > (ws.groupSumsFrom > 0 ? ws.groupSumsFrom : 16384)) {
        const long long groups = chunkBound / kChunkGroup + 1;
        hipLaunchKernelGGL(k_group_sums, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, s, ws.ffCount, ws.result,
                           ws.groupSum);
        groupSum = ws.groupSum;
    }
    if (bo) {
        hipLaunchKernelGGL(k_image_ff_start, dim3((unsigned)(g.batch + 1)), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                           ws.ffCount, groupSum, ws.result, bo->ffStart);
        hipLaunchKernelGGL(k_copy_headers, dim3((unsigned)g.batch), dim3(256), 0, s, *bo);
        hipLaunchKernelGGL(k_stuff_bytes<true>, dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                           ws.ffCount, groupSum, ws.result, bo->out, 0, 0, *bo, nullptr);
    } else {
        BatchOutput none = {};
        hipLaunchKernelGGL(k_stuff_bytes<false>, dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                           ws.ffCount, groupSum, ws.result, d_out, prefixLen, outCapacity, none, ws.hostResult);
    }
    return hipGetLastError();
}

hipError_t launch_entropy(const KGeom& g, const int16_t* d_coef, const u32* d_tables, const EntropyWorkspace& ws,
                          uint8_t* d_out, long long prefixLen, long long outCapacity, int writeEoi, hipStream_t s) {
    (void)writeEoi;   // carried in g.writeEoi
    return run_entropy_stage(g, d_coef, d_tables, ws, d_out, prefixLen, outCapacity, nullptr, s);
}

hipError_t launch_entropy_batch(const KGeom& g, const int16_t* d_coef, const u32* d_tables,
                                const EntropyWorkspace& ws, const BatchOutput& bo, hipStream_t s) {
    return run_entropy_stage(g, d_coef, d_tables, ws, nullptr, 0, 0, &bo, s);
}

}  // namespace jpeghip
