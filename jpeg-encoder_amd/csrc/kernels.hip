// Device kernels of the MI355X (gfx950) JPEG encode hot path.
//
// Build with -ffp-contract=off: the transform must round every multiply and add on its
// own, exactly like the reference's AVX path (SIMD256.h:155-186, 337-441).  Division is
// the compiler's correctly rounded fp32 division, rounding is v_rndne_f32 (ties to even).
#include "device_common.h"
#include "kernels.h"

#include "jpegenc_hip.h"

#include <cstdlib>

namespace jpeghip {

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

// Largest magnitude a quantized coefficient is allowed to have.  Byte pixels cannot get
// near it; float pixels outside 0..1 (or a channel gain) can, and a value beyond 16 bits
// would wrap in the int16 store and send the entropy stage past its 256-entry tables.  The
// reference asserts instead (Encoder.cpp:269-270, 309-310); out-of-range input is clamped
// here so that it stays a caller error and never becomes a memory error.
#define COEF_LIMIT 2047.0f

__device__ __forceinline__ float clamp_coef(float v) {
    return __builtin_fminf(__builtin_fmaxf(v, -COEF_LIMIT), COEF_LIMIT);   // a NaN becomes -COEF_LIMIT
}

// Which of a block's eight groups of eight zig-zag ordered coefficients (16 bytes each) hold a
// non-zero value: bit k for group k.  The entropy stage reads this byte per block instead of
// looking at the empty groups themselves.
__device__ __forceinline__ u32 group_mask(const u32 (&p)[32]) {
    u32 m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u32 any = p[4 * k] | p[4 * k + 1] | p[4 * k + 2] | p[4 * k + 3];
        m |= (any != 0 ? 1u : 0u) << k;
    }
    return m;
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
                                                                   int16_t* __restrict__ coef, unsigned char* __restrict__ gmask,
                                                                   u64* __restrict__ clearHist) {
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
        const int lo = (int)rintf(clamp_coef(v[kZZ[z]] / qc[kZZ[z]]));
        const int hi = (int)rintf(clamp_coef(v[kZZ[z + 1]] / qc[kZZ[z + 1]]));
        packed[z >> 1] = ((u32)lo & 0xffffu) | ((u32)hi << 16);
    }
    uint4* dst = (uint4*)((unsigned char*)coef + b * g.coefPitch);                // groups 0..3
    uint4* dstHigh = (uint4*)((unsigned char*)coef + b * g.coefPitch + g.coefHigh);   // groups 4..7
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_uint4(packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) dstHigh[i] = make_uint4(packed[16 + 4 * i], packed[17 + 4 * i], packed[18 + 4 * i], packed[19 + 4 * i]);
    gmask[b] = (unsigned char)group_mask(packed);
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

// The colour matrix (SIMD256.h:155-186) on the same pixel of two rows.  A product with 0.5 is
// exact, so "t + 0.5 * b" rounds once, like fma(b, 0.5, t): the two terms with the factor 0.5
// take one instruction each instead of two (an input small enough for 0.5 * b to be inexact
// changes a sum that the bias of 0.5 absorbs right after).
__device__ __forceinline__ void ycbcr_pair(f2 r, f2 g, f2 b, f2& y, f2& cb, f2& cr) {
    f2 t = r * 0.2990f;
    t = t + g * 0.5870f;
    y = t + b * 0.1140f;
    t = r * -0.1687f;
    t = t + g * -0.3312f;
    t = fma2(b, splat(0.5000f), t);
    cb = t + 0.5f;
    const f2 u = g * -0.4186f;
    t = fma2(r, splat(0.5000f), u);   // 0.5 * r + (-0.4186 * g), as t = 0.5 * r; t = t + -0.4186 * g
    t = t + b * -0.0813f;
    cr = t + 0.5f;
}

__device__ __forceinline__ void pixel_ycbcr_pair(const u32 (&ra)[6], const u32 (&rb)[6], int i, f2& y, f2& cb, f2& cr) {
    const f2 r = unorm8_pair(ra[(3 * i) >> 2], rb[(3 * i) >> 2], (3 * i) & 3);
    const f2 g = unorm8_pair(ra[(3 * i + 1) >> 2], rb[(3 * i + 1) >> 2], (3 * i + 1) & 3);
    const f2 b = unorm8_pair(ra[(3 * i + 2) >> 2], rb[(3 * i + 2) >> 2], (3 * i + 2) & 3);
    ycbcr_pair(r, g, b, y, cb, cr);
}

// x255 (SIMD256.h:484-494), transform, quantize, zig-zag: row-pair values in, the block's
// 32 dwords of int16 pairs out.  kClamp: for pixels that are not bytes.
template <bool kClamp = false>
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
        f2 quot = fma2(b[i], hi, b[i] * lo);
        if (kClamp) quot = (f2){clamp_coef(quot.x), clamp_coef(quot.y)};
        const f2 t = quot + ROUND_MAGIC;
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

// x255, transform, quantize, zig-zag on row-pair values (see aan2d_pairs); the wave's 64 blocks go through LDS so that every store
// instruction writes whole 128-byte lines (8 lanes per block) instead of 64 scattered
// 16-byte pieces.  `stage` is this wave's area of kRounds-th of 64 x 144 bytes (the 16
// bytes of padding per block keep the 16-byte LDS accesses conflict free); with kRounds
// = 2 the two half waves take turns, which halves the LDS footprint.  blockStride:
// distance in blocks between the blocks of consecutive lanes in the coefficient buffer;
// lanes at or beyond `count` hold no block.
constexpr int kStagePitch = 144;

__device__ __forceinline__ void store_stream(uint4* at, uint4 v) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (u32x4*)at);
}

// Where a wave's blocks go: the address of the first block's groups 0..3, the distance to the
// next block's, and the distance from a block's groups 0..3 to its groups 4..7 (KGeom::coefPitch,
// KGeom::coefHigh).
struct CoefDst {
    unsigned char* p;
    u32 pitch;
    long long high;
    bool stream;   // streaming stores, see coef_dst
};
template <bool kMayStream = true>
__device__ __forceinline__ CoefDst coef_dst(const KGeom& g, int16_t* __restrict__ coef, long long block) {
    // Batch passes store their coefficients with streaming stores: a pass alternates between transform
    // and histogram launches, and coefficients that wait in the memory-side cache are written back
    // while the histogram parts read (256 x 1080p: histogram parts 0.362 -> 0.300 ms, transform 0.631
    // -> 0.638).  For one large image it is a wash (transform + 2.5 us, histogram pass - 3.5 us).
    // (kMayStream false: kernels of small launches, which do without the test in front of every store.)
    return CoefDst{(unsigned char*)coef + block * g.coefPitch, (u32)g.coefPitch, g.coefHigh, kMayStream && g.batch > 1};
}

template <int kRounds, bool kClamp = false>
__device__ __forceinline__ void finish_block_staged(f2 (&a)[32], const float* __restrict__ rhi,
                                                    const float* __restrict__ rlo, unsigned char* stage,
                                                    CoefDst waveBase, unsigned char* __restrict__ waveMask,
                                                    int blockStride, int count) {
    const int lane = threadIdx.x & 63;
    const long long laneOff = (lane & 3) * 16 + ((lane & 4) ? waveBase.high : 0ll);   // eight lanes per block, 16 bytes each
    constexpr int kPerRound = 64 / kRounds;
    u32 p[32];
    transform_quantize_pairs<kClamp>(a, rhi, rlo, p);
    if (lane < count) waveMask[lane * blockStride] = (unsigned char)group_mask(p);
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
                uint4* at = (uint4*)(waveBase.p + (long long)blk * blockStride * waveBase.pitch + laneOff);
                if (waveBase.stream) store_stream(at, t);
                else *at = t;
            }
        }
        if (kRounds > 1) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Luma variant for the 4:2:0 kernel: lane l holds block (l & 3) of the wave's MCU (l >> 2);
// an MCU's four luma blocks are contiguous (512 bytes, or 256 in each of the two planes), MCUs are six blocks apart.  Two
// half-wave rounds through a 32-block stage.
template <bool kClamp = false>
__device__ __forceinline__ void finish_block_staged_y(f2 (&a)[32], const float* __restrict__ rhi,
                                                      const float* __restrict__ rlo, unsigned char* stage,
                                                      CoefDst waveBase, unsigned char* __restrict__ waveMask,
                                                      int mcuCount) {
    const int lane = threadIdx.x & 63;
    const long long laneOff = (lane & 3) * 16 + ((lane & 4) ? waveBase.high : 0ll);
    u32 p[32];
    transform_quantize_pairs<kClamp>(a, rhi, rlo, p);
    if ((lane >> 2) < mcuCount) waveMask[(lane >> 2) * 6 + (lane & 3)] = (unsigned char)group_mask(p);
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
                uint4* at = (uint4*)(waveBase.p + ((long long)mcu * 6 + (blk & 3)) * waveBase.pitch + laneOff);
                if (waveBase.stream) store_stream(at, t);
                else *at = t;
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
                                                       unsigned char* __restrict__ gmask, u64* __restrict__ clearHist) {
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
                // (row0 + row1) / 2.0f down (SIMD256.h:295-304).  Halving is exact and commutes
                // with the rounding of the sum (byte pixels keep every term far from the ends of
                // the exponent range): ((a + b) / 2 + (c + d) / 2) / 2 == ((a + b) + (c + d)) / 4,
                // each sum rounded as before.
                const f2 scb = cb[2 * j] + cb[2 * j + 1];
                const f2 scr = cr[2 * j] + cr[2 * j + 1];
                ocb[j] = (scb.x + scb.y) * 0.25f;
                ocr[j] = (scr.x + scr.y) * 0.25f;
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
        finish_block_staged_y(a, rhi, rlo, stage + wave * 32 * kStagePitch, coef_dst(g, coef, (mBase + wave * 16) * 6),
                              gmask + (mBase + wave * 16) * 6, waveMcus);
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
                               coef_dst(g, coef, mBase * 6 + 4 + ch), gmask + mBase * 6 + 4 + ch, 6, mcuCount);
    }
}

// 4:2:0 from float pixels (Image::setRawPixelData's input, values in 0..1): k_transform_420 with
// 96-byte rows.  A thread's eight rows are 768 bytes, too many registers to ask for at once and
// too many round trips to ask for one by one: two rows are converted while the next two are on
// their way.  Rows start on 16-byte boundaries (the launcher checks); blocks that reach past
// the image read single floats with the edge replicated.
__device__ __forceinline__ void load_rows2f(const unsigned char* __restrict__ rgb, long long stride, int x0, int y, bool interior,
                                            int W, int rows, float (&ra)[24], float (&rb)[24]) {
    if (interior) {
        const float4* pa = (const float4*)(rgb + (long long)y * stride + (long long)x0 * 12);
        const float4* pb = (const float4*)(rgb + (long long)(y + 1) * stride + (long long)x0 * 12);
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float4 u = pa[d], v = pb[d];
            ra[4 * d] = u.x; ra[4 * d + 1] = u.y; ra[4 * d + 2] = u.z; ra[4 * d + 3] = u.w;
            rb[4 * d] = v.x; rb[4 * d + 1] = v.y; rb[4 * d + 2] = v.z; rb[4 * d + 3] = v.w;
        }
    } else {
        const unsigned char* rowA = rgb + (long long)min(y, rows - 1) * stride;
        const unsigned char* rowB = rgb + (long long)min(y + 1, rows - 1) * stride;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long off = (long long)min(x0 + i, W - 1) * 12;
            const float* a = (const float*)(rowA + off);
            const float* b = (const float*)(rowB + off);
            ra[3 * i] = a[0]; ra[3 * i + 1] = a[1]; ra[3 * i + 2] = a[2];
            rb[3 * i] = b[0]; rb[3 * i + 1] = b[1]; rb[3 * i + 2] = b[2];
        }
    }
}

__global__ __launch_bounds__(256) void k_transform_420f(KGeom g, const unsigned char* __restrict__ rgb,
                                                        const float* __restrict__ qtab, int16_t* __restrict__ coef,
                                                        unsigned char* __restrict__ gmask, u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    __shared__ __attribute__((aligned(16))) float chroma[2 * k420Mcus * kChromaPitch];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 32 * kStagePitch];
    const float* rhi = qtab + 192;
    const float* rlo = qtab + 384;
    const int tid = threadIdx.x, q = tid & 3, ml = tid >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long mBase = (long long)blockIdx.x * k420Mcus;
    const long long left = g.nMcu - mBase;
    const int mcuCount = left < k420Mcus ? (int)left : k420Mcus;
    const long long m = mBase + (ml < mcuCount ? ml : mcuCount - 1);   // idle lanes redo the last MCU, store nothing
    {
        int mx, my;
        rgb += locate_mcu(g, m, mx, my);
        const int x0 = mx * 16 + (q & 1) * 8, y0 = my * 16 + (q >> 1) * 8;
        const bool interior = x0 + 8 <= g.W && y0 + 8 <= g.rowsAvail;
        f2 a[32];   // a[rp * 8 + i] = Y of pixel i in rows 2rp and 2rp + 1
        float* cbDst = chroma + ml * kChromaPitch + (q >> 1) * 32 + (q & 1) * 8;   // layout as in k_transform_420
        float* crDst = cbDst + k420Mcus * kChromaPitch;
        float ra[24], rb[24];
        load_rows2f(rgb, g.strideBytes, x0, y0, interior, g.W, g.rowsAvail, ra, rb);
        float ecb[4], ecr[4];   // chroma row of the even row pair, waiting for its odd partner
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            f2 cb[8], cr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ycbcr_pair((f2){ra[3 * i], rb[3 * i]}, (f2){ra[3 * i + 1], rb[3 * i + 1]}, (f2){ra[3 * i + 2], rb[3 * i + 2]},
                           a[8 * rp + i], cb[i], cr[i]);
            __builtin_amdgcn_sched_barrier(0);
            // The rows are used up: the next two are asked for into the same registers and travel
            // while the chroma of these two is folded.
            if (rp < 3) load_rows2f(rgb, g.strideBytes, x0, y0 + 2 * rp + 2, interior, g.W, g.rowsAvail, ra, rb);
            __builtin_amdgcn_sched_barrier(0);
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
            __builtin_amdgcn_sched_barrier(0);
        }
        const int waveMcus = mcuCount - wave * 16;   // MCUs of this wave that exist (may be <= 0)
        finish_block_staged_y<true>(a, rhi, rlo, stage + wave * 32 * kStagePitch, coef_dst(g, coef, (mBase + wave * 16) * 6),
                                    gmask + (mBase + wave * 16) * 6, waveMcus);
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
        finish_block_staged<2, true>(a, rhi + 64 * (1 + ch), rlo + 64 * (1 + ch), stage + wave * 32 * kStagePitch,
                                     coef_dst(g, coef, mBase * 6 + 4 + ch), gmask + mBase * 6 + 4 + ch, 6, mcuCount);
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
                                                                     int16_t* __restrict__ coef, unsigned char* __restrict__ gmask,
                                                                     u64* __restrict__ clearHist) {
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
    finish_block_staged<2>(a, rhi, rlo, stage + c * 32 * kStagePitch, coef_dst<false>(g, coef, m0w * 3 + c), gmask + m0w * 3 + c, 3, count);
}

// 4:4:4 with the conversion shared between the channels: one wave per workgroup, one lane per
// block position (three blocks).  A lane converts its 64 pixels once, keeps the three
// channels in registers (214 of them: two waves per SIMD) and runs the transform three times.
// Against the three-waves form above this drops two of the three byte -> float normalisations
// of every pixel, a quarter of that kernel's instructions, and has a third of its waves: a
// wave issues an instruction only every eighth cycle or so, so a SIMD wants four and more
// resident waves to fill its pipeline, which this form cannot give it.  Measured: 3840 x 2160
// both 23 us, 8192 x 8192 132 against 139 us; small images need the other form's many waves.
__global__ __launch_bounds__(64) void k_transform_444_shared(KGeom g, const unsigned char* __restrict__ rgb,
                                                             const float* __restrict__ qtab, int16_t* __restrict__ coef,
                                                             unsigned char* __restrict__ gmask, u64* __restrict__ clearHist) {
    clear_histograms(clearHist);
    __shared__ __attribute__((aligned(16))) unsigned char stage[16 * kStagePitch];
    const float* rhi = qtab + 192;
    const float* rlo = qtab + 384;
    const int lane = threadIdx.x;
    const long long m0w = (long long)blockIdx.x * 64;
    const long long m = m0w + lane;
    const long long left = g.nMcu - m0w;
    const int count = left < 64 ? (int)left : 64;
    const long long mc = m < g.nMcu ? m : g.nMcu - 1;   // idle lanes redo the last position, store nothing
    int mx, my;
    rgb += locate_mcu(g, mc, mx, my);
    const int x0 = mx * 8, y0 = my * 8;
    const bool interior = x0 + 8 <= g.W && y0 + 8 <= g.rowsAvail;
    u32 raw[8][6];
#pragma unroll
    for (int r = 0; r < 8; ++r) load_row8(rgb, g.strideBytes, x0, y0 + r, interior, g.W, g.rowsAvail, raw[r]);
    __builtin_amdgcn_sched_barrier(0);
    f2 ay[32], acb[32], acr[32];   // a[rp * 8 + i]: pixel i of rows 2rp and 2rp + 1
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pixel_ycbcr_pair(raw[2 * rp], raw[2 * rp + 1], i, ay[8 * rp + i], acb[8 * rp + i], acr[8 * rp + i]);
            // Finished here and now: left alone, the compiler moves the chroma sums behind the
            // luma transform and keeps the normalised r, g and b of every pixel alive instead
            // (278 registers, one wave per SIMD).
            asm volatile("" : "+v"(acb[8 * rp + i]), "+v"(acr[8 * rp + i]));
        }
        __builtin_amdgcn_sched_barrier(0);   // row pair by row pair: bounds the live registers
    }
    // The three transforms one after the other.
    finish_block_staged<4>(ay, rhi, rlo, stage, coef_dst<false>(g, coef, m0w * 3), gmask + m0w * 3, 3, count);
    __builtin_amdgcn_sched_barrier(0);
    finish_block_staged<4>(acb, rhi + 64, rlo + 64, stage, coef_dst<false>(g, coef, m0w * 3 + 1), gmask + m0w * 3 + 1, 3, count);
    __builtin_amdgcn_sched_barrier(0);
    finish_block_staged<4>(acr, rhi + 128, rlo + 128, stage, coef_dst<false>(g, coef, m0w * 3 + 2), gmask + m0w * 3 + 2, 3, count);
}

// Blocks [firstInMcu, firstInMcu + perMcu) of every MCU, one after the other with their 128
// bytes in one piece, wherever the geometry keeps the two halves of a block (one lane per
// group of eight coefficients).
__global__ __launch_bounds__(256) void k_gather_blocks(KGeom g, const int16_t* __restrict__ coef, int firstInMcu, int perMcu,
                                                       long long blocksOut, uint4* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= blocksOut * 8) return;
    const long long i = idx >> 3;
    const int grp = (int)(idx & 7);
    const long long mcu = i / perMcu;
    const long long src = mcu * g.bpm + firstInMcu + (i - mcu * perMcu);
    const char* from = (const char*)coef + src * g.coefPitch + (grp & 3) * 16 + ((grp & 4) ? g.coefHigh : 0ll);
    out[idx] = *(const uint4*)from;
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

hipError_t launch_gather_blocks(const KGeom& g, const int16_t* d_coef, int firstInMcu, int perMcu, void* d_out, hipStream_t s) {
    const long long blocksOut = g.nMcu * perMcu;
    if (blocksOut <= 0) return hipSuccess;
    const long long grid = (blocksOut * 8 + 255) / 256;
    hipLaunchKernelGGL(k_gather_blocks, dim3((unsigned)grid), dim3(256), 0, s, g, d_coef, firstInMcu, perMcu, blocksOut, (uint4*)d_out);
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
                                                                     int16_t* __restrict__ coef, unsigned char* __restrict__ gmask,
                                                                     u64* __restrict__ clearHist) {
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
    finish_block_staged<2, kFloat>(a, qtab + 192 + 64 * c, qtab + 384 + 64 * c, stage + k * 32 * kStagePitch,
                                   coef_dst(g, coef, mBase * g.bpm + k), gmask + mBase * g.bpm + k, g.bpm, count);
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

hipError_t launch_transform(const KGeom& g, const void* d_rgb, const float* d_qtab, int16_t* d_coef,
                            unsigned char* d_gmask, hipStream_t s, unsigned long long* d_clearHist) {
    if (g.trueDivision) {   // the other kernels divide by multiplying with a two-term reciprocal, exact for integer divisors only
        const long long grid = (g.totalBlocks + kGenThreads - 1) / kGenThreads;
        hipLaunchKernelGGL(k_transform_generic, dim3((unsigned)grid), dim3(kGenThreads), 0, s, g, d_rgb, d_qtab, d_coef,
                           d_gmask, d_clearHist);
        return hipGetLastError();
    }
    const bool bytesAligned = g.sampleFormat == JPEG_HIP_RGB8 && g.stages == JPEG_HIP_STAGE_ALL && g.nGains == 0 &&
                              (g.strideBytes & 3) == 0 && ((uintptr_t)d_rgb & 3) == 0 &&
                              (g.batch == 1 || (g.imageStrideBytes & 3) == 0);
    if (bytesAligned && scheme_is(g, 2, 2, JPEG_HIP_AVERAGE)) {
        const long long grid = (g.nMcu + k420Mcus - 1) / k420Mcus;
        hipLaunchKernelGGL(k_transform_420, dim3((unsigned)grid), dim3(256), 0, s, g, (const unsigned char*)d_rgb, d_qtab,
                           d_coef, d_gmask, d_clearHist);
    } else if (bytesAligned && scheme_is(g, 1, 1, JPEG_HIP_SUBSAMPLING)) {
        const long long grid = (g.nMcu + k444Positions - 1) / k444Positions;
        // One wave per 64 block positions from about a hundred thousand positions on (see
        // k_transform_444_shared); diagnostic switch JPEGENC_444_FORM=3 / 1: always three waves / one.
        static const char* form = getenv("JPEGENC_444_FORM");
        const bool shared = form ? form[0] == '1' : g.nMcu >= 100000;
        if (shared)
            hipLaunchKernelGGL(k_transform_444_shared, dim3((unsigned)grid), dim3(64), 0, s, g, (const unsigned char*)d_rgb,
                               d_qtab, d_coef, d_gmask, d_clearHist);
        else
            hipLaunchKernelGGL(k_transform_444, dim3((unsigned)grid), dim3(3 * k444Positions), 0, s, g,
                               (const unsigned char*)d_rgb, d_qtab, d_coef, d_gmask, d_clearHist);
    } else if (bytesAligned && rows_kernel_takes(g)) {
        const long long grid = (g.nMcu + 63) / 64;
        hipLaunchKernelGGL(k_transform_rows<false>, dim3((unsigned)grid), dim3(64 * g.bpm), 0, s, g,
                           (const unsigned char*)d_rgb, d_qtab, d_coef, d_gmask, d_clearHist);
    } else if (g.sampleFormat == JPEG_HIP_RGBF32 && g.nGains == 0 && g.stages == JPEG_HIP_STAGE_ALL && scheme_is(g, 2, 2, JPEG_HIP_AVERAGE) &&
               (((uintptr_t)d_rgb | (uintptr_t)g.strideBytes | (uintptr_t)g.imageStrideBytes) & 15) == 0) {
        const long long grid = (g.nMcu + k420Mcus - 1) / k420Mcus;
        hipLaunchKernelGGL(k_transform_420f, dim3((unsigned)grid), dim3(256), 0, s, g, (const unsigned char*)d_rgb, d_qtab, d_coef,
                           d_gmask, d_clearHist);
    } else if (g.sampleFormat == JPEG_HIP_RGBF32 && g.nGains == 0 && rows_kernel_takes(g)) {   // any stages, any scheme
        const long long grid = (g.nMcu + 63) / 64;
        hipLaunchKernelGGL(k_transform_rows<true>, dim3((unsigned)grid), dim3(64 * g.bpm), 0, s, g,
                           (const unsigned char*)d_rgb, d_qtab, d_coef, d_gmask, d_clearHist);
    } else {
        const long long grid = (g.totalBlocks + kGenThreads - 1) / kGenThreads;
        hipLaunchKernelGGL(k_transform_generic, dim3((unsigned)grid), dim3(kGenThreads), 0, s, g, d_rgb, d_qtab, d_coef,
                           d_gmask, d_clearHist);
    }
    return hipGetLastError();
}

}  // namespace jpeghip
