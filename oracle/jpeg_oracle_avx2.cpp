// AVX2 + OpenMP restatement of the reference's encode path.  TEST INFRASTRUCTURE ONLY: it is
// the CPU baseline bench.py times beside the GPU and is never linked, loaded or called by the
// product.  It is checked against the scalar restatement in jpeg_oracle.cpp (included below,
// so both share the Huffman builder, the header writer and the C interface): every file in
// tests/golden/oracle_hashes.json has to come out byte for byte the same.
//
// What is vectorised, eight floats per instruction as in the reference's vector path:
//   * pixel normalisation and the colour matrix, SIMD256.h:155-186
//   * the 2:1 reducers, SIMD256.h:260-304
//   * the 8x8 AAN transform with its two transposes, x255 and the table division,
//     SIMD256.h:335-522
// Every product, sum and quotient is rounded on its own and in the order of the scalar
// restatement (no FMA, -ffp-contract=off), so the results are the same bits.
//
// What is arranged differently from the reference, without changing a byte of the output:
//   * colour conversion and the reducers run over pairs of rows through per-thread row
//     buffers instead of whole planes in place;
//   * symbols are not stored: one pass over the coefficients counts them, a second one writes
//     the bits with a 64-bit accumulator;
//   * with restart intervals (an extension the reference does not have) the intervals are
//     written in parallel; without them the bit stream is serial, as in the reference.
#include <immintrin.h>
#include <omp.h>

#include "jpeg_oracle.cpp"

namespace {

// ---------------------------------------------------------------------------------------
// Colour: rgbToYCbCr on eight pixels.
// ---------------------------------------------------------------------------------------
inline __m256 matrixRow(float m0, float m1, float m2, float m3, float bias, __m256 r, __m256 g, __m256 b) {
    __m256 t = _mm256_mul_ps(_mm256_set1_ps(m0), r);
    t = _mm256_add_ps(t, _mm256_mul_ps(_mm256_set1_ps(m1), g));
    t = _mm256_add_ps(t, _mm256_mul_ps(_mm256_set1_ps(m2), b));
    t = _mm256_add_ps(t, _mm256_mul_ps(_mm256_set1_ps(m3), _mm256_set1_ps(1.0f)));
    return _mm256_add_ps(t, _mm256_set1_ps(bias));
}

inline void colour8(__m256 r, __m256 g, __m256 b, float* y, float* cb, float* cr) {
    _mm256_storeu_ps(y, matrixRow(0.2990f, 0.5870f, 0.1140f, 0.0f, 0.0f, r, g, b));
    _mm256_storeu_ps(cb, matrixRow(-0.1687f, -0.3312f, 0.5000f, 0.0f, 0.5f, r, g, b));
    _mm256_storeu_ps(cr, matrixRow(0.5000f, -0.4186f, -0.0813f, 0.0f, 0.5f, r, g, b));
}

// One image row to three rows of simW floats; pixels right of the image repeat the last one
// (Image.cpp:63-139), which gives the same numbers as converting the repeated pixel.
void convertRow(const u8* src, int W, int simW, float* y, float* cb, float* cr) {
    const __m128i pick = _mm_setr_epi8(0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11, -1, -1, -1, -1);
    const __m256 d255 = _mm256_set1_ps(255.0f);
    int x = 0;
    // Two 16-byte loads per eight pixels; the second one reads four bytes beyond them, which
    // the bound keeps inside the row.
    for (; x + 8 <= W - 2; x += 8) {
        const __m128i a = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(src + 3 * x)), pick);
        const __m128i b = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(src + 3 * x + 12)), pick);
        const __m128i rg = _mm_unpacklo_epi32(a, b);   // R0..7 G0..7
        const __m128i bx = _mm_unpackhi_epi32(a, b);   // B0..7
        const __m256 r = _mm256_div_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(rg)), d255);
        const __m256 g = _mm256_div_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_srli_si128(rg, 8))), d255);
        const __m256 bl = _mm256_div_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(bx)), d255);
        colour8(r, g, bl, y + x, cb + x, cr + x);
    }
    for (; x < W; ++x)
        rgbToYCbCr(normalise<u8>(src[3 * x]), normalise<u8>(src[3 * x + 1]), normalise<u8>(src[3 * x + 2]), y[x], cb[x], cr[x]);
    for (; x < simW; ++x) { y[x] = y[W - 1]; cb[x] = cb[W - 1]; cr[x] = cr[W - 1]; }
}

void convertRow(const float* src, int W, int simW, float* y, float* cb, float* cr) {
    const __m256i idx = _mm256_setr_epi32(0, 3, 6, 9, 12, 15, 18, 21);
    int x = 0;
    for (; x + 8 <= W; x += 8) {
        const float* p = src + 3 * x;
        colour8(_mm256_i32gather_ps(p, idx, 4), _mm256_i32gather_ps(p + 1, idx, 4), _mm256_i32gather_ps(p + 2, idx, 4),
                y + x, cb + x, cr + x);
    }
    for (; x < W; ++x) rgbToYCbCr(src[3 * x], src[3 * x + 1], src[3 * x + 2], y[x], cb[x], cr[x]);
    for (; x < simW; ++x) { y[x] = y[W - 1]; cb[x] = cb[W - 1]; cr[x] = cr[W - 1]; }
}

// Width 2:1: mean of neighbours (SIMD256.h:260-277) or the first of each pair.  n outputs,
// n a multiple of 8.  x / 2.0f and x * 0.5f are the same float for every x.
void halveRow(const float* p, float* out, int n, bool average) {
    const __m256 half = _mm256_set1_ps(0.5f);
    for (int i = 0; i < n; i += 8) {
        const __m256 a = _mm256_loadu_ps(p + 2 * i), b = _mm256_loadu_ps(p + 2 * i + 8);
        __m256 v = average ? _mm256_hadd_ps(a, b) : _mm256_shuffle_ps(a, b, 0x88);
        v = _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(v), 0xD8));
        _mm256_storeu_ps(out + i, average ? _mm256_mul_ps(v, half) : v);
    }
}

// Height 2:1 mean, SIMD256.h:295-304.
void meanRows(const float* a, const float* b, float* out, int n) {
    const __m256 half = _mm256_set1_ps(0.5f);
    for (int i = 0; i < n; i += 8)
        _mm256_storeu_ps(out + i, _mm256_mul_ps(_mm256_add_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i)), half));
}

// ---------------------------------------------------------------------------------------
// 8x8 transform.
// ---------------------------------------------------------------------------------------
inline void transpose8(__m256 r[8]) {
    const __m256 t0 = _mm256_unpacklo_ps(r[0], r[1]), t1 = _mm256_unpackhi_ps(r[0], r[1]);
    const __m256 t2 = _mm256_unpacklo_ps(r[2], r[3]), t3 = _mm256_unpackhi_ps(r[2], r[3]);
    const __m256 t4 = _mm256_unpacklo_ps(r[4], r[5]), t5 = _mm256_unpackhi_ps(r[4], r[5]);
    const __m256 t6 = _mm256_unpacklo_ps(r[6], r[7]), t7 = _mm256_unpackhi_ps(r[6], r[7]);
    const __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    const __m256 u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    const __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE);
    const __m256 u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_permute2f128_ps(u0, u4, 0x20);
    r[1] = _mm256_permute2f128_ps(u1, u5, 0x20);
    r[2] = _mm256_permute2f128_ps(u2, u6, 0x20);
    r[3] = _mm256_permute2f128_ps(u3, u7, 0x20);
    r[4] = _mm256_permute2f128_ps(u0, u4, 0x31);
    r[5] = _mm256_permute2f128_ps(u1, u5, 0x31);
    r[6] = _mm256_permute2f128_ps(u2, u6, 0x31);
    r[7] = _mm256_permute2f128_ps(u3, u7, 0x31);
}

// aan1d of the scalar restatement on eight lanes at once (SIMD256.h:337-416).
inline void aan1dv(__m256 v[8]) {
    const AanConst& c = aan();
    const __m256 a1 = _mm256_set1_ps(c.a1), a2 = _mm256_set1_ps(c.a2), a4 = _mm256_set1_ps(c.a4), a5 = _mm256_set1_ps(c.a5);
#define ADD _mm256_add_ps
#define SUB _mm256_sub_ps
#define MUL _mm256_mul_ps
    const __m256 t0 = ADD(v[0], v[7]), t1 = ADD(v[1], v[6]), t2 = ADD(v[2], v[5]), t3 = ADD(v[3], v[4]);
    const __m256 t4 = SUB(v[3], v[4]), t5 = SUB(v[2], v[5]), t6 = SUB(v[1], v[6]), t7 = SUB(v[0], v[7]);

    const __m256 r0 = ADD(t0, t3), r1 = ADD(t1, t2), r2 = SUB(t1, t2), r3 = SUB(t0, t3);
    const __m256 r4 = ADD(t4, t5), r5 = ADD(t5, t6), r6 = ADD(t6, t7), r7 = t7;

    const __m256 u0 = ADD(r0, r1), u1 = SUB(r0, r1);
    __m256 u2 = ADD(r2, r3), u3 = r3, u4 = r4, u5 = r5, u6 = r6, u7 = r7;

    const __m256 z = SUB(u6, u4);
    u2 = MUL(u2, a1);
    u4 = SUB(MUL(u4, a2), MUL(z, a5));
    u5 = MUL(u5, a1);
    u6 = SUB(MUL(u6, a4), MUL(z, a5));

    __m256 p = u2;
    u2 = ADD(u2, u3);
    u3 = SUB(u3, p);
    p = u5;
    u5 = ADD(u5, u7);
    u7 = SUB(u7, p);

    const __m256 y0 = u0, y4 = u1, y2 = u2, y6 = u3;
    const __m256 y5 = ADD(u4, u7), y1 = ADD(u5, u6), y7 = SUB(u5, u6), y3 = SUB(u7, u4);

    v[0] = MUL(y0, _mm256_set1_ps(c.s[0])); v[1] = MUL(y1, _mm256_set1_ps(c.s[1]));
    v[2] = MUL(y2, _mm256_set1_ps(c.s[2])); v[3] = MUL(y3, _mm256_set1_ps(c.s[3]));
    v[4] = MUL(y4, _mm256_set1_ps(c.s[4])); v[5] = MUL(y5, _mm256_set1_ps(c.s[5]));
    v[6] = MUL(y6, _mm256_set1_ps(c.s[6])); v[7] = MUL(y7, _mm256_set1_ps(c.s[7]));
#undef ADD
#undef SUB
#undef MUL
}

// dctQuantBlock of the scalar restatement (SIMD256.h:426-441, 470-522), with the result
// written as 16-bit coefficients in zig-zag order.
inline void dctQuantBlockV(const float* src, size_t stride, const float q[64], int16_t* zz) {
    const __m256 k255 = _mm256_set1_ps(255.0f);
    __m256 r[8];
    for (int i = 0; i < 8; ++i) r[i] = _mm256_mul_ps(_mm256_loadu_ps(src + i * stride), k255);
    transpose8(r);
    aan1dv(r);      // along each row
    transpose8(r);
    aan1dv(r);      // along each column
    r[0] = _mm256_sub_ps(r[0], _mm256_setr_ps(1024.0f, 0, 0, 0, 0, 0, 0, 0));
    alignas(32) int16_t nat[64];
    for (int i = 0; i < 8; i += 2) {
        const __m256 a = _mm256_round_ps(_mm256_div_ps(r[i], _mm256_loadu_ps(q + 8 * i)), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const __m256 b = _mm256_round_ps(_mm256_div_ps(r[i + 1], _mm256_loadu_ps(q + 8 * i + 8)), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        __m256i pk = _mm256_packs_epi32(_mm256_cvttps_epi32(a), _mm256_cvttps_epi32(b));
        pk = _mm256_permute4x64_epi64(pk, 0xD8);
        _mm256_store_si256((__m256i*)(nat + 8 * i), pk);
    }
    for (int i = 0; i < 64; ++i) zz[i] = nat[kZigZag[i]];
}

// ---------------------------------------------------------------------------------------
// Entropy coding straight from the coefficients.
// ---------------------------------------------------------------------------------------
inline uint64_t nonZeroMask(const int16_t* cz) {
    const __m256i zero = _mm256_setzero_si256();
    const __m256i* p = (const __m256i*)cz;
    const __m256i a = _mm256_packs_epi16(_mm256_cmpeq_epi16(_mm256_loadu_si256(p), zero), _mm256_cmpeq_epi16(_mm256_loadu_si256(p + 1), zero));
    const __m256i b = _mm256_packs_epi16(_mm256_cmpeq_epi16(_mm256_loadu_si256(p + 2), zero), _mm256_cmpeq_epi16(_mm256_loadu_si256(p + 3), zero));
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_permute4x64_epi64(a, 0xD8));
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_permute4x64_epi64(b, 0xD8));
    return ~((uint64_t)lo | ((uint64_t)hi << 32));
}

// DC difference of block b of a channel (calculateDCValues, Encoder.cpp:260-289; the
// predictor restarts at every restart interval).
inline int dcDiff(const int16_t* coef, int64_t b, int64_t blocksPerInterval) {
    const short dc = coef[b * 64];
    const short last = (b == 0 || (blocksPerInterval > 0 && b % blocksPerInterval == 0)) ? (short)0 : coef[(b - 1) * 64];
    return (short)(dc - last);
}

inline void countBlock(const int16_t* cz, int diff, uint64_t* dcHist, uint64_t* acHist) {
    dcHist[bitCategory(diff)]++;
    uint64_t m = nonZeroMask(cz) & ~1ull;
    int prev = 0;
    while (m) {                                   // calculateACValues, Encoder.cpp:291-352
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        int zeros = i - prev - 1;
        prev = i;
        while (zeros > 15) { acHist[0xF0]++; zeros -= 16; }
        acHist[(zeros << 4) | bitCategory(cz[i])]++;
    }
    if (prev != 63) acHist[0]++;
}

// Bits into bytes with 0xFF 0x00 stuffing: the byte stream BitSink::push(…, escape = true)
// produces, 32 bits at a time.
struct FastBits {
    std::vector<u8>& buf;
    size_t pos;
    uint64_t acc = 0;
    int n = 0;
    explicit FastBits(std::vector<u8>& b) : buf(b), pos(b.size()) {}
    void room(size_t bytes) { if (buf.size() < pos + bytes) buf.resize(std::max(buf.size() * 2, pos + bytes)); }
    inline void put(uint32_t bits, int len) {
        acc = (acc << len) | bits;
        n += len;
        if (n >= 32) {
            const uint32_t w = (uint32_t)(acc >> (n - 32));
            n -= 32;
            u8* o = buf.data() + pos;
            if (((w & ((w & 0x7f7f7f7fu) + 0x01010101u)) & 0x80808080u) == 0) {     // no byte is 0xFF
                const uint32_t be = __builtin_bswap32(w);
                memcpy(o, &be, 4);
                pos += 4;
            } else {
                for (int s = 24; s >= 0; s -= 8) {
                    const u8 v = (u8)(w >> s);
                    *o++ = v;
                    if (v == 0xff) *o++ = 0;
                }
                pos = (size_t)(o - buf.data());
            }
        }
    }
    void bytes(const u8* p, size_t k) { room(k); memcpy(buf.data() + pos, p, k); pos += k; }   // byte aligned only
    void finish(bool padOnes) {                     // fillToByte
        room(16);
        while (n >= 8) {
            const u8 v = (u8)(acc >> (n - 8));
            n -= 8;
            buf[pos++] = v;
            if (v == 0xff) buf[pos++] = 0;
        }
        if (n > 0) {
            u8 v = (u8)((acc << (8 - n)) & 0xff);
            if (padOnes) v |= (u8)(0xff >> n);
            buf[pos++] = v;
            if (v == 0xff) buf[pos++] = 0;
            n = 0;
        }
        acc = 0;
    }
};

inline uint32_t amplitude(int v, int cat) {          // Encoder.cpp:273-285, 321-333
    const unsigned pat = v < 0 ? ~(unsigned)(-v) : (unsigned)v;
    return pat & ((1u << cat) - 1);
}

inline void emitBlock(FastBits& w, const int16_t* cz, int diff, const HuffTable& dc, const HuffTable& ac) {
    int cat = bitCategory(diff);
    w.put(((uint32_t)dc.code[cat] << cat) | amplitude(diff, cat), dc.len[cat] + cat);
    uint64_t m = nonZeroMask(cz) & ~1ull;
    int prev = 0;
    while (m) {
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        int zeros = i - prev - 1;
        prev = i;
        while (zeros > 15) { w.put(ac.code[0xF0], ac.len[0xF0]); zeros -= 16; }
        const int v = cz[i];
        cat = bitCategory(v);
        const int s = (zeros << 4) | cat;
        w.put(((uint32_t)ac.code[s] << cat) | amplitude(v, cat), ac.len[s] + cat);
    }
    if (prev != 63) w.put(ac.code[0], ac.len[0]);
}

bool vectorSchemeOk(const Scheme& sc) {
    for (int c = 0; c < 3; ++c)
        if ((sc.wf[c] != 1 && sc.wf[c] != 2) || (sc.hf[c] != 1 && sc.hf[c] != 2)) return false;
    return true;
}

template <typename Sample>
void encodeImageAvx2(const Sample* rgb, int W, int H, size_t strideSamples, const Scheme& sc, const float qt[3][64],
                     int restartInterval, bool padOnes, const uint64_t* histOverride, int rstBase, int headerH,
                     EncodeResult& res) {
    // Sampling factors other than 1 and 2, and channel gains, take the scalar path.
    if (!vectorSchemeOk(sc) || !g_gains.empty() || W < 1 || H < 1) {
        encodeImage<Sample>(rgb, W, H, strideSamples, sc, qt, restartInterval, padOnes, histOverride, rstBase, headerH, res);
        return;
    }
    const int simW = W + (W % sc.stepW == 0 ? 0 : sc.stepW - W % sc.stepW);
    const int simH = H + (H % sc.stepH == 0 ? 0 : sc.stepH - H % sc.stepH);
    res.simW = simW; res.simH = simH;
    StageClock clk;
    int cw[3], chh[3];
    Plane ch[3];
    for (int c = 0; c < 3; ++c) {
        cw[c] = simW / sc.wf[c];
        chh[c] = simH / sc.hf[c];
        res.chW[c] = cw[c]; res.chH[c] = chh[c];
        ch[c].resize((size_t)cw[c] * chh[c]);
    }

    // Planes, colour and the reducers, two image rows at a time (simH is a multiple of 8).
#pragma omp parallel
    {
        std::vector<float> full((size_t)3 * simW), half((size_t)3 * 2 * simW);
#pragma omp for schedule(static)
        for (int yp = 0; yp < simH / 2; ++yp) {
            const float* rows[3][2];
            for (int r = 0; r < 2; ++r) {
                const int y = 2 * yp + r;
                const int sy = y < H ? y : H - 1;
                float* f[3] = {full.data(), full.data() + simW, full.data() + 2 * (size_t)simW};
                float* dst[3];
                // A channel that keeps its width and height is written straight into its plane.
                for (int c = 0; c < 3; ++c)
                    dst[c] = (sc.wf[c] == 1 && sc.hf[c] == 1) ? ch[c].data() + (size_t)y * simW : f[c];
                convertRow(rgb + (size_t)sy * strideSamples, W, simW, dst[0], dst[1], dst[2]);
                for (int c = 0; c < 3; ++c) {
                    if (sc.wf[c] == 1 && sc.hf[c] == 1) continue;
                    float* h = half.data() + ((size_t)c * 2 + r) * simW;
                    if (sc.wf[c] == 2) halveRow(dst[c], h, cw[c], sc.wm[c] == 1);
                    else memcpy(h, dst[c], (size_t)simW * sizeof(float));
                    rows[c][r] = h;
                }
            }
            for (int c = 0; c < 3; ++c) {
                if (sc.wf[c] == 1 && sc.hf[c] == 1) continue;
                float* p = ch[c].data();
                const size_t w = (size_t)cw[c];
                if (sc.hf[c] == 1) {
                    memcpy(p + (size_t)(2 * yp) * w, rows[c][0], w * sizeof(float));
                    memcpy(p + (size_t)(2 * yp + 1) * w, rows[c][1], w * sizeof(float));
                } else if (sc.hm[c] == 1) {
                    meanRows(rows[c][0], rows[c][1], p + (size_t)yp * w, cw[c]);
                } else {
                    memcpy(p + (size_t)yp * w, rows[c][0], w * sizeof(float));
                }
            }
        }
    }
    clk.lap("planes + colour + reduce");

    // Transform and quantization, blocks in MCU order (createBlocks, Encoder.cpp:168-227).
    const int mcusPerRow = simW / sc.stepW, mcuRows = simH / sc.stepH;
    const int64_t nMcu = (int64_t)mcusPerRow * mcuRows;
    for (int c = 0; c < 3; ++c) {
        const int bw = sc.invW[c], bh = sc.invH[c];
        res.coef[c].resize((size_t)nMcu * bw * bh * 64);
        int16_t* out = res.coef[c].data();
        const float* p = ch[c].data();
        const size_t width = (size_t)cw[c];
#pragma omp parallel for schedule(static)
        for (int my = 0; my < mcuRows; ++my)
            for (int mx = 0; mx < mcusPerRow; ++mx)
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        const int64_t b = (((int64_t)my * mcusPerRow + mx) * bh + by) * bw + bx;
                        dctQuantBlockV(p + ((size_t)my * bh + by) * 8 * width + ((size_t)mx * bw + bx) * 8, width, qt[c], out + b * 64);
                    }
    }
    clk.lap("dct");

    // Histograms (Encoder::createHuffmanTable, Encoder.cpp:368-416), per thread and summed.
    memset(res.hist, 0, sizeof res.hist);
    int64_t perMcu[3], perInterval[3];
    for (int c = 0; c < 3; ++c) {
        perMcu[c] = (int64_t)sc.invW[c] * sc.invH[c];
        perInterval[c] = restartInterval > 0 ? perMcu[c] * restartInterval : 0;
    }
#pragma omp parallel
    {
        uint64_t local[4][256];
        memset(local, 0, sizeof local);
        for (int c = 0; c < 3; ++c) {
            const int t = c == 0 ? 0 : 1;
            const int16_t* coef = res.coef[c].data();
            const int64_t nb = nMcu * perMcu[c];
#pragma omp for schedule(static) nowait
            for (int64_t b = 0; b < nb; ++b) countBlock(coef + b * 64, dcDiff(coef, b, perInterval[c]), local[t], local[2 + t]);
        }
#pragma omp critical
        for (int t = 0; t < 4; ++t)
            for (int s = 0; s < 256; ++s) res.hist[t][s] += local[t][s];
    }
    chooseTables(histOverride, res);
    clk.lap("histograms + tables");

    std::vector<u8> hdr;
    headerBytes(W, H, headerH, sc, qt, res.tables, restartInterval, hdr);
    res.scanOffset = hdr.size();

    // The entropy-coded segment (serializeScanData / pushBlock, Encoder.cpp:490-536).  Every
    // thread writes a run of whole restart intervals into a buffer of its own.
    const int64_t intervals = restartInterval > 0 ? (nMcu + restartInterval - 1) / restartInterval : 1;
    const int threads = (int)std::min<int64_t>(intervals, omp_get_max_threads());
    std::vector<std::vector<u8>> part(threads);
    size_t maxBlockBytes = 0;                      // the most one MCU can take, stuffing included
    for (int c = 0; c < 3; ++c) maxBlockBytes += (size_t)perMcu[c] * (64 * 27 / 8 + 8) * 2;
#pragma omp parallel num_threads(threads)
    {
        const int tid = omp_get_thread_num();
        const int nth = omp_get_num_threads();
        const int64_t i0 = intervals * tid / nth, i1 = intervals * (tid + 1) / nth;
        std::vector<u8>& buf = part[tid];
        buf.reserve((size_t)((i1 - i0) * (restartInterval > 0 ? restartInterval : nMcu)) * 48 + 4096);
        FastBits w(buf);
        buf.resize(buf.capacity());
        for (int64_t iv = i0; iv < i1; ++iv) {
            if (iv > 0) {
                const u8 rst[2] = {0xff, (u8)(0xd0 + ((rstBase + iv - 1) & 7))};
                w.bytes(rst, 2);
            }
            const int64_t m0 = restartInterval > 0 ? iv * restartInterval : 0;
            const int64_t m1 = restartInterval > 0 ? std::min<int64_t>(nMcu, m0 + restartInterval) : nMcu;
            for (int64_t m = m0; m < m1; ++m) {
                w.room(maxBlockBytes);
                for (int c = 0; c < 3; ++c) {
                    const int t = c == 0 ? 0 : 1;
                    const int16_t* coef = res.coef[c].data();
                    for (int64_t k = 0; k < perMcu[c]; ++k) {
                        const int64_t b = m * perMcu[c] + k;
                        emitBlock(w, coef + b * 64, dcDiff(coef, b, perInterval[c]), res.tables[t], res.tables[2 + t]);
                    }
                }
            }
            w.finish(padOnes);
        }
        buf.resize(w.pos);
    }
    size_t scan = 0;
    std::vector<size_t> at(threads);
    for (int t = 0; t < threads; ++t) { at[t] = scan; scan += part[t].size(); }
    res.scanLen = scan;
    res.jpeg.resize(hdr.size() + scan + 2);
    memcpy(res.jpeg.data(), hdr.data(), hdr.size());
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int t = 0; t < threads; ++t) memcpy(res.jpeg.data() + hdr.size() + at[t], part[t].data(), part[t].size());
    res.jpeg[hdr.size() + scan] = 0xff;
    res.jpeg[hdr.size() + scan + 1] = 0xd9;
    clk.lap("bit stream");
}

}  // namespace

extern "C" {

// Same arguments and result handle as orc_encode.
void* orcv_encode(const void* rgb, int sampleType, int W, int H, int64_t strideSamples, const int32_t* scheme,
                  const float* qt, int restartInterval, int padOnes, const uint64_t* histOverride, int rstBase,
                  int headerH) {
    OrcResult* o = new OrcResult();
    Scheme sc(scheme);
    float q[3][64];
    memcpy(q, qt, sizeof q);
    if (sampleType == 0)
        encodeImageAvx2<u8>((const u8*)rgb, W, H, (size_t)strideSamples, sc, q, restartInterval, padOnes != 0,
                            histOverride, rstBase, headerH, o->r);
    else
        encodeImageAvx2<float>((const float*)rgb, W, H, (size_t)strideSamples, sc, q, restartInterval, padOnes != 0,
                               histOverride, rstBase, headerH, o->r);
    return o;
}

int orcv_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
