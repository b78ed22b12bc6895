// TEST INFRASTRUCTURE ONLY -- CPU oracle for the JPEG encode hot path.
//
// This file is a from-scratch scalar restatement of the algorithm the reference
// encoder (Irame/JPEG-Encoder, tree "JPEG-Encoder-SIMD/") runs between "interleaved
// RGB in memory" and "JFIF bytes in memory".  It exists so that tests, the smoke check
// and the benchmark's cpu_baseline leg have something to compare the HIP path with.
// Nothing in the shipped library links, loads or calls it.
//
// Pin status: the reference tree cannot be compiled in this image (its translation
// units include the Windows SDK headers <tchar.h>, <SDKDDKVer.h> and "intrin.h"), so
// this restatement is pinned by the known-answer vectors of the reference's own
// ad-hoc tests (testDCT, testHuffmanEncoding, colour conversion) that are stored
// under tests/golden/reference_kats.json, not by a reference binary.
//
// Every function names the reference file:line it follows.  Floating point is
// evaluated exactly as the reference's 256-bit AVX path does: every multiply and add
// is rounded separately (build with -ffp-contract=off, no -mfma), true IEEE division,
// round-half-to-even.  Extensions that the reference does not have (restart
// intervals, selectable pad bit) are marked "extension".
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

typedef uint8_t u8;

// ------------------------------------------------------------------------------------
// Constants.  const_math.h:16-41 builds cosines with a triple-angle recursion in
// double (tolerance 1e-3); SIMD256.h:335-336 turns them into the AAN factors in float.
// The recursion is restated here and the results are compared with the hex floats the
// compiled reference holds (tests/test_oracle_kats.py, through orc_constants()).
// ------------------------------------------------------------------------------------
const double kTol = 0.001;                       // const_math.h:16
const double kPi = 3.14159265358979323846;       // M_PI
const double kPi2 = 1.57079632679489661923;      // M_PI_2
const float kPif = (float)kPi;                   // stdafx.h:24
const float kSqrt1_2f = (float)0.70710678118654752440;  // stdafx.h:27

double sinHelper(double x) {                     // const_math.h:31-34
    if (x < kTol) return x;
    double s = sinHelper(x / 3.0);
    return 3 * s - 4 * (s * s * s);
}
double cSin(double x) { return sinHelper(x < 0 ? -x + kPi : x); }   // const_math.h:36-39
double cCos(double x) { return cSin(kPi2 - x); }                     // const_math.h:41

float Cf(size_t k) { return k == 0 ? 1.0f : (float)cCos((double)((float)k * kPif / 16)); }  // SIMD256.h:335
float Sf(size_t k) { return k == 0 ? kSqrt1_2f / 2.0f : 1.0f / (4.0f * Cf(k)); }            // SIMD256.h:336

struct AanConst {
    float a1, a2, a4, a5, s[8];
    AanConst() {
        a1 = Cf(4);                // SIMD256.h:348
        a2 = Cf(2) - Cf(6);        // SIMD256.h:349
        a4 = Cf(6) + Cf(2);        // SIMD256.h:351
        a5 = Cf(6);                // SIMD256.h:352
        for (int k = 0; k < 8; ++k) s[k] = Sf(k);
    }
};
const AanConst& aan() { static AanConst c; return c; }

// ZigZag.h:10-19
const int kZigZag[64] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------
// Colour conversion, SIMD256.h:155-186.  One pixel; each product and sum rounded on
// its own, left to right, including the "+ 0*1" term of the 4x4 matrix and the bias.
// ------------------------------------------------------------------------------------
inline void rgbToYCbCr(float r, float g, float b, float& y, float& cb, float& cr) {
    static const float M[3][4] = {{0.2990f, 0.5870f, 0.1140f, 0.0f},
                                  {-0.1687f, -0.3312f, 0.5000f, 0.0f},
                                  {0.5000f, -0.4186f, -0.0813f, 0.0f}};
    static const float bias[3] = {0.0f, 0.5f, 0.5f};
    float out[3];
    for (int i = 0; i < 3; ++i) {
        float t = M[i][0] * r;
        t = t + M[i][1] * g;
        t = t + M[i][2] * b;
        t = t + M[i][3] * 1.0f;
        t = t + bias[i];
        out[i] = t;
    }
    y = out[0]; cb = out[1]; cr = out[2];
}

// ------------------------------------------------------------------------------------
// One-dimensional 8-point AAN transform, SIMD256.h:337-416, on one lane.
// ------------------------------------------------------------------------------------
inline void aan1d(float* v0, float* v1, float* v2, float* v3, float* v4, float* v5, float* v6, float* v7) {
    const AanConst& c = aan();
    float t0 = *v0 + *v7, t1 = *v1 + *v6, t2 = *v2 + *v5, t3 = *v3 + *v4;
    float t4 = *v3 - *v4, t5 = *v2 - *v5, t6 = *v1 - *v6, t7 = *v0 - *v7;

    float r0 = t0 + t3, r1 = t1 + t2, r2 = t1 - t2, r3 = t0 - t3;
    float r4 = t4 + t5, r5 = t5 + t6, r6 = t6 + t7, r7 = t7;

    float u0 = r0 + r1, u1 = r0 - r1, u2 = r2 + r3, u3 = r3;
    float u4 = r4, u5 = r5, u6 = r6, u7 = r7;

    float z = u6 - u4;
    u2 = u2 * c.a1;
    u4 = (u4 * c.a2) - (z * c.a5);
    u5 = u5 * c.a1;
    u6 = (u6 * c.a4) - (z * c.a5);

    float p = u2;
    u2 = u2 + u3;
    u3 = u3 - p;
    p = u5;
    u5 = u5 + u7;
    u7 = u7 - p;

    float y0 = u0, y4 = u1, y2 = u2, y6 = u3;
    float y5 = u4 + u7, y1 = u5 + u6, y7 = u5 - u6, y3 = u7 - u4;

    *v0 = y0 * c.s[0]; *v1 = y1 * c.s[1]; *v2 = y2 * c.s[2]; *v3 = y3 * c.s[3];
    *v4 = y4 * c.s[4]; *v5 = y5 * c.s[5]; *v6 = y6 * c.s[6]; *v7 = y7 * c.s[7];
}

// SIMD256.h:426-441: transpose, 1-D, transpose, 1-D, then DC -= 1024.  The first pass
// therefore runs along each row, the second along each column.
inline void aan2d(float b[64]) {
    for (int r = 0; r < 8; ++r) {
        float* p = b + 8 * r;
        aan1d(p, p + 1, p + 2, p + 3, p + 4, p + 5, p + 6, p + 7);
    }
    for (int c = 0; c < 8; ++c) {
        float* p = b + c;
        aan1d(p, p + 8, p + 16, p + 24, p + 32, p + 40, p + 48, p + 56);
    }
    b[0] = b[0] - 1024.0f;
}

// SIMD256.h:470-522: x*255, 2-D transform, divide by the table (natural order, IEEE
// division), round to nearest even.  `src` addresses an 8x8 block with row stride.
inline void dctQuantBlock(float* src, size_t stride, const float q[64]) {
    float b[64];
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) b[8 * r + c] = src[r * stride + c] * 255.0f;
    aan2d(b);
    for (int i = 0; i < 64; ++i) b[i] = nearbyintf(b[i] / q[i]);
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) src[r * stride + c] = b[8 * r + c];
}

// ------------------------------------------------------------------------------------
// Sampling scheme, SamplingScheme.h:25-44.
// ------------------------------------------------------------------------------------
int gcdi(int a, int b) { return b == 0 ? a : gcdi(b, a % b); }
int lcmi(int a, int b) { return (a * b) / gcdi(a, b); }

struct Scheme {
    int wf[3], wm[3], hf[3], hm[3];  // factor and method (0 = Subsampling, 1 = Average)
    int stepW, stepH;                // SamplingScheme.h:36-38
    int invW[3], invH[3];            // SamplingScheme.h:39-42 (JPEG H and V factors)
    explicit Scheme(const int32_t* s) {
        for (int c = 0; c < 3; ++c) { wf[c] = s[4 * c]; wm[c] = s[4 * c + 1]; hf[c] = s[4 * c + 2]; hm[c] = s[4 * c + 3]; }
        stepW = lcmi(lcmi(wf[0], wf[1]), wf[2]) * 8;
        stepH = lcmi(lcmi(hf[0], hf[1]), hf[2]) * 8;
        for (int c = 0; c < 3; ++c) { invW[c] = stepW / wf[c] / 8; invH[c] = stepH / hf[c] / 8; }
    }
};

// ------------------------------------------------------------------------------------
// Bit buffer, BitBuffer.cpp:128-204 (pushBits with the escape flag) restated byte for
// byte, plus fillToByteBorder (BitBuffer.cpp:12-16).  Source bits are MSB-aligned in
// `src`, exactly as the reference hands over BEushort patterns and Huffman codes.
// ------------------------------------------------------------------------------------
struct BitSink {
    std::vector<u8> data;
    size_t bitPos;
    BitSink() : data(1, 0), bitPos(0) {}
    void reserveBits(size_t n) {                 // BitBuffer.cpp:245-254
        size_t need = (n + bitPos + 7) / 8;
        if (need > data.size()) {
            size_t sz = data.size();
            while (need > sz) sz *= 2;
            data.resize(sz, 0);
        }
    }
    static u8 join(u8 left, u8 right, size_t leftCount) {   // BitBuffer.cpp:257-268
        size_t rightCount = 8 - leftCount;
        left = (u8)(left << rightCount);
        right = (u8)(right >> leftCount);
        u8 lm = (u8)(0xff << rightCount), rm = (u8)(0xff >> leftCount);
        return (u8)((left & lm) | (right & rm));
    }
    // `src` must be readable one byte past the last byte that holds payload bits
    // (the reference reads it too and masks it away).
    void push(size_t nbits, const u8* src, bool escape) {
        if (nbits == 0) return;
        reserveBits(nbits + 16);
        size_t freeBits = 8 - bitPos % 8;
        size_t byteOff = bitPos / 8;
        if (nbits <= freeBits) {
            data[byteOff] |= (u8)((src[0] >> (8 - freeBits)) & (u8)(0xff << (freeBits - nbits)));
            bitPos += nbits;
            if (escape && data[byteOff] == 0xff) {
                reserveBits(nbits + 8);
                data[byteOff + 1] = 0x00;
                bitPos += 8;
            }
            return;
        }
        data[byteOff++] |= (u8)(src[0] >> (8 - freeBits));
        bitPos += freeBits;
        if (escape && data[byteOff - 1] == 0xff) {
            reserveBits(nbits + 8);
            data[byteOff++] = 0x00;
            bitPos += 8;
        }
        size_t srcOff = freeBits, srcByte = 0, leftCount = 8 - freeBits;
        while (srcOff < nbits) {
            reserveBits(nbits + 24);
            data[byteOff++] = join(src[srcByte], src[srcByte + 1], leftCount);
            srcByte++;
            if (escape && (nbits - srcOff >= 8) && data[byteOff - 1] == 0xff) {
                reserveBits(nbits - srcOff + 8);
                data[byteOff++] = 0x00;
                bitPos += 8;
            }
            bitPos += 8;
            srcOff += 8;
        }
        data[byteOff - 1] &= (u8)(0xff << (srcOff - nbits));
        bitPos -= (srcOff - nbits);
    }
    void pushBytes(const u8* p, size_t n) {       // header segments: unescaped, byte aligned
        std::vector<u8> tmp(p, p + n);
        tmp.push_back(0);
        push(n * 8, tmp.data(), false);
    }
    // BitBuffer.cpp:12-16 pads with zero bits.  padOnes is an extension (the JPEG
    // standard asks for one bits); a completed 0xFF byte is then stuffed as usual.
    void fillToByte(bool padOnes) {
        size_t rem = bitPos % 8;
        if (rem == 0) return;
        if (!padOnes) { bitPos = ((bitPos + 7) / 8) * 8; reserveBits(8); return; }
        u8 ones[2] = {0xff, 0x00};
        push(8 - rem, ones, true);
    }
    size_t byteLen() const { return (bitPos + 7) / 8; }
};

// ------------------------------------------------------------------------------------
// Length-limited Huffman table, HuffmanCodingByte.h:102-210 (package-merge with a
// zero-frequency dummy leaf; ties are resolved by libstdc++'s std::sort, which is not
// stable, so the same call sequence on the same element order is kept).
// ------------------------------------------------------------------------------------
struct PmNode {
    uint64_t freq;
    int left, right;   // indices into the node pool, -1 for leaves
    int leaf;          // index into the leaf array for leaves, -1 otherwise
};

struct HuffTable {
    u8 len[256];
    uint16_t code[256];       // right-aligned code bits
    uint64_t wide[256];       // the same in full (limits above 16 bits)
    bool present[256];
    int symbolCount;
    HuffTable() : symbolCount(0) { memset(len, 0, sizeof len); memset(code, 0, sizeof code); memset(wide, 0, sizeof wide); memset(present, 0, sizeof present); }
};

void incLeafLengths(const std::vector<PmNode>& pool, int idx, std::vector<uint32_t>& leafLen) {
    const PmNode& n = pool[idx];      // HuffmanCoding.h:315-330
    if (n.leaf >= 0) { leafLen[n.leaf]++; return; }
    incLeafLengths(pool, n.left, leafLen);
    incLeafLengths(pool, n.right, leafLen);
}

bool buildHuffman(const uint64_t counts[256], HuffTable& out, size_t maxLen = 16) {
    out = HuffTable();
    std::vector<PmNode> pool;
    std::vector<int> orig;                 // node indices of the leaves, ascending symbol
    std::vector<int> leafSymbol;
    uint64_t total = 0;
    for (int s = 0; s < 256; ++s) total += counts[s];
    if (total == 0) return true;           // HuffmanCodingByte.h:110: empty table
    for (int s = 0; s < 256; ++s) {        // HuffmanCodingByte.h:124-132
        if (counts[s] != 0) {
            out.symbolCount++;
            PmNode n = {counts[s], -1, -1, (int)leafSymbol.size()};
            leafSymbol.push_back(s);
            orig.push_back((int)pool.size());
            pool.push_back(n);
        }
    }
    {                                      // HuffmanCodingByte.h:135: dummy leaf (symbol 0, frequency 0)
        PmNode n = {0, -1, -1, (int)leafSymbol.size()};
        leafSymbol.push_back(0);
        orig.push_back((int)pool.size());
        pool.push_back(n);
    }
    auto byFreq = [&pool](int a, int b) { return pool[a].freq < pool[b].freq; };
    std::vector<int> cur = orig;
    for (size_t l = 1; l < maxLen; ++l) {  // HuffmanCodingByte.h:139-156
        std::sort(cur.begin(), cur.end(), byFreq);
        std::vector<int> next = orig;
        for (size_t i = 1; i < cur.size(); i += 2) {
            PmNode n = {pool[cur[i - 1]].freq + pool[cur[i]].freq, cur[i - 1], cur[i], -1};
            next.push_back((int)pool.size());
            pool.push_back(n);
        }
        cur = next;
    }
    std::sort(cur.begin(), cur.end(), byFreq);   // HuffmanCodingByte.h:159-162
    std::vector<uint32_t> leafLen(leafSymbol.size(), 0);
    for (size_t i = 0; i < 2 * orig.size() - 2; ++i) incLeafLengths(pool, cur[i], leafLen);   // :165-168

    std::vector<int> order(leafSymbol.size());   // HuffmanCodingByte.h:171-178
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&leafLen](int a, int b) { return leafLen[a] > leafLen[b]; });

    size_t lastLen = leafLen[order[0]];          // HuffmanCodingByte.h:181-207
    uint64_t code = ~uint64_t(0) >> (64 - lastLen);
    for (size_t i = 0; i < order.size(); ++i) {
        int leaf = order[i];
        if (pool[orig[leaf]].freq == 0) continue;
        size_t curLen = leafLen[leaf];
        if (curLen != lastLen) code >>= lastLen - curLen;
        code--;
        int sym = leafSymbol[leaf];
        out.len[sym] = (u8)curLen;
        out.code[sym] = (uint16_t)code;
        out.wide[sym] = code;
        out.present[sym] = true;
        lastLen = curLen;
    }
    return true;
}

// DHT segment, JPEGSegments.h:102-140 and :203-207: marker, length, info, 16 counts,
// then the symbols ordered by (code length, code value).
void dhtBytes(const HuffTable& t, int htNum, int htType, std::vector<u8>& out) {
    std::vector<int> syms;
    for (int s = 0; s < 256; ++s) if (t.present[s]) syms.push_back(s);
    std::sort(syms.begin(), syms.end(), [&t](int a, int b) {
        if (t.len[a] == t.len[b]) return t.code[a] < t.code[b];
        return t.len[a] < t.len[b];
    });
    unsigned length = 2 + 1 + 16 + (unsigned)syms.size();
    out.push_back(0xff); out.push_back(0xc4);
    out.push_back((u8)(length >> 8)); out.push_back((u8)length);
    out.push_back((u8)((htNum & 15) | (htType << 4)));
    u8 cnt[16] = {0};
    for (int s : syms) cnt[t.len[s] - 1]++;
    out.insert(out.end(), cnt, cnt + 16);
    for (int s : syms) out.push_back((u8)s);
}

// Encoder.cpp:354-366
inline int bitCategory(int v) {
    if (v == 0) return 0;
    return 32 - __builtin_clz((unsigned)abs(v));
}

struct BlockSymbols {
    u8 dcCat;
    uint16_t dcBits;                   // left-aligned pattern (Encoder.cpp:283)
    std::vector<u8> acSym;
    std::vector<uint16_t> acBits;      // left-aligned patterns
};

// Fixed Huffman tables for the next encodes (extension; see encodeImage).
bool g_fixedTables = false;
// Encoder::multiplyColorChannelBy calls made after convertToYCbCr (Encoder.cpp:35-38): the
// plane of `channel` is multiplied by `value`, before (stage 0) or after (stage 1)
// reduceResolutionBySchema.
struct GainOp { int channel; int stage; float value; };
std::vector<GainOp> g_gains;
u8 g_fixedLen[4][256];
uint16_t g_fixedCode[4][256];

// An array without the zero fill of std::vector (a serial pass over every byte that also
// takes the page faults away from the parallel loops).
template <typename T>
struct RawArray {
    T* p = nullptr;
    size_t n = 0;
    RawArray() = default;
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    ~RawArray() { free(p); }
    void resize(size_t count) {
        free(p);
        p = (T*)malloc((count ? count : 1) * sizeof(T));
        if (!p) abort();
        n = count;
    }
    T& operator[](size_t i) { return p[i]; }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
};
typedef RawArray<float> Plane;   // every sample is written by the de-interleave

struct EncodeResult {
    std::vector<u8> jpeg;
    RawArray<int16_t> coef[3];         // per channel, blocks in MCU order, zig-zag inside a block
    uint64_t hist[4][256];             // DC-Y, DC-C, AC-Y, AC-C
    HuffTable tables[4];
    size_t scanOffset, scanLen;        // entropy-coded segment inside jpeg
    int simW, simH;
    int chW[3], chH[3];
};

// Encoder::createHuffmanTable, Encoder.cpp:368-416: four tables from the four histograms.
// Extension (sharded images): tables may come from histograms summed over all bands.
// Extension (fixed tables, e.g. the standard ones of ITU-T T.81 Annex K.3): no table
// optimisation at all; a symbol without a code is a caller error.
void chooseTables(const uint64_t* histOverride, EncodeResult& res) {
    for (int t = 0; t < 4; ++t) {
        if (g_fixedTables) {
            HuffTable& ht = res.tables[t];
            for (int s = 0; s < 256; ++s) {
                ht.len[s] = g_fixedLen[t][s];
                ht.code[s] = g_fixedCode[t][s];
                ht.present[s] = g_fixedLen[t][s] != 0;
            }
        } else {
            buildHuffman(histOverride ? histOverride + 256 * t : res.hist[t], res.tables[t]);
        }
    }
}

// Everything in front of the entropy-coded segment, in the order of Encoder::serialize,
// Encoder.cpp:439-488.
void headerBytes(int W, int H, int headerH, const Scheme& sc, const float qt[3][64], const HuffTable tables[4],
                 int restartInterval, std::vector<u8>& hdr) {
    const u8 soiApp0[] = {0xff, 0xd8, 0xff, 0xe0, 0x00, 0x10, 0x4a, 0x46, 0x49, 0x46, 0x00, 0x01, 0x01,
                          0x00, 0x00, 0x48, 0x00, 0x48, 0x00, 0x00};          // JPEGSegments.h:50-75
    hdr.insert(hdr.end(), soiApp0, soiApp0 + sizeof soiApp0);
    {                                                                           // JPEGSegments.h:77-100
        const int fh = headerH > 0 ? headerH : H;   // extension: a band reports the whole image's height
        const u8 sof[] = {0xff, 0xc0, 0x00, 0x11, 0x08, (u8)(fh >> 8), (u8)fh, (u8)(W >> 8), (u8)W, 0x03};
        hdr.insert(hdr.end(), sof, sof + sizeof sof);
        for (int c = 0; c < 3; ++c) {
            hdr.push_back((u8)(c + 1));
            hdr.push_back((u8)((sc.invW[c] << 4) | (0xf & sc.invH[c])));
            hdr.push_back((u8)c);
        }
    }
    dhtBytes(tables[0], 0, 0, hdr);    // Encoder.cpp:455-458, 476-479
    dhtBytes(tables[1], 1, 0, hdr);
    dhtBytes(tables[2], 0, 1, hdr);
    dhtBytes(tables[3], 1, 1, hdr);
    for (int c = 0; c < 3; ++c) {          // JPEGSegments.h:152-167
        hdr.push_back(0xff); hdr.push_back(0xdb); hdr.push_back(0x00); hdr.push_back(0x43);
        hdr.push_back((u8)c);
        for (int i = 0; i < 64; ++i) hdr.push_back((u8)qt[c][kZigZag[i]]);
    }
    if (restartInterval > 0) {             // extension: DRI segment
        const u8 dri[] = {0xff, 0xdd, 0x00, 0x04, (u8)(restartInterval >> 8), (u8)restartInterval};
        hdr.insert(hdr.end(), dri, dri + sizeof dri);
    }
    const u8 sos[] = {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00};  // JPEGSegments.h:175-190
    hdr.insert(hdr.end(), sos, sos + sizeof sos);
}

// ------------------------------------------------------------------------------------
// The encode path, JPEG-Encoder-SIMD.cpp:397-441.
// ------------------------------------------------------------------------------------
template <typename Sample>
float normalise(Sample s);
template <> float normalise<u8>(u8 s) { return s / 255.0f; }      // ImageLoader.cpp:190-193
template <> float normalise<float>(float s) { return s; }        // Image::setRawPixelData takes floats

// ORC_TIMING=1 in the environment prints the seconds spent per stage to stderr.
struct StageClock {
    bool on = getenv("ORC_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};

template <typename Sample>
void encodeImage(const Sample* rgb, int W, int H, size_t strideSamples, const Scheme& sc, const float qt[3][64],
                 int restartInterval, bool padOnes, const uint64_t* histOverride, int rstBase, int headerH,
                 EncodeResult& res) {
    // Image::Image, Image.cpp:6-16: padded ("simulated") size.
    const int simW = W + (W % sc.stepW == 0 ? 0 : sc.stepW - W % sc.stepW);
    const int simH = H + (H % sc.stepH == 0 ? 0 : sc.stepH - H % sc.stepH);
    res.simW = simW; res.simH = simH;
    const size_t plane = (size_t)simW * simH;
    StageClock clk;
    Plane ch[3];   // every sample is written by the de-interleave below
    for (int c = 0; c < 3; ++c) ch[c].resize(plane);

    // Image::setRawPixelData, Image.cpp:63-139: de-interleave; pixels outside the real
    // image repeat the nearest edge pixel.
#pragma omp parallel for
    for (int y = 0; y < simH; ++y) {
        int sy = y < H ? y : H - 1;
        for (int x = 0; x < simW; ++x) {
            int sx = x < W ? x : W - 1;
            const Sample* p = rgb + (size_t)sy * strideSamples + (size_t)sx * 3;
            size_t o = (size_t)y * simW + x;
            ch[0][o] = normalise<Sample>(p[0]);
            ch[1][o] = normalise<Sample>(p[1]);
            ch[2][o] = normalise<Sample>(p[2]);
        }
    }

    clk.lap("planes");
    // Encoder::convertToYCbCr, Encoder.cpp:15-23.
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)plane; ++i) {
        float y, cb, cr;
        rgbToYCbCr(ch[0][i], ch[1][i], ch[2][i], y, cb, cr);
        ch[0][i] = y; ch[1][i] = cb; ch[2][i] = cr;
    }

    for (const GainOp& op : g_gains)        // multiplyAVX over the whole plane, SIMD256.h:246-258
        if (op.stage == 0)
            for (size_t i = 0; i < plane; ++i) ch[op.channel][i] *= op.value;

    clk.lap("colour");
    // Encoder::reduceResolutionBySchema, Encoder.cpp:157-166: all width passes, then all
    // height passes, in place and compacting.
    int cw[3] = {simW, simW, simW}, chh[3] = {simH, simH, simH};
    for (int c = 0; c < 3; ++c) {           // Encoder.cpp:40-92
        int f = sc.wf[c];
        if (f == 1) continue;
        float* p = ch[c].data();
        int64_t n = (int64_t)cw[c] * chh[c];
        cw[c] /= f;
        if (sc.wm[c] == 1 && f == 2) {      // halfWidthResolutionAverageAVX, SIMD256.h:260-277
            for (int64_t s = 0; s < n; s += 2) p[s / 2] = (p[s] + p[s + 1]) / 2.0f;
        } else if (sc.wm[c] == 0) {         // subsampling keeps the first of each group
            for (int64_t s = 0; s < n; s += f) p[s / f] = p[s];
        } else {                            // Encoder.cpp:80-91
            float sum = 0.0f;
            for (int64_t s = 0, d = 0; s < n;) {
                sum += p[s++];
                if (s % f == 0) { p[d++] = sum / f; sum = 0.0f; }
            }
        }
    }
    for (int c = 0; c < 3; ++c) {           // Encoder.cpp:95-155
        int f = sc.hf[c];
        if (f == 1) continue;
        float* p = ch[c].data();
        const int64_t w = cw[c];
        const int64_t n = w * chh[c];
        chh[c] /= f;
        if (sc.hm[c] == 1 && f == 2) {      // halfHeightResolutionAverageAVX, SIMD256.h:295-304
            // Row by row (the vector order of the reference's kernel); every output depends on
            // its own two inputs only, and output row r is written after rows 2r and 2r + 1 are read.
            for (int64_t r = 0; r < n / (w * 2); ++r) {
                float* d = p + r * w;
                const float* a = p + 2 * r * w;
                const float* b = a + w;
                for (int64_t x = 0; x < w; ++x) d[x] = (a[x] + b[x]) / 2.0f;
            }
        } else if (sc.hm[c] == 0) {
            for (int64_t s = 0; s < n; s += w * f) memmove(p + s / f, p + s, (size_t)w * sizeof(float));
        } else {                            // Encoder.cpp:137-154
            float sum = 0.0f;
            int cnt = 0;
            for (int64_t x = 0; x < w; ++x) {
                for (int64_t s = x, d = x; s < n; s += w) {
                    sum += p[s];
                    if (++cnt == f) { p[d] = sum / f; d += w; sum = 0.0f; cnt = 0; }
                }
            }
        }
    }
    for (int c = 0; c < 3; ++c) { res.chW[c] = cw[c]; res.chH[c] = chh[c]; }
    for (const GainOp& op : g_gains)
        if (op.stage == 1)
            for (int64_t i = 0; i < (int64_t)cw[op.channel] * chh[op.channel]; ++i) ch[op.channel][i] *= op.value;

    clk.lap("reduce");
    // Encoder::applyDCT per channel, Encoder.cpp:229-258.
    std::vector<BlockSymbols> sym[3];
    const int mcusPerRow = simW / sc.stepW;
    for (int c = 0; c < 3; ++c) {
        // createBlocks, Encoder.cpp:168-227: MCU rows, MCU columns, block rows inside
        // the MCU, block columns inside the MCU.
        const size_t width = cw[c], height = chh[c];
        std::vector<size_t> origin;
        const size_t mcuH = 8 * sc.invH[c], mcuW = 8 * sc.invW[c];
        for (size_t my = 0; my < height; my += mcuH)
            for (size_t mx = 0; mx < width; mx += mcuW)
                for (size_t by = 0; by < mcuH; by += 8)
                    for (size_t bx = 0; bx < mcuW; bx += 8) origin.push_back((my + by) * width + mx + bx);
        float* p = ch[c].data();
        const int64_t nb = (int64_t)origin.size();
#pragma omp parallel for
        for (int64_t b = 0; b < nb; ++b) dctQuantBlock(p + origin[b], width, qt[c]);

        res.coef[c].resize((size_t)nb * 64);
        sym[c].resize(nb);
        const int blocksPerMcu = sc.invW[c] * sc.invH[c];
        // calculateDCValues, Encoder.cpp:260-289 (serial predictor).  Extension: the
        // predictor restarts at 0 every restartInterval MCUs.
        short lastDC = 0;
        for (int64_t b = 0; b < nb; ++b) {
            if (restartInterval > 0 && b % ((int64_t)blocksPerMcu * restartInterval) == 0) lastDC = 0;
            short dc = (short)p[origin[b]];
            short diff = (short)(dc - lastDC);
            unsigned short pat = (unsigned short)diff;
            if (diff < 0) pat = (unsigned short)~((unsigned short)-diff);
            int cat = bitCategory(diff);
            pat = (unsigned short)(cat == 0 ? 0 : pat << (16 - cat));
            sym[c][b].dcCat = (u8)cat;
            sym[c][b].dcBits = pat;
            lastDC = dc;
        }
        // calculateACValues, Encoder.cpp:291-352.
#pragma omp parallel for
        for (int64_t b = 0; b < nb; ++b) {
            BlockSymbols& bs = sym[c][b];
            int16_t* cz = res.coef[c].data() + b * 64;
            for (int i = 0; i < 64; ++i) cz[i] = (short)p[origin[b] + (kZigZag[i] % 8) + width * (kZigZag[i] / 8)];
            int zeros = 0;
            for (int i = 1; i < 64; ++i) {
                short v = cz[i];
                if (v != 0) {
                    while (zeros > 15) { bs.acBits.push_back(0); bs.acSym.push_back(0xF0); zeros -= 16; }
                    unsigned short pat = v < 0 ? (unsigned short)~((unsigned short)-v) : (unsigned short)v;
                    int cat = bitCategory(v);
                    pat = (unsigned short)(pat << (16 - cat));
                    bs.acBits.push_back(pat);
                    bs.acSym.push_back((u8)(zeros << 4 | cat));
                    zeros = 0;
                } else {
                    ++zeros;
                }
            }
            if (zeros != 0) { bs.acBits.push_back(0); bs.acSym.push_back(0); }
        }
    }

    clk.lap("dct + symbols");
    // Encoder::createHuffmanTable, Encoder.cpp:368-416: four tables, chroma shared.
    memset(res.hist, 0, sizeof res.hist);
    for (int c = 0; c < 3; ++c) {
        int t = c == 0 ? 0 : 1;
        for (const BlockSymbols& bs : sym[c]) {
            res.hist[t][bs.dcCat]++;
            for (u8 s : bs.acSym) res.hist[2 + t][s]++;
        }
    }
    chooseTables(histOverride, res);

    clk.lap("histograms + tables");
    // Encoder::serialize, Encoder.cpp:439-488.
    BitSink sink;
    std::vector<u8> hdr;
    headerBytes(W, H, headerH, sc, qt, res.tables, restartInterval, hdr);
    sink.pushBytes(hdr.data(), hdr.size());
    res.scanOffset = sink.byteLen();

    // Encoder::serializeScanData / pushBlock, Encoder.cpp:490-536.
    const size_t nMcu = sym[0].size() / ((size_t)sc.invW[0] * sc.invH[0]);
    (void)mcusPerRow;
    auto pushBlock = [&](int c, size_t b) {
        const int t = c == 0 ? 0 : 1;
        const BlockSymbols& bs = sym[c][b];
        u8 buf[4];
        auto pushCode = [&](const HuffTable& ht, int s) {
            uint32_t v = (uint32_t)ht.code[s] << (32 - ht.len[s]);
            buf[0] = (u8)(v >> 24); buf[1] = (u8)(v >> 16); buf[2] = (u8)(v >> 8); buf[3] = 0;
            sink.push(ht.len[s], buf, true);
        };
        auto pushPattern = [&](int n, uint16_t pat) {
            buf[0] = (u8)(pat >> 8); buf[1] = (u8)pat; buf[2] = 0; buf[3] = 0;
            sink.push(n, buf, true);
        };
        pushCode(res.tables[t], bs.dcCat);
        pushPattern(bs.dcCat, bs.dcBits);
        for (size_t i = 0; i < bs.acSym.size(); ++i) {
            pushCode(res.tables[2 + t], bs.acSym[i]);
            pushPattern(bs.acSym[i] & 15, bs.acBits[i]);
        }
    };
    for (size_t m = 0; m < nMcu; ++m) {
        if (restartInterval > 0 && m > 0 && m % restartInterval == 0) {   // extension: RSTn
            sink.fillToByte(padOnes);
            const u8 rst[2] = {0xff, (u8)(0xd0 + ((rstBase + m / restartInterval - 1) & 7))};
            sink.pushBytes(rst, 2);
        }
        for (int c = 0; c < 3; ++c) {
            size_t f = (size_t)sc.invW[c] * sc.invH[c];
            for (size_t k = 0; k < f; ++k) pushBlock(c, m * f + k);
        }
    }
    sink.fillToByte(padOnes);
    res.scanLen = sink.byteLen() - res.scanOffset;
    const u8 eoi[2] = {0xff, 0xd9};
    sink.pushBytes(eoi, 2);
    res.jpeg.assign(sink.data.begin(), sink.data.begin() + sink.byteLen());
    clk.lap("bit stream");
}

}  // namespace

// ----------------------------------------------------------------------------------------
// C interface used by tests / smoke / the cpu_baseline leg (ctypes).
// ----------------------------------------------------------------------------------------
extern "C" {

// AAN constants as this restatement derives them: a1 a2 a4 a5 s0..s7.
void orc_constants(float out[12]) {
    const AanConst& c = aan();
    out[0] = c.a1; out[1] = c.a2; out[2] = c.a4; out[3] = c.a5;
    for (int k = 0; k < 8; ++k) out[4 + k] = c.s[k];
}

void orc_color_convert(float r, float g, float b, float out[3]) { rgbToYCbCr(r, g, b, out[0], out[1], out[2]); }

// DCT::araiDCTAVX (DCT.cpp:373-376): 2-D transform of 64 row-major floats, DC - 1024.
void orc_dct_block(const float in[64], float out[64]) {
    float b[64];
    memcpy(b, in, sizeof b);
    aan2d(b);
    memcpy(out, b, sizeof b);
}

// DCT::araiDCTandQuantisationAVX (DCT.cpp:385-388): input in [0,1].
void orc_dct_quant_block(const float in[64], const float q[64], float out[64]) {
    float b[64];
    memcpy(b, in, sizeof b);
    dctQuantBlock(b, 8, q);
    memcpy(out, b, sizeof b);
}

// HuffmanTable<byte>::create(16, symbols) from a histogram.  dht receives the DHT
// segment (marker included); returns its length.
int orc_huffman(const uint64_t counts[256], int htNum, int htType, uint8_t lengths[256], uint16_t codes[256],
                uint8_t* dht, int dhtCap) {
    HuffTable t;
    buildHuffman(counts, t);
    memcpy(lengths, t.len, 256);
    memcpy(codes, t.code, 512);
    std::vector<u8> seg;
    if (t.symbolCount > 0) dhtBytes(t, htNum, htType, seg);
    if ((int)seg.size() > dhtCap) return -1;
    if (!seg.empty()) memcpy(dht, seg.data(), seg.size());
    return (int)seg.size();
}

// HuffmanTable<byte>::create(codeWordLength, symbols) for any limit; -1 where the reference
// returns a null table (HuffmanCodingByte.h:105, :121).
int orc_huffman_limit(const uint64_t counts[256], int codeWordLength, uint8_t lengths[256], uint64_t codes[256]) {
    if (codeWordLength > 64) return -1;
    if ((double)codeWordLength < std::ceil(std::log2(256.0 + 1.0))) return -1;
    HuffTable t;
    buildHuffman(counts, t, (size_t)codeWordLength);
    memcpy(lengths, t.len, 256);
    memcpy(codes, t.wide, sizeof t.wide);
    return 0;
}

// BitBuffer::pushBits[Escaped] replay: n pushes of (nbits[i], 4 source bytes each,
// MSB aligned).  Returns the byte length; *bitLen receives the bit length.
int64_t orc_bitbuffer_replay(int n, const int32_t* nbits, const uint8_t* src4, int escape, int fill, uint8_t* out,
                             int64_t cap, int64_t* bitLen) {
    BitSink s;
    for (int i = 0; i < n; ++i) {
        u8 buf[5] = {src4[4 * i], src4[4 * i + 1], src4[4 * i + 2], src4[4 * i + 3], 0};
        s.push((size_t)nbits[i], buf, escape != 0);
    }
    if (fill) s.fillToByte(false);
    *bitLen = (int64_t)s.bitPos;
    int64_t len = (int64_t)s.byteLen();
    if (len > cap) return -1;
    memcpy(out, s.data.data(), (size_t)len);
    return len;
}

struct OrcResult {
    EncodeResult r;
};

// scheme: 12 ints, per channel {widthFactor, widthMethod, heightFactor, heightMethod},
// method 0 = Subsampling, 1 = Average.  qt: 3 x 64 floats, natural order.
// sampleType 0: rgb is uint8 (value/255.0f is applied), 1: rgb is float in [0,1].
// histOverride (may be null), rstBase and headerH serve band-wise encodes of a sharded
// image: tables from summed histograms, global RSTn numbering, whole-image height in SOF0.
void* orc_encode(const void* rgb, int sampleType, int W, int H, int64_t strideSamples, const int32_t* scheme,
                 const float* qt, int restartInterval, int padOnes, const uint64_t* histOverride, int rstBase,
                 int headerH) {
    OrcResult* o = new OrcResult();
    Scheme sc(scheme);
    float q[3][64];
    memcpy(q, qt, sizeof q);
    if (sampleType == 0)
        encodeImage<u8>((const u8*)rgb, W, H, (size_t)strideSamples, sc, q, restartInterval, padOnes != 0,
                        histOverride, rstBase, headerH, o->r);
    else
        encodeImage<float>((const float*)rgb, W, H, (size_t)strideSamples, sc, q, restartInterval, padOnes != 0,
                           histOverride, rstBase, headerH, o->r);
    return o;
}
// Tables [DC-Y, DC-C, AC-Y, AC-C] x 256 to use instead of optimised ones; null switches back.
void orc_set_fixed_tables(const uint8_t* lengths, const uint16_t* codes) {
    g_fixedTables = lengths != nullptr && codes != nullptr;
    if (g_fixedTables) {
        memcpy(g_fixedLen, lengths, sizeof g_fixedLen);
        memcpy(g_fixedCode, codes, sizeof g_fixedCode);
    }
}

// Channel gains applied by the following encodes (n = 0 switches them off).
void orc_set_channel_gains(int n, const int32_t* channels, const int32_t* stages, const float* values) {
    g_gains.clear();
    for (int i = 0; i < n; ++i) g_gains.push_back({channels[i], stages[i], values[i]});
}

void orc_free(void* h) { delete (OrcResult*)h; }
int64_t orc_jpeg_size(void* h) { return (int64_t)((OrcResult*)h)->r.jpeg.size(); }
void orc_jpeg_copy(void* h, uint8_t* dst) { const auto& j = ((OrcResult*)h)->r.jpeg; memcpy(dst, j.data(), j.size()); }
int64_t orc_scan_offset(void* h) { return (int64_t)((OrcResult*)h)->r.scanOffset; }
int64_t orc_scan_len(void* h) { return (int64_t)((OrcResult*)h)->r.scanLen; }
int64_t orc_coef_count(void* h, int c) { return (int64_t)((OrcResult*)h)->r.coef[c].size(); }
void orc_coef_copy(void* h, int c, int16_t* dst) { const auto& v = ((OrcResult*)h)->r.coef[c]; memcpy(dst, v.data(), v.size() * 2); }
void orc_hist_copy(void* h, uint64_t* dst) { memcpy(dst, ((OrcResult*)h)->r.hist, sizeof(uint64_t) * 4 * 256); }
void orc_table_copy(void* h, int t, uint8_t* lengths, uint16_t* codes) {
    memcpy(lengths, ((OrcResult*)h)->r.tables[t].len, 256);
    memcpy(codes, ((OrcResult*)h)->r.tables[t].code, 512);
}
void orc_dims(void* h, int32_t out[8]) {
    const EncodeResult& r = ((OrcResult*)h)->r;
    out[0] = r.simW; out[1] = r.simH;
    for (int c = 0; c < 3; ++c) { out[2 + 2 * c] = r.chW[c]; out[3 + 2 * c] = r.chH[c]; }
}

}  // extern "C"
