// Host side of the entropy stage: length-limited Huffman tables from symbol histograms
// and the JFIF header segments.  No GPU code in this file.
//
// The reference builds its tables with a package-merge pass over shared_ptr trees
// (HuffmanCodingByte.h:102-210) and breaks frequency ties through libstdc++'s unstable
// std::sort, so byte-identical DHT segments need the same sort calls on sequences of the
// same length and initial order.  This implementation keeps those calls but works on a
// flat node pool and derives the code lengths by pushing selection counts from parents to
// children instead of walking every selected tree.
#include <algorithm>
#include <cstring>
#include <vector>

#include "internal.h"

namespace jpeghip {

namespace {

const int kZigZagOrder[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

}  // namespace

const int* zigzag_order() { return kZigZagOrder; }

void huffman_from_histogram(const uint64_t counts[256], uint8_t lengths[256], uint16_t codes[256]) {
    uint64_t wide[256];
    huffman_from_histogram_limit(counts, 16, lengths, wide);
    for (int s = 0; s < 256; ++s) codes[s] = (uint16_t)wide[s];
}

void huffman_from_histogram_limit(const uint64_t counts[256], int maxLen, uint8_t lengths[256], uint64_t codes[256]) {
    memset(lengths, 0, 256);
    memset(codes, 0, 256 * sizeof(uint64_t));

    // Leaves in ascending symbol order, then the zero-frequency dummy that keeps the
    // all-ones code unused (HuffmanCodingByte.h:124-135).
    int leafSymbol[257];
    int nLeaves = 0;
    for (int s = 0; s < 256; ++s)
        if (counts[s] != 0) leafSymbol[nLeaves++] = s;
    if (nLeaves == 0) return;  // empty table, HuffmanCodingByte.h:110
    const int dummy = nLeaves;
    leafSymbol[nLeaves++] = 0;

    const int kMaxLen = maxLen;
    // Pool: leaves first, then every package in creation order, so a package's children
    // always sit at lower indices than the package itself.
    std::vector<uint64_t> freq;
    std::vector<int> lhs, rhs;
    freq.reserve(nLeaves * kMaxLen);
    lhs.reserve(nLeaves * kMaxLen);
    rhs.reserve(nLeaves * kMaxLen);
    for (int i = 0; i < nLeaves; ++i) {
        freq.push_back(i == dummy ? 0 : counts[leafSymbol[i]]);
        lhs.push_back(-1);
        rhs.push_back(-1);
    }
    auto lessFreq = [&freq](int a, int b) { return freq[a] < freq[b]; };

    std::vector<int> level(nLeaves), next;
    for (int i = 0; i < nLeaves; ++i) level[i] = i;
    for (int round = 1; round < kMaxLen; ++round) {  // HuffmanCodingByte.h:139-156
        std::sort(level.begin(), level.end(), lessFreq);
        next.resize(nLeaves);
        for (int i = 0; i < nLeaves; ++i) next[i] = i;
        for (size_t i = 1; i < level.size(); i += 2) {
            next.push_back((int)freq.size());
            freq.push_back(freq[level[i - 1]] + freq[level[i]]);
            lhs.push_back(level[i - 1]);
            rhs.push_back(level[i]);
        }
        level.swap(next);
    }
    std::sort(level.begin(), level.end(), lessFreq);  // HuffmanCodingByte.h:159-162

    // The cheapest 2n-2 items each add one to the length of every leaf below them
    // (HuffmanCodingByte.h:165-168).  Selection counts flow parent -> children.
    std::vector<uint32_t> weight(freq.size(), 0);
    for (int i = 0; i < 2 * nLeaves - 2; ++i) weight[level[i]]++;
    for (int n = (int)freq.size() - 1; n >= nLeaves; --n) {
        weight[lhs[n]] += weight[n];
        weight[rhs[n]] += weight[n];
    }

    // Longest code first, ties in std::sort order (HuffmanCodingByte.h:171-178), then
    // count down from the all-ones word (HuffmanCodingByte.h:181-207).
    std::vector<int> order(nLeaves);
    for (int i = 0; i < nLeaves; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&weight](int a, int b) { return weight[a] > weight[b]; });
    uint32_t lastLen = weight[order[0]];
    uint64_t code = ~uint64_t(0) >> (64 - lastLen);
    for (int i = 0; i < nLeaves; ++i) {
        const int leaf = order[i];
        if (leaf == dummy) continue;  // the only zero-frequency leaf
        const uint32_t len = weight[leaf];
        if (len != lastLen) code >>= (lastLen - len);
        --code;
        lengths[leafSymbol[leaf]] = (uint8_t)len;
        codes[leafSymbol[leaf]] = code;
        lastLen = len;
    }
}

// A table the kernels and the DHT writer can work with: no code longer than 16 bits, no
// code wider than its length, and lengths that satisfy the Kraft inequality (a prefix code
// with these lengths exists).
bool huffman_table_is_sane(const uint8_t lengths[256], const uint16_t codes[256]) {
    uint64_t kraft = 0;   // in units of 2^-16
    for (int s = 0; s < 256; ++s) {
        const int len = lengths[s];
        if (len == 0) continue;
        if (len > 16) return false;
        if (len < 16 && (codes[s] >> len) != 0) return false;
        kraft += 1ull << (16 - len);
    }
    return kraft <= (1ull << 16);
}

int write_dht(const uint8_t lengths[256], const uint16_t codes[256], int tableId, int coefType, uint8_t* out,
              int capacity) {
    if (!huffman_table_is_sane(lengths, codes)) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    // Symbols ordered by (length, code), JPEGSegments.h:117-129.
    int syms[256];
    int n = 0;
    for (int s = 0; s < 256; ++s)
        if (lengths[s] != 0) syms[n++] = s;
    if (n == 0) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    std::sort(syms, syms + n, [&](int a, int b) {
        return lengths[a] != lengths[b] ? lengths[a] < lengths[b] : codes[a] < codes[b];
    });
    const int total = 2 + 2 + 1 + 16 + n;
    if (capacity < total) return JPEG_HIP_ERR_BUFFER_TOO_SMALL;
    uint8_t* p = out;
    *p++ = 0xff;
    *p++ = 0xc4;
    const int segLen = total - 2;
    *p++ = (uint8_t)(segLen >> 8);
    *p++ = (uint8_t)segLen;
    *p++ = (uint8_t)((tableId & 15) | ((coefType & 1) << 4));
    memset(p, 0, 16);
    for (int i = 0; i < n; ++i) p[lengths[syms[i]] - 1]++;
    p += 16;
    for (int i = 0; i < n; ++i) *p++ = (uint8_t)syms[i];
    return total;
}

int write_headers(const jpeg_hip_params& prm, const Geometry& g, const uint8_t lengths[4][256],
                  const uint16_t codes[4][256], uint8_t* out, int capacity) {
    std::vector<uint8_t> h;
    h.reserve(1024);
    auto put = [&h](std::initializer_list<int> bytes) {
        for (int b : bytes) h.push_back((uint8_t)b);
    };
    put({0xff, 0xd8});                                                                  // SOI, JPEGSegments.h:50-54
    put({0xff, 0xe0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00, 0x01, 0x01, 0x00, 0x00, 0x48, 0x00, 0x48, 0x00, 0x00});  // :62-75
    put({0xff, 0xc0, 0x00, 0x11, 0x08, (prm.height >> 8) & 255, prm.height & 255, (prm.width >> 8) & 255,
         prm.width & 255, 0x03});                                                        // SOF0, :77-100
    for (int c = 0; c < 3; ++c) put({c + 1, (g.hFactor[c] << 4) | (g.vFactor[c] & 15), c});
    uint8_t seg[2 + 2 + 1 + 16 + 256];
    const int ids[4] = {0, 1, 0, 1}, types[4] = {JPEG_HIP_DC, JPEG_HIP_DC, JPEG_HIP_AC, JPEG_HIP_AC};  // Encoder.cpp:455-479
    for (int t = 0; t < 4; ++t) {
        int n = write_dht(lengths[t], codes[t], ids[t], types[t], seg, (int)sizeof seg);
        if (n < 0) return n;
        h.insert(h.end(), seg, seg + n);
    }
    for (int c = 0; c < 3; ++c) {                                                        // DQT, :152-167
        put({0xff, 0xdb, 0x00, 0x43, c});
        for (int i = 0; i < 64; ++i) h.push_back((uint8_t)prm.qtable[c][kZigZagOrder[i]]);
    }
    if (prm.restart_interval > 0)
        put({0xff, 0xdd, 0x00, 0x04, (prm.restart_interval >> 8) & 255, prm.restart_interval & 255});
    put({0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00});  // SOS, :175-190
    if ((int)h.size() > capacity) return JPEG_HIP_ERR_BUFFER_TOO_SMALL;
    memcpy(out, h.data(), h.size());
    return (int)h.size();
}

// ---- typical tables of ITU-T T.81 Annex K.3 ------------------------------------------------
namespace {
const uint8_t kStdBits[4][16] = {
    {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},          // K.3 DC luminance
    {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},          // K.4 DC chrominance
    {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d},       // K.5 AC luminance
    {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77},       // K.6 AC chrominance
};
const uint8_t kStdDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChrVal[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
}  // namespace

void standard_huffman(uint8_t lengths[4][256], uint16_t codes[4][256]) {
    const uint8_t* vals[4] = {kStdDcVal, kStdDcVal, kStdAcLumVal, kStdAcChrVal};
    for (int t = 0; t < 4; ++t) {
        memset(lengths[t], 0, 256);
        memset(codes[t], 0, 256 * sizeof(uint16_t));
        // Annex C: codes of one length are consecutive; going to the next length appends a 0.
        unsigned code = 0;
        int k = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int i = 0; i < kStdBits[t][len - 1]; ++i) {
                const uint8_t sym = vals[t][k++];
                lengths[t][sym] = (uint8_t)len;
                codes[t][sym] = (uint16_t)code++;
            }
            code <<= 1;
        }
    }
}

}  // namespace jpeghip
