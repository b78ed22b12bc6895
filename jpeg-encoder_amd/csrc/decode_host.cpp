// Host half of the verification decoder: JFIF parsing and Huffman decoding of a baseline,
// three-component, single-scan JPEG into quantized coefficients laid out exactly like the
// encoder's (blocks in scan order, 64 zig-zag ordered values per block), so that a file
// can be checked against the coefficients it was made from and handed to the inverse
// kernels.  The reference has no JPEG decoder; the pieces it does have for the way back
// (HuffmanTable::decode2, HuffmanCoding.h:244-304: match the next maxCodewordLength bits
// against the codes sorted by length and value) are what the 16-bit lookup below amounts to.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "decode.h"

namespace jpeghip {

namespace {

struct HuffLookup {
    // For every 16-bit window: (code length << 8) | symbol, 0 if no code matches.
    std::vector<uint16_t> entry;
    bool defined = false;
};

bool build_lookup(const uint8_t bits[16], const uint8_t* values, int count, HuffLookup& out) {
    out.entry.assign(65536, 0);
    unsigned code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1]; ++i) {
            if (k >= count || code >= (1u << len)) return false;
            const unsigned first = code << (16 - len), span = 1u << (16 - len);
            const uint16_t e = (uint16_t)((len << 8) | values[k++]);
            for (unsigned w = first; w < first + span; ++w) out.entry[w] = e;
            ++code;
        }
        code <<= 1;
    }
    out.defined = true;
    return k == count;
}

// Bit reader over one restart interval's bytes; 0xFF00 unstuffing on the fly
// (the inverse of BitBuffer.cpp:156-161).
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;   // bits left-aligned
    int n = 0;
    void fill() {
        while (n <= 56) {
            uint64_t byte = 0;
            if (p < end) {
                byte = *p++;
                if (byte == 0xFF && p < end && *p == 0x00) ++p;
            }
            acc |= byte << (56 - n);
            n += 8;
        }
    }
    unsigned peek16() {
        if (n < 16) fill();
        return (unsigned)(acc >> 48);
    }
    void skip(int k) {
        acc <<= k;
        n -= k;
    }
    unsigned take(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        const unsigned v = (unsigned)(acc >> (64 - k));
        skip(k);
        return v;
    }
};

inline int extend(unsigned v, int s) {   // T.81 F.2.2.1
    return s == 0 ? 0 : ((int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v);
}

bool decode_symbol(BitReader& br, const HuffLookup& t, int& sym) {
    const uint16_t e = t.entry[br.peek16()];
    if (e == 0) return false;
    br.skip(e >> 8);
    sym = e & 0xff;
    return true;
}

struct Segment {
    const uint8_t* begin;
    const uint8_t* end;
};

}  // namespace

int parse_headers(const uint8_t* d, int64_t len, DecodedStream& out, std::string& err) {
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) {
        err = "not a JPEG file (no SOI)";
        return JPEG_HIP_ERR_INVALID_ARGUMENT;
    }
    bool haveSof = false, haveQ[4] = {false, false, false, false};
    out.restartInterval = 0;
    int64_t i = 2;
    while (i + 4 <= len) {
        if (d[i] != 0xFF) {
            err = "marker expected";
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        }
        const int marker = d[i + 1];
        if (marker == 0xFF) {   // fill byte
            ++i;
            continue;
        }
        const int64_t segLen = ((int64_t)d[i + 2] << 8) | d[i + 3];
        if (segLen < 2 || i + 2 + segLen > len) {
            err = "segment runs past the end of the file";
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        }
        const uint8_t* s = d + i + 4;
        const int64_t n = segLen - 2;
        if (marker == 0xDB) {   // DQT, JPEGSegments.h:142-176
            int64_t j = 0;
            while (j < n) {
                const int pq = s[j] >> 4, tq = s[j] & 15;
                if (pq != 0 || tq > 3 || j + 65 > n) {
                    err = "unsupported quantization table";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
                for (int k = 0; k < 64; ++k) out.qtables[tq][k] = s[j + 1 + k];
                haveQ[tq] = true;
                j += 65;
            }
        } else if (marker == 0xC0) {   // SOF0, JPEGSegments.h:77-100
            if (n < 15 || s[0] != 8 || s[5] != 3) {
                err = "only 8-bit, three-component baseline frames are supported";
                return JPEG_HIP_ERR_INVALID_ARGUMENT;
            }
            out.height = (s[1] << 8) | s[2];
            out.width = (s[3] << 8) | s[4];
            for (int c = 0; c < 3; ++c) {
                out.componentId[c] = s[6 + 3 * c];
                out.hSamp[c] = s[7 + 3 * c] >> 4;
                out.vSamp[c] = s[7 + 3 * c] & 15;
                out.tq[c] = s[8 + 3 * c];
                if (out.hSamp[c] < 1 || out.hSamp[c] > 4 || out.vSamp[c] < 1 || out.vSamp[c] > 4 || out.tq[c] > 3) {
                    err = "bad component parameters";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
            }
            haveSof = true;
        } else if (marker >= 0xC1 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
            err = "only baseline (SOF0) files are supported";
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        } else if (marker == 0xC4) {   // DHT, JPEGSegments.h:102-140
            int64_t j = 0;
            while (j < n) {
                if (j + 17 > n) {
                    err = "truncated Huffman table";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
                const int tc = s[j] >> 4, th = s[j] & 15;
                int count = 0;
                for (int k = 0; k < 16; ++k) count += s[j + 1 + k];
                if (tc > 1 || th > 1 || count > 256 || j + 17 + count > n) {
                    err = "unsupported Huffman table";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
                DecodedStream::Dht& t = out.dht[tc * 2 + th];
                memcpy(t.bits, s + j + 1, 16);
                memcpy(t.values, s + j + 17, (size_t)count);
                t.count = count;
                j += 17 + count;
            }
        } else if (marker == 0xDD) {   // DRI
            if (n != 2) {
                err = "bad DRI segment";
                return JPEG_HIP_ERR_INVALID_ARGUMENT;
            }
            out.restartInterval = (s[0] << 8) | s[1];
        } else if (marker == 0xDA) {   // SOS, JPEGSegments.h:178-201
            if (!haveSof || n < 10 || s[0] != 3) {
                err = "only a single interleaved three-component scan is supported";
                return JPEG_HIP_ERR_INVALID_ARGUMENT;
            }
            for (int c = 0; c < 3; ++c) {
                if (s[1 + 2 * c] != out.componentId[c]) {
                    err = "scan components out of frame order";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
                out.td[c] = s[2 + 2 * c] >> 4;
                out.ta[c] = s[2 + 2 * c] & 15;
                if (out.td[c] > 1 || out.ta[c] > 1 || out.dht[out.td[c]].count == 0 || out.dht[2 + out.ta[c]].count == 0 ||
                    !haveQ[out.tq[c]]) {
                    err = "scan refers to a table the file does not define";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
            }
            if (s[7] != 0 || s[8] != 63 || s[9] != 0) {
                err = "not a sequential scan";
                return JPEG_HIP_ERR_INVALID_ARGUMENT;
            }
            out.scanOffset = i + 2 + segLen;
            int hMax = 1, vMax = 1;
            for (int c = 0; c < 3; ++c) {
                hMax = std::max(hMax, out.hSamp[c]);
                vMax = std::max(vMax, out.vSamp[c]);
            }
            out.blocksPerMcu = 0;
            for (int c = 0; c < 3; ++c) {
                if (hMax % out.hSamp[c] != 0 || vMax % out.vSamp[c] != 0) {
                    err = "fractional sampling ratios are not supported";
                    return JPEG_HIP_ERR_INVALID_ARGUMENT;
                }
                out.chanOffset[c] = out.blocksPerMcu;
                out.blocksPerMcu += out.hSamp[c] * out.vSamp[c];
            }
            out.mcuW = hMax * 8;
            out.mcuH = vMax * 8;
            out.mcusPerRow = (out.width + out.mcuW - 1) / out.mcuW;
            out.mcuRows = (out.height + out.mcuH - 1) / out.mcuH;
            out.totalBlocks = (int64_t)out.mcusPerRow * out.mcuRows * out.blocksPerMcu;
            return JPEG_HIP_OK;
        }
        i += 2 + segLen;
    }
    err = "no scan found";
    return JPEG_HIP_ERR_INVALID_ARGUMENT;
}

int decode_entropy(const uint8_t* d, int64_t len, DecodedStream& out, std::string& err) {
    const int st = parse_headers(d, len, out, err);
    if (st != JPEG_HIP_OK) return st;
    HuffLookup lut[4];
    for (int t = 0; t < 4; ++t)
        if (out.dht[t].count > 0 && !build_lookup(out.dht[t].bits, out.dht[t].values, out.dht[t].count, lut[t])) {
            err = "inconsistent Huffman table";
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        }
    // Cut the entropy-coded data at the RSTn markers (position-independent: a 0xFF that is
    // neither stuffed nor a restart marker ends the scan).
    std::vector<Segment> segs;
    const uint8_t* p = d + out.scanOffset;
    const uint8_t* const fileEnd = d + len;
    const uint8_t* segBegin = p;
    bool sawEoi = false;
    while (p < fileEnd) {
        if (*p != 0xFF) {
            ++p;
            continue;
        }
        if (p + 1 >= fileEnd) break;
        const uint8_t m = p[1];
        if (m == 0x00) {
            p += 2;
        } else if (m >= 0xD0 && m <= 0xD7) {
            if (m != 0xD0 + (segs.size() & 7)) {
                err = "restart markers out of sequence";
                return JPEG_HIP_ERR_INVALID_ARGUMENT;
            }
            segs.push_back({segBegin, p});
            p += 2;
            segBegin = p;
        } else if (m == 0xFF) {
            ++p;
        } else {
            sawEoi = m == 0xD9;
            break;
        }
    }
    segs.push_back({segBegin, p});
    if (!sawEoi) {
        err = "no EOI after the scan";
        return JPEG_HIP_ERR_INVALID_ARGUMENT;
    }
    const int64_t nMcu = (int64_t)out.mcusPerRow * out.mcuRows;
    const int64_t perSeg = out.restartInterval > 0 ? out.restartInterval : nMcu;
    if ((int64_t)segs.size() != (nMcu + perSeg - 1) / perSeg) {
        err = "number of restart intervals does not match the frame";
        return JPEG_HIP_ERR_INVALID_ARGUMENT;
    }
    out.coef.assign((size_t)out.totalBlocks * 64, 0);

    std::vector<int> status(segs.size(), 0);
    auto work = [&](size_t s0, size_t s1) {
        for (size_t si = s0; si < s1; ++si) {
            BitReader br{segs[si].begin, segs[si].end};
            int pred[3] = {0, 0, 0};
            const int64_t m0 = (int64_t)si * perSeg, m1 = std::min(nMcu, m0 + perSeg);
            for (int64_t m = m0; m < m1 && status[si] == 0; ++m) {
                int16_t* blk = out.coef.data() + (size_t)m * out.blocksPerMcu * 64;
                for (int c = 0; c < 3 && status[si] == 0; ++c) {
                    const HuffLookup& dc = lut[out.td[c]];
                    const HuffLookup& ac = lut[2 + out.ta[c]];
                    for (int k = 0; k < out.hSamp[c] * out.vSamp[c]; ++k, blk += 64) {
                        int sym;
                        if (!decode_symbol(br, dc, sym) || sym > 11) {
                            status[si] = 1;
                            break;
                        }
                        pred[c] += extend(br.take(sym), sym);
                        blk[0] = (int16_t)pred[c];
                        int z = 1;
                        while (z < 64) {
                            if (!decode_symbol(br, ac, sym)) {
                                status[si] = 1;
                                break;
                            }
                            const int run = sym >> 4, size = sym & 15;
                            if (size == 0) {
                                if (run != 15) break;   // EOB
                                z += 16;                // ZRL
                                continue;
                            }
                            z += run;
                            if (z > 63) {
                                status[si] = 1;
                                break;
                            }
                            blk[z++] = (int16_t)extend(br.take(size), size);
                        }
                        if (status[si] != 0) break;
                    }
                }
            }
        }
    };
    unsigned hw = std::thread::hardware_concurrency();
    size_t workers = hw == 0 ? 1 : (hw > 16 ? 16 : hw);
    if (workers > segs.size()) workers = segs.size();
    if (workers <= 1) {
        work(0, segs.size());
    } else {
        std::vector<std::thread> pool;
        const size_t per = (segs.size() + workers - 1) / workers;
        for (size_t w = 0; w < workers; ++w) {
            const size_t a = w * per, b = std::min(segs.size(), a + per);
            if (a < b) pool.emplace_back(work, a, b);
        }
        for (auto& t : pool) t.join();
    }
    for (int s : status)
        if (s != 0) {
            err = "corrupt entropy-coded data";
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        }
    return JPEG_HIP_OK;
}

}  // namespace jpeghip
