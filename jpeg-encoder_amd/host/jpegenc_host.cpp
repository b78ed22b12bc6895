// Implementation of the host-side mirror of the reference's classes (jpegenc_host.h).
// Pixel work happens on the GPU through the C ABI; this file holds bookkeeping, file I/O
// and the bit container.
#include "jpegenc_host.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

// ---- SamplingDefinition --------------------------------------------------------------------
void SamplingDefinition::toAbi(int32_t out[12]) const {
    for (int c = 0; c < 3; ++c) {
        out[4 * c] = reductionOptions[c].widthFactor;
        out[4 * c + 1] = reductionOptions[c].widthMethod;
        out[4 * c + 2] = reductionOptions[c].heightFactor;
        out[4 * c + 3] = reductionOptions[c].heightMethod;
    }
}

// ---- quantization tables (QuantizationTables.h:40-110) -----------------------------------
namespace JPEGQuantization {
const QTable luminance = {{16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                           14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                           18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99}};
const QTable chrominance = {{17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                             99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
namespace Photoshop {
const QTable luminance12 = {{1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 2, 2, 3,
                             1, 1, 1, 1, 2, 2, 3, 3, 1, 1, 1, 2, 2, 3, 3, 3, 1, 1, 2, 2, 3, 3, 3, 3, 2, 2, 2, 3, 3, 3, 3, 3}};
const QTable chrominance12 = {{1, 1, 1, 2, 3, 3, 3, 3, 1, 1, 1, 2, 3, 3, 3, 3, 1, 1, 2, 3, 3, 3, 3, 3, 2, 2, 3, 3, 3, 3, 3, 3,
                               3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3}};
const QTable luminance5 = {{12, 8,  13, 21, 26, 32, 34, 17, 8,  9,  12, 20, 27, 23, 12, 12, 13, 12, 16, 26, 23, 12,
                            12, 12, 21, 20, 26, 23, 12, 12, 12, 12, 26, 27, 23, 12, 12, 12, 12, 12, 32, 23, 12, 12,
                            12, 12, 12, 12, 34, 12, 12, 12, 12, 12, 12, 12, 17, 12, 12, 12, 12, 12, 12, 12}};
const QTable chrominance5 = {{13, 13, 17, 27, 20, 20, 17, 17, 13, 14, 17, 14, 14, 12, 12, 12, 17, 17, 14, 14, 12, 12,
                              12, 12, 27, 14, 14, 12, 12, 12, 12, 12, 20, 14, 12, 12, 12, 12, 12, 12, 20, 12, 12, 12,
                              12, 12, 12, 12, 17, 12, 12, 12, 12, 12, 12, 12, 17, 12, 12, 12, 12, 12, 12, 12}};
}  // namespace Photoshop

static QTable scaled(const QTable& base, int quality) {
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    QTable t;
    for (int i = 0; i < 64; ++i) {
        int v = ((int)base.floats[i] * scale + 50) / 100;
        t.floats[i] = (float)std::min(255, std::max(1, v));
    }
    return t;
}
QTable scaledLuminance(int quality) { return scaled(luminance, quality); }
QTable scaledChrominance(int quality) { return scaled(chrominance, quality); }
}  // namespace JPEGQuantization

// ---- BitBuffer (BitBuffer.h:5-93, BitBuffer.cpp) -----------------------------------------------
// Bits are kept MSB first in a byte vector.  The escaped variants follow a byte that
// completes as 0xFF with a 0x00 byte (BitBuffer.cpp:156-161, 171-176, 190-195).
BitBuffer::BitBuffer(size_t initialBufferSizeInBit) : data(), dataBitOffset(0) {
    data.reserve(initialBufferSizeInBit == 0 ? 1 : (initialBufferSizeInBit + 7) / 8);
}

void BitBuffer::appendBit(bool bit, bool escape) {
    const size_t byteIdx = dataBitOffset >> 3;
    if (byteIdx >= data.size()) data.resize(byteIdx + 1, 0);
    if (bit) data[byteIdx] |= (byte)(0x80u >> (dataBitOffset & 7));
    ++dataBitOffset;
    if (escape && (dataBitOffset & 7) == 0 && data[byteIdx] == 0xff) {
        data.push_back(0x00);
        dataBitOffset += 8;
    }
}

void BitBuffer::pushBitsImpl(size_t numOfBits, const void* buffer, size_t offset, bool escape) {
    if (numOfBits == 0) return;
    const byte* src = static_cast<const byte*>(buffer);
    if (!escape && offset == 0 && (dataBitOffset & 7) == 0) {   // byte aligned bulk append
        const size_t whole = numOfBits / 8;
        data.resize(dataBitOffset / 8);
        data.insert(data.end(), src, src + whole);
        dataBitOffset += whole * 8;
        src += whole;
        numOfBits -= whole * 8;
    }
    for (size_t i = 0; i < numOfBits; ++i) {
        const size_t bitIdx = offset + i;
        appendBit((src[bitIdx >> 3] >> (7 - (bitIdx & 7))) & 1, escape);
    }
}

void BitBuffer::fillToByteBorder() {
    dataBitOffset = ((dataBitOffset + 7) / 8) * 8;   // BitBuffer.cpp:12-16: the pad bits stay zero
    if (data.size() < dataBitOffset / 8) data.resize(dataBitOffset / 8, 0);
}
void BitBuffer::pushBit(bool val) { appendBit(val, false); }
void BitBuffer::pushBits(const BitBuffer& b) { pushBitsImpl(b.getSize(), b.data.data(), 0, false); }
void BitBuffer::pushBits(size_t n, const void* buf, size_t offset) { pushBitsImpl(n, buf, offset, false); }
void BitBuffer::pushBits(size_t n, const void* buf) { pushBitsImpl(n, buf, 0, false); }
void BitBuffer::pushBitsEscaped(const BitBuffer& b) { pushBitsImpl(b.getSize(), b.data.data(), 0, true); }
void BitBuffer::pushBitsEscaped(size_t n, const void* buf) { pushBitsImpl(n, buf, 0, true); }
void BitBuffer::pushBitsEscaped(size_t n, const void* buf, size_t offset) { pushBitsImpl(n, buf, offset, true); }

bool BitBuffer::getBit(size_t index) const { return (data[index >> 3] >> (7 - (index & 7))) & 1; }

void BitBuffer::getBits(size_t index, void* outVoid, size_t numOfBits) const {
    byte* out = static_cast<byte*>(outVoid);
    for (size_t i = 0; i < (numOfBits + 7) / 8; ++i) out[i] = 0;
    for (size_t i = 0; i < numOfBits; ++i) {
        const size_t src = index + i;
        const bool bit = src < dataBitOffset && getBit(src);
        if (bit) out[i >> 3] |= (byte)(0x80u >> (i & 7));
    }
}

void BitBuffer::clear() {
    std::fill(data.begin(), data.end(), 0);
    data.clear();
    dataBitOffset = 0;
}

void BitBuffer::writeToFile(std::string filePath) {
    std::ofstream f(filePath, std::ios::out | std::ios::trunc | std::ios::binary);
    if (!f.is_open()) {
        std::cout << "Failed to open bitbuffer output file '" << filePath << "'" << std::endl;
        return;
    }
    f.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)((dataBitOffset + 7) / 8));
}

int BitBuffer::compare(const BitBuffer& other) const {
    return memcmp(data.data(), other.data.data(), (dataBitOffset + 7) / 8);
}
bool BitBuffer::operator==(const BitBuffer& o) const { return o.dataBitOffset == dataBitOffset && compare(o) == 0; }
bool BitBuffer::operator<(const BitBuffer& o) const { return o.dataBitOffset == dataBitOffset && compare(o) < 0; }
bool BitBuffer::operator>(const BitBuffer& o) const { return o.dataBitOffset == dataBitOffset && compare(o) > 0; }

std::ostream& operator<<(std::ostream& strm, const BitBuffer& b) {
    for (size_t i = 0; i < b.dataBitOffset; ++i) {
        strm << (b.getBit(i) ? "1" : "0");
        if ((i & 7) == 3) strm << " ";
        if ((i & 7) == 7) strm << "  ";
    }
    return strm;
}

// ---- HuffmanTable<byte> (HuffmanCodingByte.h) ------------------------------------------------
HuffmanTable<byte>::HuffmanTable(size_t maxLen) : codeMap(), maxCodewordLength(maxLen), codedSymbolCount(0) {
    memset(lengths, 0, sizeof lengths);
    memset(codes, 0, sizeof codes);
}

HuffmanTablePtr<byte> HuffmanTable<byte>::fromLengthsAndCodes(const uint8_t len[256], const uint16_t cod[256]) {
    auto t = std::make_shared<HuffmanTable<byte>>(16);
    memcpy(t->lengths, len, 256);
    memcpy(t->codes, cod, 512);
    for (int s = 0; s < 256; ++s) {
        if (len[s] == 0) continue;
        t->codedSymbolCount++;
        auto code = std::make_shared<BitBuffer>();
        const uint32_t aligned = (uint32_t)cod[s] << (32 - len[s]);
        const byte be[4] = {(byte)(aligned >> 24), (byte)(aligned >> 16), (byte)(aligned >> 8), (byte)aligned};
        code->pushBits(len[s], be);
        t->codeMap[s] = SymbolCodePair((byte)s, code);
    }
    return t;
}

HuffmanTablePtr<byte> HuffmanTable<byte>::create(size_t codeWordLength, const std::vector<byte>& srcData) {
    if (codeWordLength > 64) return nullptr;   // HuffmanCodingByte.h:105
    if (srcData.empty()) return std::make_shared<HuffmanTable<byte>>(codeWordLength);
    uint64_t counts[256] = {0};
    for (byte s : srcData) counts[s]++;
    uint8_t len[256];
    if (codeWordLength == 16) {   // the JPEG limit: the table the encoder and the DHT writer take
        uint16_t cod[256];
        if (jpeg_hip_huffman_from_histogram(counts, len, cod) != JPEG_HIP_OK) return nullptr;
        return fromLengthsAndCodes(len, cod);
    }
    // Any other limit the reference accepts (below 9 it returns a null table, :121).
    uint64_t wide[256];
    if (jpeg_hip_huffman_from_histogram_limit(counts, (int)codeWordLength, len, wide) != JPEG_HIP_OK) return nullptr;
    auto t = std::make_shared<HuffmanTable<byte>>(codeWordLength);
    bool jpegSized = true;
    for (int s = 0; s < 256; ++s) {
        if (len[s] == 0) continue;
        t->codedSymbolCount++;
        auto code = std::make_shared<BitBuffer>();
        const uint64_t aligned = wide[s] << (64 - len[s]);
        byte be[8];
        for (int k = 0; k < 8; ++k) be[k] = (byte)(aligned >> (56 - 8 * k));
        code->pushBits(len[s], be);
        t->codeMap[s] = SymbolCodePair((byte)s, code);
        if (len[s] > 16) jpegSized = false;
    }
    if (jpegSized) {   // usable wherever a 16-bit table is
        memcpy(t->lengths, len, 256);
        for (int s = 0; s < 256; ++s) t->codes[s] = (uint16_t)wide[s];
    }
    return t;
}

BitBufferPtr HuffmanTable<byte>::encode(const std::vector<byte>& srcData) {
    auto out = std::make_shared<BitBuffer>();
    for (byte b : srcData) out->pushBits(*codeMap[b].code);
    return out;
}
ConstBitBufferPtr HuffmanTable<byte>::encode(const byte& srcData) const { return codeMap[srcData].code; }

std::vector<byte> HuffmanTable<byte>::decode(BitBufferPtr in) {
    std::vector<byte> out;
    // Lengths and code bits from the code words themselves (a table made for another limit than
    // 16 may hold codes of up to 64 bits).
    size_t lens[256] = {0};
    uint64_t vals[256] = {0};
    for (int s = 0; s < 256; ++s) {
        if (!codeMap[s].code) continue;
        lens[s] = codeMap[s].code->getSize();
        for (size_t k = 0; k < lens[s]; ++k) vals[s] = (vals[s] << 1) | (codeMap[s].code->getBit(k) ? 1u : 0u);
    }
    uint64_t cur = 0;
    size_t n = 0;
    for (size_t i = 0; i < in->getSize(); ++i) {
        cur = (cur << 1) | (in->getBit(i) ? 1u : 0u);
        ++n;
        for (int s = 0; s < 256; ++s)
            if (lens[s] == n && vals[s] == cur) {
                out.push_back((byte)s);
                cur = 0;
                n = 0;
                break;
            }
    }
    return out;
}

std::ostream& operator<<(std::ostream& strm, const HuffmanTable<byte>& t) {
    for (const auto& p : t) strm << (int)p.symbol << ": " << *p.code << std::endl;
    return strm;
}

// ---- Image (Image.h:63-100, Image.cpp) -----------------------------------------------------
static size_t roundUp(size_t v, size_t step) { return (v + step - 1) / step * step; }

Image::Image(size_t width, size_t height, SamplingDefinition scheme)
    : imageSize(width, height),
      simulatedSize(roundUp(width, scheme.stepSize.width), roundUp(height, scheme.stepSize.height)),
      samplingScheme(scheme) {}

std::vector<float>& Image::floats() const {
    const size_t n = imageSize.width * imageSize.height * 3;
    if (pixels.size() != n) {
        pixels.assign(n, 0.0f);   // a new image is black
        for (size_t i = 0; i < pixels8.size(); ++i) pixels[i] = pixels8[i] / 255.0f;   // ImageLoader.cpp:190-193
    }
    return pixels;
}

void Image::setRawPixelData(float* rgb) {
    pixels8.clear();
    pixels.assign(rgb, rgb + imageSize.width * imageSize.height * 3);
}

void Image::setRawPixelData8(const byte* rgb) {
    pixels8.assign(rgb, rgb + imageSize.width * imageSize.height * 3);
    std::vector<float>().swap(pixels);
}

size_t Image::clampedIndex(size_t x, size_t y) const {
    x = std::min(x, imageSize.width - 1);    // Image.cpp:63-139: padding repeats the edge pixel
    y = std::min(y, imageSize.height - 1);
    return (y * imageSize.width + x) * 3;
}

void Image::getPixel(PixelData32& ref, size_t x, size_t y) const {
    const size_t i = clampedIndex(x, y);
    if (!pixels8.empty()) {
        ref.R = pixels8[i] / 255.0f;
        ref.G = pixels8[i + 1] / 255.0f;
        ref.B = pixels8[i + 2] / 255.0f;
        return;
    }
    const std::vector<float>& pixels = floats();
    ref.R = pixels[i];
    ref.G = pixels[i + 1];
    ref.B = pixels[i + 2];
}

void Image::setPixel(size_t x, size_t y, const PixelData32& c) {
    if (x >= imageSize.width || y >= imageSize.height) return;
    const size_t i = clampedIndex(x, y);
    std::vector<float>& pixels = floats();
    pixels[i] = c.R;
    pixels[i + 1] = c.G;
    pixels[i + 2] = c.B;
    pixels8.clear();
}

void Image::applySepia() {
    std::vector<float>& pixels = floats();
    for (size_t i = 0; i + 2 < pixels.size(); i += 3) {
        const float r = pixels[i], g = pixels[i + 1], b = pixels[i + 2];
        float t = 0.3930f * r;
        t = t + 0.7690f * g;
        t = t + 0.1890f * b;
        pixels[i] = t + 0.0f;   // the matrix's fourth column times alpha = 1
        t = 0.3490f * r;
        t = t + 0.6860f * g;
        t = t + 0.1680f * b;
        pixels[i + 1] = t + 0.0f;
        t = 0.2720f * r;
        t = t + 0.5340f * g;
        t = t + 0.1310f * b;
        pixels[i + 2] = t + 0.0f;
    }
    pixels8.clear();
}

std::vector<float> Image::getRawPixelData() {
    std::vector<float> out(imageSize.width * imageSize.height * 4);   // sized as the reference does; 3 used per pixel
    const std::vector<float>& pixels = floats();
    std::copy(pixels.begin(), pixels.end(), out.begin());
    return out;
}

std::vector<float> Image::getRawPixelDataSimulated() {
    std::vector<float> out(simulatedSize.width * simulatedSize.height * 4);
    const std::vector<float>& pixels = floats();
    size_t o = 0;
    for (size_t y = 0; y < simulatedSize.height; ++y)
        for (size_t x = 0; x < simulatedSize.width; ++x) {
            const size_t i = clampedIndex(x, y);
            out[o++] = pixels[i];
            out[o++] = pixels[i + 1];
            out[o++] = pixels[i + 2];
        }
    return out;
}

// ---- Encoder (Encoder.h:6-58, Encoder.cpp) ---------------------------------------------------
void Encoder::check(int status, const char* what) const {
    if (status == JPEG_HIP_OK) return;
    // The reference aborts on failure (JPEG-Encoder-SIMD.cpp:411-415); an exception carries
    // the reason instead.
    throw std::runtime_error(std::string(what) + ": " + jpeg_hip_status_string(status) + " (" +
                             (ctx ? jpeg_hip_last_error(ctx) : "no context") + ")");
}

void Encoder::open(const QTableSet& qtables) {
    ctx = nullptr;
    stages = 0;
    transformed = tablesReady = false;
    dctDone[0] = dctDone[1] = dctDone[2] = false;
    memset(&params, 0, sizeof params);
    params.width = (int32_t)image->getImageSize().width;
    params.height = (int32_t)image->getImageSize().height;
    image->getSamplingScheme().toAbi(params.sampling);
    for (int c = 0; c < 3; ++c) {
        tableCopies[c] = qtables[c];
        memcpy(params.qtable[c], tableCopies[c].floats, sizeof(float) * 64);
    }
    const char* dev = getenv("JPEGENC_DEVICE");
    check(jpeg_hip_create(dev ? atoi(dev) : 0, &ctx), "jpeg_hip_create");
}

Encoder::Encoder(const Image& img, const QTableSet& qtables) : image(std::make_shared<Image>(img)) { open(qtables); }
Encoder::Encoder(ImagePtr imagePtr, const QTableSet& qtables) : image(imagePtr) { open(qtables); }
Encoder::~Encoder() { jpeg_hip_destroy(ctx); }

// The three stage calls of EncodeJPEG (JPEG-Encoder-SIMD.cpp:421-433) are recorded; the
// fused kernel runs when the first channel's DCT is asked for.
void Encoder::convertToYCbCr() { stages |= JPEG_HIP_STAGE_YCBCR; }
void Encoder::reduceResolutionBySchema() { stages |= JPEG_HIP_STAGE_SUBSAMPLE; }

void Encoder::multiplyColorChannelBy(ColorChannelName channel, float val) {
    if (transformed) throw std::runtime_error("multiplyColorChannelBy is not supported after applyDCT");
    if (stages == 0) {   // the channels still are R, G, B
        pixelOps.push_back({JPEG_HIP_FILTER_MULTIPLY, (int)channel, val});
        return;
    }
    if (!(stages & JPEG_HIP_STAGE_YCBCR))
        throw std::runtime_error("multiplyColorChannelBy after reduceResolutionBySchema needs convertToYCbCr first");
    if (gainOps.size() >= JPEG_HIP_MAX_GAINS) throw std::runtime_error("too many multiplyColorChannelBy calls");
    gainOps.push_back({(int)channel, (stages & JPEG_HIP_STAGE_SUBSAMPLE) != 0, val});
}

void Encoder::applySepia() {
    if (stages != 0 || transformed) throw std::runtime_error("applySepia works on RGB pixels (before convertToYCbCr)");
    pixelOps.push_back({JPEG_HIP_FILTER_SEPIA, 0, 0.0f});
}

void Encoder::ensureTransformed() {
    if (transformed) return;
    check(jpeg_hip_begin(ctx, &params), "jpeg_hip_begin");
    if (image->hasBytePixels())
        check(jpeg_hip_upload_pixels(ctx, image->pixels8.data(), JPEG_HIP_RGB8, (int64_t)image->imageSize.width * 3),
              "jpeg_hip_upload_pixels");
    else
        check(jpeg_hip_upload_pixels(ctx, image->floats().data(), JPEG_HIP_RGBF32,
                                     (int64_t)image->imageSize.width * 3 * sizeof(float)),
              "jpeg_hip_upload_pixels");
    for (const PixelOp& op : pixelOps)
        check(jpeg_hip_filter_pixels(ctx, op.filter, op.channel, op.value), "jpeg_hip_filter_pixels");
    for (const GainOp& op : gainOps)
        check(jpeg_hip_channel_gain(ctx, op.channel, op.value, op.afterSubsampling ? 1 : 0), "jpeg_hip_channel_gain");
    check(jpeg_hip_transform(ctx, stages), "jpeg_hip_transform");
    transformed = true;
}

void Encoder::applyDCT(ColorChannelName channel) {
    ensureTransformed();
    dctDone[(int)channel] = true;
}

void Encoder::ensureTables() {
    if (tablesReady) return;
    if (!(dctDone[0] && dctDone[1] && dctDone[2]))
        throw std::runtime_error("Huffman tables need applyDCT on all three channels first");
    uint64_t hist[4][256];
    check(jpeg_hip_histograms_host(ctx, &hist[0][0]), "jpeg_hip_histograms_host");
    check(jpeg_hip_build_huffman(ctx, hist), "jpeg_hip_build_huffman");
    uint8_t len[4][256];
    uint16_t cod[4][256];
    check(jpeg_hip_get_huffman(ctx, len, cod), "jpeg_hip_get_huffman");
    // huffmanTables[channel (Y, Cb)][type (DC, AC)], Encoder.h:22; ABI order DC-Y, DC-C, AC-Y, AC-C.
    for (int ch = 0; ch < 2; ++ch)
        for (int type = 0; type < 2; ++type)
            huffmanTables[ch][type] = HuffmanTable<byte>::fromLengthsAndCodes(len[2 * type + ch], cod[2 * type + ch]);
    tablesReady = true;
}

HuffmanTablePtr<byte> Encoder::getHuffmanTable(CoefficientType type, ColorChannelName channel) {
    ensureTables();
    const int ch = (int)channel == Cr ? (int)Cb : (int)channel;   // Cb and Cr share tables, Encoder.cpp:420-424
    return huffmanTables[ch][(int)type];
}

std::vector<byte> Encoder::scanBytes() {
    ensureTables();
    int64_t n = 0;
    check(jpeg_hip_emit_scan_host(ctx, nullptr, 0, &n), "jpeg_hip_emit_scan_host");   // size query
    std::vector<byte> out((size_t)n);
    check(jpeg_hip_emit_scan_host(ctx, out.data(), n, &n), "jpeg_hip_emit_scan_host");
    return out;
}

void Encoder::serializeScanData(BitBuffer& bitBuffer) {
    const std::vector<byte> scan = scanBytes();
    bitBuffer.pushBits(scan.size() * 8, scan.data());   // already escaped and byte aligned
}

void Encoder::serialize(BitBuffer& bitBuffer) {
    ensureTables();
    uint8_t len[4][256];
    uint16_t cod[4][256];
    check(jpeg_hip_get_huffman(ctx, len, cod), "jpeg_hip_get_huffman");
    byte header[4096];
    const int n = jpeg_hip_write_headers(&params, len, cod, header, (int)sizeof header);
    if (n < 0) check(n, "jpeg_hip_write_headers");
    bitBuffer.pushBits((size_t)n * 8, header);   // SOI .. SOS, Encoder.cpp:473-483
    serializeScanData(bitBuffer);
    const byte eoi[2] = {0xff, 0xd9};
    bitBuffer.pushBits(16, eoi);
}

std::vector<int16_t> Encoder::getCoefficients(ColorChannelName channel) {
    ensureTransformed();
    jpeg_hip_layout lay;
    check(jpeg_hip_compute_layout(&params, &lay), "jpeg_hip_compute_layout");
    std::vector<int16_t> out((size_t)lay.blocks[(int)channel] * 64);
    check(jpeg_hip_get_coefficients(ctx, (int)channel, out.data(), (int64_t)out.size()), "jpeg_hip_get_coefficients");
    return out;
}

// ---- ImageLoader (ImageLoader.h, ImageLoader.cpp) ---------------------------------------------
std::string ImageLoader::fileExtension(const std::string& filename) {
    return filename.substr(filename.find_last_of(".") + 1);
}

ImagePtr ImageLoader::Load(const std::string& filename, SamplingDefinition scheme) {
    const std::string ext = fileExtension(filename);
    if (ext == "png") return LoadPNG(filename, scheme);
    if (ext == "ppm") return LoadPPM(filename, scheme);
    std::cout << "Failed to load image. Unknown file extension " << ext << std::endl;
    return nullptr;
}

void ImageLoader::Save(const std::string& filename, ImagePtr image) {
    const std::string ext = fileExtension(filename);
    if (ext == "png") return SavePNG(filename, image);
    if (ext == "ppm") return SavePPM(filename, image);
    std::cout << "Failed to save image. Unknown file extension " << ext << std::endl;
}

void ImageLoader::Save(const std::string& filename, EncoderPtr encoder) {
    const std::string ext = fileExtension(filename);
    if (ext == "jpg" || ext == "jpeg") return SaveJPG(filename, encoder);
    std::cout << "Invalid file name '" << filename << "'. Can't save encoder to non jpg file." << std::endl;
}

void ImageLoader::SaveJPG(std::string path, EncoderPtr encoder) {
    BitBuffer bitBuffer;
    encoder->serialize(bitBuffer);
    bitBuffer.writeToFile(path);
}

// Token reader for PNM headers and ASCII samples: skips white space and '#' comments.
static bool nextToken(std::istream& in, std::string& tok) {
    tok.clear();
    int c;
    while ((c = in.get()) != EOF) {
        if (c == '#') {
            while ((c = in.get()) != EOF && c != '\n') {}
        } else if (!isspace(c)) {
            tok.push_back((char)c);
            break;
        }
    }
    while ((c = in.peek()) != EOF && !isspace(c)) tok.push_back((char)in.get());
    return !tok.empty();
}

ImagePtr ImageLoader::LoadPPM(std::string path, SamplingDefinition scheme) {
    std::ifstream in(path, std::ios::binary);
    std::string tok;
    if (!in.is_open() || !nextToken(in, tok) || (tok != "P3" && tok != "P6")) {
        std::cout << "Failed to load ppm " << path << std::endl;
        return nullptr;
    }
    const bool ascii = tok == "P3";
    int width = 0, height = 0, maxValue = 0;
    if (nextToken(in, tok)) width = atoi(tok.c_str());
    if (nextToken(in, tok)) height = atoi(tok.c_str());
    if (nextToken(in, tok)) maxValue = atoi(tok.c_str());
    if (width <= 0 || height <= 0 || maxValue <= 0 || maxValue > 65535) {
        std::cout << "Failed to load ppm " << path << ": bad header" << std::endl;
        return nullptr;
    }
    const size_t n = (size_t)width * height * 3;
    ImagePtr image = std::make_shared<Image>((size_t)width, (size_t)height, scheme);
    if (!ascii && maxValue == 255) {   // the common case in one read: bytes as they are
        in.get();   // the single white-space byte after maxval
        std::vector<byte> bytes(n, 0);
        in.read(reinterpret_cast<char*>(bytes.data()), (std::streamsize)n);
        image->setRawPixelData8(bytes.data());
        return image;
    }
    std::vector<int> samples(n, 0);
    if (ascii) {
        for (size_t i = 0; i < n && nextToken(in, tok); ++i) samples[i] = atoi(tok.c_str());
    } else {
        in.get();
        for (size_t i = 0; i < n; ++i) {
            int v = in.get();
            if (maxValue > 255) v = (v << 8) | in.get();
            samples[i] = v < 0 ? 0 : v;
        }
    }
    if (maxValue == 255) {   // same arithmetic as value / 255.0f, done on the GPU
        std::vector<byte> bytes(n);
        for (size_t i = 0; i < n; ++i) bytes[i] = (byte)samples[i];
        image->setRawPixelData8(bytes.data());
    } else {                 // ImageLoader.cpp:126-128: stoi(value) / maxValue in float
        std::vector<float> f(n);
        const float mv = (float)maxValue;
        for (size_t i = 0; i < n; ++i) f[i] = samples[i] / mv;
        image->setRawPixelData(f.data());
    }
    return image;
}

void ImageLoader::SavePPM(std::string path, ImagePtr image) {
    std::ofstream out(path);
    const Dimension2D& size = image->getImageSize();
    out << "P3\n" << size.width << " " << size.height << "\n255\n";
    PixelData32 p;
    for (size_t y = 0; y < size.height; ++y) {
        for (size_t x = 0; x < size.width; ++x) {
            image->getPixel(p, x, y);
            const float ch[3] = {p.R, p.G, p.B};
            for (float v : ch) out << std::min(255, std::max(0, (int)(v * 255 + 0.5f))) << " ";   // ImageLoader.cpp:169-171
            out << " ";
        }
        out << "\n";
    }
}

// PNG reader on zlib: every colour type and bit depth of the format (grey 1/2/4/8/16, RGB 8/16,
// palette 1/2/4/8, grey+alpha and RGBA 8/16), plain or Adam7 interlaced.  Samples come out as the
// 8-bit RGB that lodepng::decode hands the reference (ImageLoader.cpp:180-193): the high byte of a
// 16-bit sample, value * 255 / (2^depth - 1) for a grey sample of fewer than 8 bits, alpha dropped.
static uint32_t be32(const byte* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

namespace {

struct PngFormat {
    int depth, colour, channels;
    const std::vector<byte>* palette;
    size_t lineBytes(uint32_t w) const { return ((size_t)w * channels * depth + 7) / 8; }
    size_t filterUnit() const { return (size_t)std::max(1, channels * depth / 8); }
};

// Undoes the line filters of one (sub-)image of w x h pixels at `raw` in place and writes its
// pixels to rgb at (x0 + x * dx, y0 + y * dy) of an image `width` pixels wide.  Returns the bytes used.
size_t pngPass(const PngFormat& f, byte* raw, uint32_t w, uint32_t h, byte* rgb, uint32_t width, uint32_t x0, uint32_t y0,
               uint32_t dx, uint32_t dy) {
    if (w == 0 || h == 0) return 0;
    const size_t stride = f.lineBytes(w), bpp = f.filterUnit();
    std::vector<byte> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        byte* cur = &raw[(stride + 1) * y + 1];
        const byte* prev = y > 0 ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        switch (cur[-1]) {
        case 1:
            for (size_t i = bpp; i < stride; ++i) cur[i] = (byte)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < stride; ++i) cur[i] = (byte)(cur[i] + prev[i]);
            break;
        case 3:
            for (size_t i = 0; i < bpp && i < stride; ++i) cur[i] = (byte)(cur[i] + (prev[i] >> 1));
            for (size_t i = bpp; i < stride; ++i) cur[i] = (byte)(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < bpp && i < stride; ++i) cur[i] = (byte)(cur[i] + prev[i]);   // a = c = 0: the predictor is b
            for (size_t i = bpp; i < stride; ++i) {
                const int a = cur[i - bpp], b = prev[i], c = prev[i - bpp];
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                cur[i] = (byte)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
            }
            break;
        default:
            break;
        }
        byte* line = &rgb[((size_t)(y0 + y * dy) * width + x0) * 3];
        if (f.colour == 2 && f.depth == 8 && dx == 1) {
            memcpy(line, cur, (size_t)w * 3);
            continue;
        }
        for (uint32_t x = 0; x < w; ++x) {
            byte* o = line + (size_t)x * dx * 3;
            unsigned v[4] = {0, 0, 0, 0};   // the pixel's samples, 8 bits each (or the palette index)
            if (f.depth >= 8) {
                const byte* src = &cur[(size_t)x * f.channels * (f.depth / 8)];
                for (int k = 0; k < f.channels; ++k) v[k] = src[(size_t)k * (f.depth / 8)];   // 16 bits: the high byte
            } else {
                const size_t bit = (size_t)x * f.depth;
                const unsigned raw1 = (cur[bit >> 3] >> (8 - f.depth - (bit & 7))) & ((1u << f.depth) - 1u);
                v[0] = f.colour == 3 ? raw1 : raw1 * 255u / ((1u << f.depth) - 1u);
            }
            if (f.colour == 3) {
                const size_t pi = (size_t)v[0] * 3;
                for (int k = 0; k < 3; ++k) o[k] = pi + k < f.palette->size() ? (*f.palette)[pi + k] : 0;
            } else if (f.colour == 2 || f.colour == 6) {
                o[0] = (byte)v[0]; o[1] = (byte)v[1]; o[2] = (byte)v[2];
            } else {
                o[0] = o[1] = o[2] = (byte)v[0];
            }
        }
    }
    return (stride + 1) * h;
}

}  // namespace

ImagePtr ImageLoader::LoadPNG(std::string path, SamplingDefinition scheme) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    std::vector<byte> file;
    if (in.is_open()) {
        file.resize((size_t)in.tellg());
        in.seekg(0);
        in.read(reinterpret_cast<char*>(file.data()), (std::streamsize)file.size());
    }
    static const byte sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) {
        std::cout << "Failed to decode png " << path << ": not a PNG file" << std::endl;
        return nullptr;
    }
    uint32_t width = 0, height = 0;
    int depth = 0, colour = 0, interlace = 0;
    std::vector<byte> idat, palette;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const std::string type(reinterpret_cast<const char*>(&file[pos + 4]), 4);
        const byte* body = &file[pos + 8];
        if (pos + 12 + len > file.size()) break;
        if (type == "IHDR" && len >= 13) {
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colour = body[9];
            interlace = body[12];
        } else if (type == "PLTE") {
            palette.assign(body, body + len);
        } else if (type == "IDAT") {
            idat.insert(idat.end(), body, body + len);
        } else if (type == "IEND") {
            break;
        }
        pos += 12 + len;
    }
    const int channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    // Depths the format allows per colour type (PNG specification, table 11.1).
    const bool depthOk = colour == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                         : colour == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                       : (depth == 8 || depth == 16);
    // (JPEG takes 65535 x 65535 at most: larger headers are refused before anything is sized by them.)
    if (width == 0 || height == 0 || width > 65535 || height > 65535 || channels == 0 || !depthOk || interlace > 1) {
        std::cout << "Failed to decode png " << path << ": unsupported format" << std::endl;
        return nullptr;
    }
    const PngFormat fmt = {depth, colour, channels, &palette};
    // Adam7: seven sub-images, each with its own lines and filter bytes (first pixel, step).
    static const uint32_t px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const uint32_t pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t total = 0;
    if (interlace) {
        for (int p = 0; p < 7; ++p) {
            const uint32_t w = (width + pdx[p] - 1 - px0[p]) / pdx[p], h = (height + pdy[p] - 1 - py0[p]) / pdy[p];
            if (w != 0 && h != 0) total += (fmt.lineBytes(w) + 1) * h;
        }
    } else {
        total = (fmt.lineBytes(width) + 1) * height;
    }
    std::vector<byte> raw(total);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) {
        std::cout << "Failed to decode png " << path << ": inflate error" << std::endl;
        return nullptr;
    }
    std::vector<byte> rgb((size_t)width * height * 3);
    if (interlace) {
        size_t at = 0;
        for (int p = 0; p < 7; ++p) {
            const uint32_t w = (width + pdx[p] - 1 - px0[p]) / pdx[p], h = (height + pdy[p] - 1 - py0[p]) / pdy[p];
            at += pngPass(fmt, raw.data() + at, w, h, rgb.data(), width, px0[p], py0[p], pdx[p], pdy[p]);
        }
    } else {
        pngPass(fmt, raw.data(), width, height, rgb.data(), width, 0, 0, 1, 1);
    }
    ImagePtr image = std::make_shared<Image>((size_t)width, (size_t)height, scheme);
    image->setRawPixelData8(rgb.data());
    return image;
}

static void pngChunk(std::ofstream& out, const char* type, const std::vector<byte>& body) {
    byte len[4] = {(byte)(body.size() >> 24), (byte)(body.size() >> 16), (byte)(body.size() >> 8), (byte)body.size()};
    out.write(reinterpret_cast<const char*>(len), 4);
    out.write(type, 4);
    if (!body.empty()) out.write(reinterpret_cast<const char*>(body.data()), (std::streamsize)body.size());
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
    if (!body.empty()) crc = crc32(crc, body.data(), (uInt)body.size());
    byte c[4] = {(byte)(crc >> 24), (byte)(crc >> 16), (byte)(crc >> 8), (byte)crc};
    out.write(reinterpret_cast<const char*>(c), 4);
}

void ImageLoader::SavePNG(std::string path, ImagePtr image) {
    const Dimension2D& size = image->getSimulatedSize();   // ImageLoader.cpp:200-215 writes the padded image
    std::vector<byte> raw((size.width * 3 + 1) * size.height);
    PixelData32 p;
    for (size_t y = 0; y < size.height; ++y) {
        byte* line = &raw[(size.width * 3 + 1) * y];
        line[0] = 0;
        for (size_t x = 0; x < size.width; ++x) {
            image->getPixel(p, x, y);
            const float ch[3] = {p.R, p.G, p.B};
            for (int k = 0; k < 3; ++k) line[1 + x * 3 + k] = (byte)std::min(255, std::max(0, (int)(ch[k] * 255 + 0.5f)));
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<byte> z(zlen);
    if (compress(z.data(), &zlen, raw.data(), (uLong)raw.size()) != Z_OK) {
        std::cout << "Failed to encode png " << path << std::endl;
        return;
    }
    z.resize(zlen);
    std::ofstream out(path, std::ios::binary);
    static const byte sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<byte> ihdr = {(byte)(size.width >> 24), (byte)(size.width >> 16), (byte)(size.width >> 8), (byte)size.width,
                              (byte)(size.height >> 24), (byte)(size.height >> 16), (byte)(size.height >> 8),
                              (byte)size.height, 8, 2, 0, 0, 0};
    pngChunk(out, "IHDR", ihdr);
    pngChunk(out, "IDAT", z);
    pngChunk(out, "IEND", {});
}
