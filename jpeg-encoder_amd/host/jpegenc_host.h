// Host-side C++ mirror of the reference encoder's class surface (Irame/JPEG-Encoder,
// tree "JPEG-Encoder-SIMD/"): the same type and method names with the same meaning, so
// that a driver written against the reference -- EncodeJPEG, JPEG-Encoder-SIMD.cpp:397-441
// -- builds against this header unchanged.  Behind the methods sits the C ABI of
// libjpegenc_hip (include/jpegenc_hip.h); nothing here computes pixels on the CPU.
//
// Reference interfaces mirrored: Dimension2D.h, ColorNames.h, SamplingScheme.h:4-94,
// QuantizationTables.h:6-110, BitBuffer.h:5-93, HuffmanCodingByte.h:9-100, Image.h:15-100,
// Encoder.h:6-58, ImageLoader.h:6-26.
#ifndef JPEGENC_HOST_H
#define JPEGENC_HOST_H

#include <array>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "jpegenc_hip.h"

typedef unsigned char byte;
typedef unsigned int uint;

// ---- Dimension2D.h -----------------------------------------------------------------------
struct Dimension2D {
    size_t width, height;
    Dimension2D() : width(0), height(0) {}
    constexpr Dimension2D(size_t w, size_t h) : width(w), height(h) {}
    bool operator==(const Dimension2D& o) const { return width == o.width && height == o.height; }
};

// ---- ColorNames.h ------------------------------------------------------------------------
enum RGBColorName { R = 0, G = 1, B = 2 };
enum YCbCrColorName { Y = 0, Cb = 1, Cr = 2 };
union ColorChannelName {
    int index;
    RGBColorName rgbColorName;
    YCbCrColorName yCbCrColorName;
    ColorChannelName() : index(0) {}
    ColorChannelName(int i) : index(i) {}
    operator int() const { return index; }
};

// ---- SamplingScheme.h --------------------------------------------------------------------
enum ReductionMethod { Subsampling = JPEG_HIP_SUBSAMPLING, Average = JPEG_HIP_AVERAGE };

struct ChannelReductionOptions {
    int widthFactor;
    ReductionMethod widthMethod;
    int heightFactor;
    ReductionMethod heightMethod;
    constexpr ChannelReductionOptions(int w, ReductionMethod wm, int h, ReductionMethod hm)
        : widthFactor(w), widthMethod(wm), heightFactor(h), heightMethod(hm) {}
};

class SamplingDefinition {
    static constexpr int gcdOf(int a, int b) { return b == 0 ? a : gcdOf(b, a % b); }
    static constexpr int lcmOf(int a, int b) { return a / gcdOf(a, b) * b; }
    static constexpr int lcm3(int a, int b, int c) { return lcmOf(lcmOf(a, b), c); }

public:
    ChannelReductionOptions reductionOptions[3];
    Dimension2D stepSize;          // MCU size in pixels
    Dimension2D inverseFactor[3];  // JPEG H/V sampling factors
    constexpr SamplingDefinition(ChannelReductionOptions y, ChannelReductionOptions cb, ChannelReductionOptions cr)
        : reductionOptions{y, cb, cr},
          stepSize(lcm3(y.widthFactor, cb.widthFactor, cr.widthFactor) * 8,
                   lcm3(y.heightFactor, cb.heightFactor, cr.heightFactor) * 8),
          inverseFactor{Dimension2D(stepSize.width / y.widthFactor / 8, stepSize.height / y.heightFactor / 8),
                        Dimension2D(stepSize.width / cb.widthFactor / 8, stepSize.height / cb.heightFactor / 8),
                        Dimension2D(stepSize.width / cr.widthFactor / 8, stepSize.height / cr.heightFactor / 8)} {}
    // The twelve integers the C ABI takes (jpeg_hip_params::sampling).
    void toAbi(int32_t out[12]) const;
};

namespace Sampling {
#define JPEGENC_FULL {1, Subsampling, 1, Subsampling}
const SamplingDefinition Scheme444(JPEGENC_FULL, JPEGENC_FULL, JPEGENC_FULL);
const SamplingDefinition Scheme422(JPEGENC_FULL, {2, Subsampling, 1, Subsampling}, {2, Subsampling, 1, Subsampling});
const SamplingDefinition Scheme411(JPEGENC_FULL, {4, Subsampling, 1, Subsampling}, {4, Subsampling, 1, Subsampling});
const SamplingDefinition Scheme420(JPEGENC_FULL, {2, Average, 2, Average}, {2, Average, 2, Average});
const SamplingDefinition Scheme422Average(JPEGENC_FULL, {2, Average, 1, Subsampling}, {2, Average, 1, Subsampling});
const SamplingDefinition Scheme422Height(JPEGENC_FULL, {1, Subsampling, 2, Subsampling}, {1, Subsampling, 2, Subsampling});
const SamplingDefinition Scheme422HeightAverage(JPEGENC_FULL, {1, Subsampling, 2, Average}, {1, Subsampling, 2, Average});
const SamplingDefinition Scheme311(JPEGENC_FULL, {3, Subsampling, 1, Subsampling}, {3, Subsampling, 1, Subsampling});
const SamplingDefinition Scheme321(JPEGENC_FULL, {3, Subsampling, 2, Subsampling}, {3, Subsampling, 2, Subsampling});
#undef JPEGENC_FULL
}  // namespace Sampling

// ---- QuantizationTables.h ----------------------------------------------------------------
struct QTable {
    float floats[64];  // natural (row-major) order
};

struct QTableSet {
    const QTable* tables[3];
    QTableSet(const QTable& luminance, const QTable& chrominance) : tables{&luminance, &chrominance, &chrominance} {}
    QTableSet(const QTable& luminance, const QTable& chrominance, const QTable& chrominance2)
        : tables{&luminance, &chrominance, &chrominance2} {}
    const QTable& operator[](size_t index) const { return *tables[index]; }
};

namespace JPEGQuantization {
extern const QTable luminance;    // Annex K.1
extern const QTable chrominance;  // Annex K.2
namespace Photoshop {
extern const QTable luminance12, chrominance12, luminance5, chrominance5;
}
// Annex-K tables scaled with the IJG quality rule (the reference has fixed tables only).
QTable scaledLuminance(int quality);
QTable scaledChrominance(int quality);
}  // namespace JPEGQuantization

// ---- stdafx.h:31-47 ------------------------------------------------------------------------
struct BEushort {  // big-endian 16-bit value for serialization
    BEushort() : value(0) {}
    BEushort(unsigned short v) : value((unsigned short)((v >> 8) | (v << 8))) {}
    operator unsigned short() const { return (unsigned short)((value >> 8) | (value << 8)); }

private:
    unsigned short value;
};

// ---- BitBuffer.h ---------------------------------------------------------------------------
class BitBuffer {
    std::vector<byte> data;
    size_t dataBitOffset;
    void appendBit(bool bit, bool escape);
    void pushBitsImpl(size_t numOfBits, const void* buffer, size_t offset, bool escape);

public:
    explicit BitBuffer(size_t initialBufferSizeInBit = 0);
    size_t getSize() const { return dataBitOffset; }
    size_t getCapacity() const { return data.capacity() * 8; }
    void fillToByteBorder();  // pads with zero bits, as the reference does
    void pushBit(bool val);
    void pushBits(const BitBuffer& buffer);
    void pushBits(size_t numOfBits, const void* buffer, size_t offset);
    void pushBits(size_t numOfBits, const void* buffer);
    void pushBitsEscaped(const BitBuffer& buffer);  // a 0x00 follows every completed 0xFF byte
    void pushBitsEscaped(size_t numOfBits, const void* buffer);
    void pushBitsEscaped(size_t numOfBits, const void* buffer, size_t offset);
    bool getBit(size_t index) const;
    void getBits(size_t index, void* out, size_t numOfBits) const;
    void clear();
    void writeToFile(std::string file);
    bool operator==(const BitBuffer& other) const;
    bool operator<(const BitBuffer& other) const;
    bool operator>(const BitBuffer& other) const;
    int compare(const BitBuffer& other) const;
    const byte* bytes() const { return data.data(); }  // first getSize()/8 rounded-up bytes are valid
    friend std::ostream& operator<<(std::ostream& strm, const BitBuffer& bitBuffer);
    template <typename T>
    void push(T& value) { pushBits(sizeof(T) * 8, &value); }
};
typedef std::shared_ptr<BitBuffer> BitBufferPtr;
typedef std::shared_ptr<const BitBuffer> ConstBitBufferPtr;

// ---- HuffmanCoding.h / HuffmanCodingByte.h --------------------------------------------------
template <typename SymbolType>
class HuffmanTable;
template <typename SymbolType>
using HuffmanTablePtr = std::shared_ptr<HuffmanTable<SymbolType>>;

template <>
class HuffmanTable<byte> {
public:
    struct SymbolCodePair {
        byte symbol;
        BitBufferPtr code;
        SymbolCodePair() : symbol(0) {}
        SymbolCodePair(byte s, BitBufferPtr c) : symbol(s), code(c) {}
    };
    struct const_iterator {  // skips symbols without a code
        const SymbolCodePair* cur;
        const SymbolCodePair* end;
        const_iterator(const SymbolCodePair* c, const SymbolCodePair* e) : cur(c), end(e) { skip(); }
        void skip() { while (cur != end && !cur->code) ++cur; }
        const SymbolCodePair& operator*() const { return *cur; }
        bool operator==(const const_iterator& o) const { return cur == o.cur; }
        bool operator!=(const const_iterator& o) const { return cur != o.cur; }
        const_iterator& operator++() { ++cur; skip(); return *this; }
    };

    explicit HuffmanTable(size_t maxCodewordLength);
    std::array<SymbolCodePair, 256> codeMap;
    size_t maxCodewordLength;
    size_t codedSymbolCount;
    uint8_t lengths[256];  // code length per symbol (0 = unused); the C ABI's representation
    uint16_t codes[256];   // right-aligned code bits

    size_t getSymbolCount() const { return codedSymbolCount; }
    const_iterator begin() const { return const_iterator(codeMap.data(), codeMap.data() + 256); }
    const_iterator end() const { return const_iterator(codeMap.data() + 256, codeMap.data() + 256); }

    // Package-merge table with a reserved all-ones code (HuffmanCodingByte.h:102-210) for every limit the
    // reference accepts: 16, the JPEG limit, through jpeg_hip_huffman_from_histogram; 9..64 through
    // jpeg_hip_huffman_from_histogram_limit; above 64 and below 9 a null table, as there (:105, :121).
    static HuffmanTablePtr<byte> create(size_t codeWordLength, const std::vector<byte>& srcData);
    static HuffmanTablePtr<byte> fromLengthsAndCodes(const uint8_t lengths[256], const uint16_t codes[256]);
    BitBufferPtr encode(const std::vector<byte>& srcData);
    ConstBitBufferPtr encode(const byte& srcData) const;
    std::vector<byte> decode(BitBufferPtr inputStream);
    friend std::ostream& operator<<(std::ostream& strm, const HuffmanTable<byte>& t);
};

// ---- Image.h -------------------------------------------------------------------------------
struct PixelData32 {
    float R, G, B;
    PixelData32() : R(0), G(0), B(0) {}
    PixelData32(float r, float g, float b) : R(r), G(g), B(b) {}
};

class Encoder;
class Image {
    friend class Encoder;
    const Dimension2D imageSize;
    const Dimension2D simulatedSize;  // padded to whole MCUs
    const SamplingDefinition samplingScheme;
    // Interleaved RGB, real size.  An image set from bytes keeps the bytes (the encoder sends
    // those and the GPU divides by 255.0f); the floats in [0,1] are made when somebody asks
    // for them.  Anything that changes a pixel works on the floats and drops the bytes.
    mutable std::vector<float> pixels;
    std::vector<byte> pixels8;
    size_t clampedIndex(size_t x, size_t y) const;
    std::vector<float>& floats() const;

public:
    Image(size_t width, size_t height, SamplingDefinition scheme);
    Image(const Image& other) = default;
    const Dimension2D& getImageSize() const { return imageSize; }
    const Dimension2D& getSimulatedSize() const { return simulatedSize; }
    const SamplingDefinition& getSamplingScheme() const { return samplingScheme; }
    // Interleaved RGB floats in [0,1], three per pixel, width*height pixels.
    void setRawPixelData(float* rgbData);
    // Interleaved RGB bytes; the encoder divides by 255.0f as ImageLoader::LoadPNG does.
    void setRawPixelData8(const byte* rgbData);
    std::vector<float> getRawPixelData();
    std::vector<float> getRawPixelDataSimulated();
    void setPixel(size_t x, size_t y, const PixelData32& color);
    void getPixel(PixelData32& ref, size_t x, size_t y) const;  // outside the real image: nearest edge pixel
    bool hasBytePixels() const { return !pixels8.empty(); }
    // Image.cpp:212-219 with the matrix and operation order of applySepiaFilterAVXImpl
    // (SIMD256.h:218-244), on the host copy of the pixels.
    void applySepia();
};
typedef std::shared_ptr<Image> ImagePtr;

// ---- Encoder.h -----------------------------------------------------------------------------
class Encoder {
public:
    enum CoefficientType { DC = JPEG_HIP_DC, AC = JPEG_HIP_AC };

private:
    ImagePtr image;
    QTable tableCopies[3];
    jpeg_hip_ctx* ctx;
    jpeg_hip_params params;
    int stages;            // JPEG_HIP_STAGE_* bits requested so far
    bool transformed;
    bool dctDone[3];
    bool tablesReady;
    HuffmanTablePtr<byte> huffmanTables[2][2];
    struct PixelOp {
        int filter, channel;
        float value;
    };
    std::vector<PixelOp> pixelOps;   // applied on the device, in order, after the upload
    struct GainOp {
        int channel;
        bool afterSubsampling;
        float value;
    };
    std::vector<GainOp> gainOps;     // multiplyColorChannelBy calls after convertToYCbCr, for the fused transform
    void open(const QTableSet& qtables);
    void ensureTransformed();
    void ensureTables();
    void check(int status, const char* what) const;
    std::vector<byte> scanBytes();

public:
    Encoder(const Image& image, const QTableSet& qtables);  // encodes a private copy
    Encoder(ImagePtr imagePtr, const QTableSet& qtables);   // shares the image
    ~Encoder();
    Encoder(const Encoder&) = delete;
    Encoder& operator=(const Encoder&) = delete;

    void convertToYCbCr();
    // Encoder.cpp:35-38.  Supported while the channels still are R, G, B, i.e. before
    // convertToYCbCr; runs on the GPU right after the pixels are uploaded.
    void multiplyColorChannelBy(ColorChannelName colorChannel, float val);
    void reduceResolutionBySchema();
    void applyDCT(ColorChannelName colorChannelName);
    HuffmanTablePtr<byte> getHuffmanTable(CoefficientType type, ColorChannelName channelName);
    void serialize(BitBuffer& bitBuffer);
    void serializeScanData(BitBuffer& bitBuffer);

    // Extensions without a counterpart in the reference.
    void applySepia();   // Image::applySepia on the encoder's device copy of the pixels
    void setStandardHuffmanTables(bool on) { params.standard_huffman = on ? 1 : 0; }
    void setRestartInterval(int mcus) { params.restart_interval = mcus; }
    void setPadWithOnes(bool on) { params.pad_ones = on ? 1 : 0; }
    // Quantized coefficients of one channel: blocks in MCU order, 64 zig-zag values each.
    std::vector<int16_t> getCoefficients(ColorChannelName channelName);
};
typedef std::shared_ptr<Encoder> EncoderPtr;

// ---- ImageLoader.h -------------------------------------------------------------------------
class ImageLoader {
public:
    static ImagePtr Load(const std::string& filename, SamplingDefinition scheme);
    static void Save(const std::string& filename, ImagePtr image);
    static void Save(const std::string& filename, EncoderPtr encoder);
    static ImagePtr LoadPPM(std::string path, SamplingDefinition scheme);  // ASCII P3 (and binary P6)
    static ImagePtr LoadPNG(std::string path, SamplingDefinition scheme);  // every colour type and bit depth, plain or interlaced
    static void SavePPM(std::string path, ImagePtr image);
    static void SavePNG(std::string path, ImagePtr image);
    static void SaveJPG(std::string path, EncoderPtr encoder);

private:
    static std::string fileExtension(const std::string& filename);
};

#endif  // JPEGENC_HOST_H
