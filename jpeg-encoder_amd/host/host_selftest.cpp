// CPU-only checks of the host classes, mirroring the reference's ad-hoc tests
// testHuffmanEncoding (JPEG-Encoder-SIMD.cpp:443-460) and bitBufferTest (:33-105).
// Prints one "name: value" line per check; tests/test_host_classes.py compares them with
// the reference's known answers.
#include <iostream>
#include <sstream>

#include "Encoder.h"
#include "ImageLoader.h"

static std::string bits(const BitBuffer& b) {
    std::string s;
    for (size_t i = 0; i < b.getSize(); ++i) s.push_back(b.getBit(i) ? '1' : '0');
    return s;
}
static std::string hex(const BitBuffer& b) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < (b.getSize() + 7) / 8; ++i) { s.push_back(d[b.bytes()[i] >> 4]); s.push_back(d[b.bytes()[i] & 15]); }
    return s;
}

int main(int argc, char** argv) {
    // host_selftest convert <image in> <image out>: ImageLoader::Load + Save on the host alone.
    if (argc == 4 && std::string(argv[1]) == "convert") {
        ImagePtr image = ImageLoader::Load(argv[2], Sampling::Scheme444);
        if (!image) return 1;
        ImageLoader::Save(argv[3], image);
        return 0;
    }
    // testHuffmanEncoding
    std::vector<byte> all{0, 1, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
                          6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6};
    auto table = HuffmanTable<byte>::create(16, all);
    for (const auto& p : *table) std::cout << "code" << (int)p.symbol << ": " << bits(*p.code) << "\n";
    std::vector<byte> probe{3, 0, 4};
    auto enc = table->encode(probe);
    std::cout << "encoded: " << bits(*enc) << "\n";
    auto dec = table->decode(enc);
    std::cout << "decoded:";
    for (byte b : dec) std::cout << " " << (int)b;
    std::cout << "\nsymbols: " << table->getSymbolCount() << "\n";

    // Other limits on the code length (HuffmanCodingByte.h:105, :121): a table of at most 9 bits
    // for a skewed source, round trip through it; limits the reference refuses give no table.
    {
        std::vector<byte> skew;
        for (int s = 0; s < 40; ++s)
            for (int k = 0; k < (1 << (s < 20 ? 20 - s : 0)) && k < 4096; ++k) skew.push_back((byte)s);
        auto t9 = HuffmanTable<byte>::create(9, skew);
        size_t longest = 0;
        for (const auto& p : *t9) longest = p.code->getSize() > longest ? p.code->getSize() : longest;
        std::vector<byte> msg{0, 39, 7, 20, 1};
        auto dec9 = t9->decode(t9->encode(msg));
        std::cout << "limit9: longest " << longest << " symbols " << t9->getSymbolCount() << " roundtrip "
                  << (dec9 == msg ? "ok" : "FAILED") << "\n";
        auto t40 = HuffmanTable<byte>::create(40, skew);
        auto dec40 = t40->decode(t40->encode(msg));
        std::cout << "limit40: roundtrip " << (dec40 == msg ? "ok" : "FAILED") << "\n";
        std::cout << "limit8: " << (HuffmanTable<byte>::create(8, skew) ? "table" : "null") << "\n";
        std::cout << "limit65: " << (HuffmanTable<byte>::create(65, skew) ? "table" : "null") << "\n";
    }

    // BitBuffer: escaped pushes, zero padding, unaligned getBits
    BitBuffer bb;
    const byte ones[2] = {0xff, 0xff};
    bb.pushBitsEscaped(3, ones);
    bb.pushBitsEscaped(13, ones);   // completes two 0xFF bytes -> two stuffed zeros
    const byte pat[2] = {0xa5, 0x80};
    bb.pushBits(9, pat);
    bb.fillToByteBorder();
    std::cout << "bitbuffer: " << hex(bb) << " size " << bb.getSize() << "\n";
    byte out[2] = {0, 0};
    bb.getBits(33, out, 9);
    std::cout << "getbits: " << std::hex << (int)out[0] << " " << (int)out[1] << std::dec << "\n";
    BEushort be(0x1234);
    std::cout << "beushort: " << std::hex << (int)reinterpret_cast<byte*>(&be)[0] << (int)reinterpret_cast<byte*>(&be)[1]
              << std::dec << " " << (unsigned short)be << "\n";

    // Sampling definitions and padding
    Image img(1920, 1080, Sampling::Scheme420);
    std::cout << "sim420: " << img.getSimulatedSize().width << "x" << img.getSimulatedSize().height << " step "
              << Sampling::Scheme420.stepSize.width << "x" << Sampling::Scheme420.stepSize.height << " inv "
              << Sampling::Scheme420.inverseFactor[0].width << Sampling::Scheme420.inverseFactor[0].height
              << Sampling::Scheme420.inverseFactor[1].width << Sampling::Scheme420.inverseFactor[1].height << "\n";
    std::cout << "step411: " << Sampling::Scheme411.stepSize.width << "x" << Sampling::Scheme411.stepSize.height << "\n";
    std::cout << "step321: " << Sampling::Scheme321.stepSize.width << "x" << Sampling::Scheme321.stepSize.height << "\n";
    Image small(3, 2, Sampling::Scheme444);
    float px[18] = {0, .1f, .2f, .3f, .4f, .5f, .6f, .7f, .8f, .9f, 1, 0, .25f, .5f, .75f, 1, 0, 1};
    small.setRawPixelData(px);
    PixelData32 p;
    small.getPixel(p, 7, 7);   // outside: nearest edge pixel
    std::cout << "edge: " << p.R << " " << p.G << " " << p.B << "\n";
    std::cout << "qlum0: " << JPEGQuantization::luminance.floats[0] << " q90: "
              << JPEGQuantization::scaledLuminance(90).floats[0] << " " << JPEGQuantization::scaledChrominance(90).floats[63]
              << "\n";
    return 0;
}
