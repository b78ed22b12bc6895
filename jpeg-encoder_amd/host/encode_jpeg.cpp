// Command line driver that replays the reference's EncodeJPEG call sequence
// (JPEG-Encoder-SIMD.cpp:397-441) on the GPU-backed classes:
//   encode_jpeg <source .ppm/.png> <destination .jpg> [444|422|411|420] [quality 1..100 | 0 = Annex K tables]
//               [options: sepia, gain=<channel 0..2>:<factor> (on R, G, B),
//                ycbcr-gain=<channel>:<factor> (after convertToYCbCr), reduced-gain=<channel>:<factor>
//                (after reduceResolutionBySchema), standard-huffman, restart=<MCUs>, timing]
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "Encoder.h"
#include "ImageLoader.h"

struct Options {
    bool sepia = false, standardHuffman = false, timing = false;
    int gainChannel = -1, ycbcrGainChannel = -1, reducedGainChannel = -1, restart = 0;
    float gain = 1.0f, ycbcrGain = 1.0f, reducedGain = 1.0f;
};

static void EncodeJPEG(const std::string& srcFile, const std::string& dstFile, const SamplingDefinition& scheme,
                       const QTableSet& qtables, const Options& opt) {
    auto last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {   // the reference's driver prints such a line per step (JPEG-Encoder-SIMD.cpp:401-437)
        const auto now = std::chrono::steady_clock::now();
        if (opt.timing) std::cerr << what << ": " << std::chrono::duration<double, std::milli>(now - last).count() << " ms" << std::endl;
        last = now;
    };
    ImagePtr image = ImageLoader::Load(srcFile, scheme);
    lap("load");
    if (!image) {
        std::cout << "Failed to load source image " << srcFile << std::endl;
        exit(2);
    }
    EncoderPtr encoder = std::make_shared<Encoder>(image, qtables);
    lap("encoder (device context)");
    if (opt.sepia) encoder->applySepia();
    if (opt.gainChannel >= 0) encoder->multiplyColorChannelBy((ColorChannelName)opt.gainChannel, opt.gain);
    encoder->setStandardHuffmanTables(opt.standardHuffman);
    encoder->setRestartInterval(opt.restart);
    encoder->convertToYCbCr();
    if (opt.ycbcrGainChannel >= 0) encoder->multiplyColorChannelBy((ColorChannelName)opt.ycbcrGainChannel, opt.ycbcrGain);
    encoder->reduceResolutionBySchema();
    if (opt.reducedGainChannel >= 0)
        encoder->multiplyColorChannelBy((ColorChannelName)opt.reducedGainChannel, opt.reducedGain);
    encoder->applyDCT(YCbCrColorName::Y);
    encoder->applyDCT(YCbCrColorName::Cb);
    encoder->applyDCT(YCbCrColorName::Cr);
    lap("convert, reduce, transform");
    ImageLoader::Save(dstFile, encoder);
    lap("tables, scan, file");
}

int main(int argc, char* argv[]) {
    if (argc < 3) {
        std::cerr << "Usage: " << argv[0] << " <Source File> <Destination File> [444|422|411|420] [quality]" << std::endl;
        return 1;
    }
    const std::string name = argc > 3 ? argv[3] : "420";
    const SamplingDefinition& scheme = name == "444"   ? Sampling::Scheme444
                                       : name == "422" ? Sampling::Scheme422
                                       : name == "411" ? Sampling::Scheme411
                                                       : Sampling::Scheme420;
    Options opt;
    for (int i = 5; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "sepia") opt.sepia = true;
        else if (a == "standard-huffman") opt.standardHuffman = true;
        else if (a == "timing") opt.timing = true;
        else if (a.rfind("restart=", 0) == 0) opt.restart = atoi(a.c_str() + 8);
        else if (a.rfind("gain=", 0) == 0 && a.size() > 7 && a[6] == ':') {
            opt.gainChannel = a[5] - '0';
            opt.gain = (float)atof(a.c_str() + 7);
        } else if (a.rfind("ycbcr-gain=", 0) == 0 && a.size() > 13 && a[12] == ':') {
            opt.ycbcrGainChannel = a[11] - '0';
            opt.ycbcrGain = (float)atof(a.c_str() + 13);
        } else if (a.rfind("reduced-gain=", 0) == 0 && a.size() > 15 && a[14] == ':') {
            opt.reducedGainChannel = a[13] - '0';
            opt.reducedGain = (float)atof(a.c_str() + 15);
        } else {
            std::cerr << "unknown option " << a << std::endl;
            return 1;
        }
    }
    try {
        if (argc > 4 && atoi(argv[4]) > 0) {
            const QTable lum = JPEGQuantization::scaledLuminance(atoi(argv[4]));
            const QTable chr = JPEGQuantization::scaledChrominance(atoi(argv[4]));
            EncodeJPEG(argv[1], argv[2], scheme, QTableSet(lum, chr), opt);
        } else {
            EncodeJPEG(argv[1], argv[2], scheme, QTableSet(JPEGQuantization::luminance, JPEGQuantization::chrominance), opt);
        }
    } catch (const std::exception& e) {
        std::cerr << "encode failed: " << e.what() << std::endl;
        return 3;
    }
    return 0;
}
