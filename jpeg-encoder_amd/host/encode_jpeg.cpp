// Command line driver that replays the reference's EncodeJPEG call sequence
// (JPEG-Encoder-SIMD.cpp:397-441) on the GPU-backed classes:
//   encode_jpeg <source .ppm/.png> <destination .jpg> [444|422|411|420] [quality 1..100]
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "Encoder.h"
#include "ImageLoader.h"

static void EncodeJPEG(const std::string& srcFile, const std::string& dstFile, const SamplingDefinition& scheme,
                       const QTableSet& qtables) {
    ImagePtr image = ImageLoader::Load(srcFile, scheme);
    if (!image) {
        std::cout << "Failed to load source image " << srcFile << std::endl;
        exit(2);
    }
    EncoderPtr encoder = std::make_shared<Encoder>(image, qtables);
    encoder->convertToYCbCr();
    encoder->reduceResolutionBySchema();
    encoder->applyDCT(YCbCrColorName::Y);
    encoder->applyDCT(YCbCrColorName::Cb);
    encoder->applyDCT(YCbCrColorName::Cr);
    ImageLoader::Save(dstFile, encoder);
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
    try {
        if (argc > 4) {
            const QTable lum = JPEGQuantization::scaledLuminance(atoi(argv[4]));
            const QTable chr = JPEGQuantization::scaledChrominance(atoi(argv[4]));
            EncodeJPEG(argv[1], argv[2], scheme, QTableSet(lum, chr));
        } else {
            EncodeJPEG(argv[1], argv[2], scheme, QTableSet(JPEGQuantization::luminance, JPEGQuantization::chrominance));
        }
    } catch (const std::exception& e) {
        std::cerr << "encode failed: " << e.what() << std::endl;
        return 3;
    }
    return 0;
}
