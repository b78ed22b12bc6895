// Verification decoder of libjpegenc_hip: host-side stream decoding and the launch
// interface of the inverse kernels.  Not part of the public ABI.
#ifndef JPEGHIP_DECODE_H
#define JPEGHIP_DECODE_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "jpegenc_hip.h"

namespace jpeghip {

// What the headers of a baseline file say, and (after decode_entropy) its coefficients.
struct DecodedStream {
    struct Dht {
        uint8_t bits[16];
        uint8_t values[256];
        int count = 0;
    };
    int width = 0, height = 0;
    int componentId[3], hSamp[3], vSamp[3], tq[3], td[3], ta[3];
    uint8_t qtables[4][64];        // zig-zag order, as stored in DQT
    Dht dht[4];                    // DC 0, DC 1, AC 0, AC 1
    int restartInterval = 0;
    int64_t scanOffset = 0;        // first entropy-coded byte
    int mcuW = 0, mcuH = 0, mcusPerRow = 0, mcuRows = 0, blocksPerMcu = 0;
    int chanOffset[3];
    int64_t totalBlocks = 0;
    std::vector<int16_t> coef;     // blocks in scan order, 64 zig-zag ordered values each
};

int parse_headers(const uint8_t* data, int64_t len, DecodedStream& out, std::string& err);
int decode_entropy(const uint8_t* data, int64_t len, DecodedStream& out, std::string& err);

// Geometry of the inverse kernels.
struct DGeom {
    int width, height;         // real image size
    int mcusPerRow, mcuRows, bpm;
    int off1, off2;            // first Cb / Cr block inside an MCU
    int hS[3], vS[3];          // blocks per MCU across / down
    int wf[3], hf[3];          // pixels per sample across / down
    int planeW[3], planeH[3];  // samples per channel plane
    long long planeOffset[3];  // floats from the start of the plane buffer
    long long totalBlocks;
};

// Dequantization + inverse DCT of every block into float sample planes (values 0..1).
// d_q: float[3][64] divisors in zig-zag order, d_cos: float[8][8] cosine table.
hipError_t launch_inverse_blocks(const DGeom& g, const int16_t* d_coef, const float* d_q, const float* d_cos,
                                 float* d_planes, hipStream_t s);
// Sample planes -> interleaved RGB8 (sample replication, YCbCr -> RGB, rounding to bytes).
hipError_t launch_planes_to_rgb(const DGeom& g, const float* d_planes, uint8_t* d_rgb, long long strideBytes,
                                hipStream_t s);
// d_sum[0] += sum over all samples of (a - b)^2 for two RGB8 images; d_sum is cleared first.
hipError_t launch_squared_error(const uint8_t* d_a, long long strideA, const uint8_t* d_b, long long strideB, int width,
                                int height, unsigned long long* d_sum, hipStream_t s);

// Rows y0.. of `count` synthetic images (kind 0 noise, 1 natural-like; seeds seed, seed + 1, ...).
hipError_t launch_synthetic_images(unsigned char* d_out, int kind, int width, int rows, int y0, unsigned seed,
                                   long long strideBytes, long long imageStrideBytes, int count, hipStream_t s);

}  // namespace jpeghip

#endif
