// Internal declarations shared by the host and device translation units of
// libjpegenc_hip.  Not part of the public ABI.
#ifndef JPEGHIP_INTERNAL_H
#define JPEGHIP_INTERNAL_H

#include <cstdint>

#include "jpegenc_hip.h"

namespace jpeghip {

// Sizes derived from a parameter block (Image::Image, Image.cpp:6-16; SamplingDefinition,
// SamplingScheme.h:32-43; Encoder::createBlocks, Encoder.cpp:168-227).
struct Geometry {
    int width, height;          // real image size
    int simW, simH;             // padded ("simulated") size
    int stepW, stepH;           // MCU size in pixels
    int mcusPerRow, mcuRows;    // whole image
    int bandRow0, bandRows;     // MCU rows encoded by this context
    int wf[3], wm[3], hf[3], hm[3];
    int hFactor[3], vFactor[3];
    int blocksPerMcu;
    int chanOffset[3];          // first block of each channel inside an MCU
    int chanBlocks[3];          // blocks of each channel inside an MCU
    int64_t nMcu;               // MCUs in the band
    int64_t totalBlocks;
    int restartInterval;        // MCUs, 0 = none
    int64_t intervals;          // restart intervals in the band
    int64_t firstInterval;      // index of the band's first interval in the whole image
    bool startsImage, endsImage;
    bool integerDivisors;       // every quantization divisor is an integer 1..255: the reciprocal form of the division is exact
};

// Fills `g`; returns a JPEG_HIP_* status.
int make_geometry(const jpeg_hip_params& p, Geometry& g);

const int* zigzag_order();
// ITU-T T.81 Annex K.3 typical tables (K.3-K.6) as [DC-Y, DC-C, AC-Y, AC-C], canonical codes (Annex C).
void standard_huffman(uint8_t lengths[4][256], uint16_t codes[4][256]);
void huffman_from_histogram(const uint64_t counts[256], uint8_t lengths[256], uint16_t codes[256]);
// The same for any limit on the code length (HuffmanTable<byte>::create's first argument, 9..64).
void huffman_from_histogram_limit(const uint64_t counts[256], int maxLen, uint8_t lengths[256], uint64_t codes[256]);
bool huffman_table_is_sane(const uint8_t lengths[256], const uint16_t codes[256]);
int write_dht(const uint8_t lengths[256], const uint16_t codes[256], int tableId, int coefType, uint8_t* out,
              int capacity);
int write_headers(const jpeg_hip_params& prm, const Geometry& g, const uint8_t lengths[4][256],
                  const uint16_t codes[4][256], uint8_t* out, int capacity);

// Worst-case size of the unstuffed entropy-coded data of `blocks` 8x8 blocks: every
// coefficient costs at most a 16-bit code and 11 amplitude bits.
inline int64_t worst_case_scan_bytes(int64_t blocks) { return blocks * 216 + 64; }

}  // namespace jpeghip

#endif
