// Launch interface between the host C ABI and the device kernels of libjpegenc_hip.
#ifndef JPEGHIP_KERNELS_H
#define JPEGHIP_KERNELS_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace jpeghip {

// Everything the kernels need to know about the band being encoded, passed by value.
struct KGeom {
    int W;                   // real image width in pixels
    int rowsAvail;           // real pixel rows reachable from the RGB pointer
    long long strideBytes;   // RGB row stride
    int sampleFormat;        // JPEG_HIP_RGB8 / JPEG_HIP_RGBF32
    int stages;              // JPEG_HIP_STAGE_* bits
    int trueDivision;        // a quantization divisor is not an integer 1..255: only the generic kernel (IEEE division) is exact
    int mcusPerRow;
    long long nMcu;          // MCUs in the band
    long long totalBlocks;
    int bpm;                 // blocks per MCU
    int off1, off2;          // first Cb / Cr block inside an MCU
    int hF[3], vF[3];        // blocks per MCU across / down, per channel
    int wf[3], wm[3], hf[3], hm[3];
    long long mcusPerInterval;   // restart interval, or nMcu when restarts are off
    long long intervals;
    long long firstInterval;     // global index of the band's first interval
    int padOnes;
    int writeEoi;                // the band ends the image: EOI follows its last interval
    // Batch of equally shaped images encoded in one pass: image i occupies MCUs
    // [i * mcusPerImage, (i + 1) * mcusPerImage), its pixels start imageStrideBytes after the
    // previous image's, and every image is one interval with its own tables and EOI.
    int batch;
    long long mcusPerImage;
    long long blocksPerImage;
    long long imageStrideBytes;
    int tableStride;             // words between the table images of consecutive pictures (kTabImageWords, or 0: one shared set)
    // One byte per block of the band (or batch), written by the transform: bit k is set when the
    // block's k-th group of eight zig-zag ordered coefficients is not all zero.
    const unsigned char* groupMask;
    // Where a block's coefficients lie.  Groups 0..3 of block b start coefPitch * b bytes into the
    // buffer, its groups 4..7 coefHigh bytes behind them: 128 and 64 when a block's 128 bytes
    // are contiguous; 64 and half the buffer's size when the two halves of all blocks form
    // two planes (the second half of most blocks of a photographic image is empty and is then
    // never fetched: memory is read in lines of 128 bytes).  coefHigh stays below 2^31.
    int coefPitch;
    long long coefHigh;
    // Encoder::multiplyColorChannelBy calls made after convertToYCbCr, in call order.
    int nGains;
    struct {
        int channel;
        int afterSubsampling;
        float value;
    } gains[4];
};

// Where a batch's files go: slot i starts at out + i * slotBytes and receives header i
// (hdrLen[i] bytes from hdr + i * hdrPitch) followed by the image's scan and EOI.
struct BatchOutput {
    uint8_t* out;
    long long slotBytes;
    const uint8_t* hdr;
    const uint32_t* hdrLen;
    int hdrPitch;
    unsigned long long* ffStart;   // [batch + 1]: stuffing bytes in front of each image's stream
};

// What the bit writer for dense content keeps behind the states of its `tiles` tiles of 256 blocks, in 64-bit
// words: a head that starts at zero (the device's counter of tile numbers, the number of listed tiles, a ticket
// counter per XCD, 128 bytes apart), a table of chunks of tile numbers per XCD (starts at zero as well), two edge
// entries per tile, the list of tiles left to the second kernel.
constexpr int kDenseXcds = 16;           // as many as the hardware register's field can name
constexpr int kDenseXcdStride = 16;      // words between the ticket counters of two XCDs
constexpr int kDenseHead = 2 + kDenseXcds * kDenseXcdStride;
constexpr unsigned kDenseChunk = 32;     // tile numbers an XCD fetches from the device's counter at a time
constexpr unsigned kDenseMaxGrid = 2048; // workgroups of the dense bit writer at most
constexpr unsigned kDenseAhead = 1;      // chunks an XCD has in hand beyond the one its workgroups draw from (>= 1)
// Chunks an XCD can come to fetch: every tile and every workgroup's last, unsuccessful ticket could be its own,
// plus those fetched ahead.
constexpr long long dense_chunks_per_xcd(long long tiles) {
    return (tiles + kDenseMaxGrid) / kDenseChunk + kDenseMaxGrid / (8 * kDenseChunk) + kDenseAhead + 3;
}
constexpr long long dense_zeroed_words(long long tiles) { return tiles + kDenseHead + kDenseXcds * dense_chunks_per_xcd(tiles); }
constexpr long long dense_state_words(long long tiles) { return dense_zeroed_words(tiles) + 3 * tiles; }

// Device workspace of the entropy stage (all sizes fixed by the geometry).
struct EntropyWorkspace {
    unsigned long long* ivOff;   // [intervals + 1]: byte offset of every interval in the unstuffed stream
    uint32_t* ffCount;           // [maxChunks]
    unsigned long long* groupSum; // [maxChunks / 64 + 1]: the counts of 64 consecutive chunks added up
    long long groupSumsFrom;      // streams of more chunks than this get the group sums (0: the built-in limit)
    unsigned long long* result;  // [4]: stuffed length, error flags (1: output too small, 2: tile chain broken),
                                 //      unstuffed bytes, number of 4 KiB chunks
    uint8_t* unstuffed;          // [unstuffedCap]
    long long unstuffedCap;
    long long maxChunks;
    unsigned long long* tileState;   // [tiles + 1] of the single-pass bit writer (tiles of 256 blocks); dense_state_words
    long long streamBound;           // upper bound of the unstuffed stream in bytes, or < 0 when unknown
    long long avgBlockBits;          // mean entropy-coded bits per block according to the histograms, or < 0 when unknown
    int groupTile;                   // blocks per tile of the bit writer on groups: 256, 64 or 16; 0: the one-lane-per-block writer
    // Room for the staged bits of the tiles of 256 blocks (group_slot_bytes), or null: the tiles
    // then chain with a look-back and place their bits themselves.
    uint32_t* tileSlots;
    const uint8_t* header;           // file header staged in device memory, copied to the front of the output
    int headerLen;                   // (0: none); prefixLen of launch_entropy covers it
    // A table image (kTabImageWords words) and, right behind it, headerLen header bytes in pinned host
    // memory that have not been sent to the device yet, or null: the first kernel of the
    // stage fetches them into d_tables / header itself, which is cheaper than a copy call.
    const uint32_t* pendingTables;
    // Pinned host copy of the result record, or null: [0] stuffed length, [1] output too
    // small, [2] tile chain broken, written by the kernels themselves so that the host needs
    // no copy after the stage (cleared by the host before the launch).
    unsigned long long* hostResult;
};

// ---- Huffman tables as the kernels read them ---------------------------------------------------
// A table set travels from the host as one image of kTabImageWords words per picture: the four
// tables as they are (256 entries each, length << 16 | code; order DC-Y, DC-C, AC-Y, AC-C) and,
// behind them, the token templates the bit writers look up, built once on the host instead of
// by every workgroup: one word per (table, run, category) -- the symbol's code shifted left by the
// category, so that the amplitude bits can be ORed in, in bits 26..0 and the total number of bits
// (code + amplitude, at most 27) in bits 31..27.  Rows of 16 categories lie kTabPitch words apart
// (with 16 the runs r and r + 2 of one category share an LDS bank: 32 banks for dword reads, and
// the lanes of a look-up differ in little else than their runs), 23 rows per AC table (a run of
// 16..22 in front of a group's first coefficient aliases run & 15), then the DC rows, the
// end-of-block and the zero-run-length entries.  Category 0 inside the AC rows stands for a zero
// coefficient: an empty token.
constexpr int kTabPitch = 17;
constexpr int kTabAcWords = 23 * kTabPitch;
constexpr int kTabDc = 2 * kTabAcWords;
constexpr int kTabEob = kTabDc + 32;
constexpr int kTabZrl = kTabEob + 2;
constexpr int kTabWords = kTabZrl + 2;
constexpr int kTabTemplates = 1024;                               // where the templates start inside an image
constexpr int kTabImageWords = (kTabTemplates + kTabWords + 3) / 4 * 4;   // (whole 16-byte pieces)
constexpr uint32_t kTokAbsent = 0xF8000000u;                      // the table has no code for a symbol (31 is no length)

// Template word i (0 <= i < kTabWords) of a table set.
inline uint32_t token_template(const uint32_t* src, int i) {
    uint32_t e, cat = 0;
    bool zeroCoef = false;
    if (i < kTabDc) {
        const int c = i / kTabAcWords, rem = i - c * kTabAcWords;
        const int row = rem / kTabPitch;
        cat = (uint32_t)(rem - row * kTabPitch) & 15u;   // (columns behind the sixteenth are never read)
        zeroCoef = cat == 0;
        e = src[(2 + c) * 256 + (((row & 15) << 4) | (int)cat)];
    } else if (i < kTabEob) {
        cat = (uint32_t)(i - kTabDc) & 15u;
        e = src[((i - kTabDc) >> 4) * 256 + (int)cat];
    } else {
        e = src[(2 + ((i - kTabEob) & 1)) * 256 + (i < kTabZrl ? 0 : 0xF0)];
    }
    const uint32_t len = e >> 16, code = e & 0xffffu;
    return zeroCoef ? 0u : ((len == 0 || len + cat > 27u) ? kTokAbsent : ((code << cat) | ((len + cat) << 27)));
}

// Fills a table image from lengths and codes (uint8/uint16 [4][256]).
inline void fill_table_image(const uint8_t lengths[4][256], const uint16_t codes[4][256], uint32_t* image) {
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 256; ++s) image[t * 256 + s] = ((uint32_t)lengths[t][s] << 16) | codes[t][s];
    for (int i = 0; i < kTabWords; ++i) image[kTabTemplates + i] = token_template(image, i);
    for (int i = kTabTemplates + kTabWords; i < kTabImageWords; ++i) image[i] = 0;
}

constexpr int kStuffChunk = 4096;   // unstuffed bytes handled by one workgroup pass

// Fused colour conversion + subsampling + DCT + quantization + zig-zag.  d_clearHist, when
// given, is one histogram set (uint64[4][256]) the kernel clears on the side for a following
// launch_histogram(..., clear = false).
// d_gmask receives one byte per block: which of its eight groups of eight coefficients are not all zero.
hipError_t launch_transform(const KGeom& g, const void* d_rgb, const float* d_qtab, int16_t* d_coef,
                            unsigned char* d_gmask, hipStream_t s, unsigned long long* d_clearHist = nullptr);

// Blocks [firstInMcu, firstInMcu + perMcu) of every MCU of the band to d_out, 128 contiguous bytes per
// block, from wherever g keeps the blocks' halves.
hipError_t launch_gather_blocks(const KGeom& g, const int16_t* d_coef, int firstInMcu, int perMcu, void* d_out, hipStream_t s);

// Pixel filter (JPEG_HIP_FILTER_*) over `rows` rows of `width` pixels: src in sampleFormat
// with srcStride bytes per row, dst float RGB with 3 * width floats per row (may be src).
hipError_t launch_pixel_filter(const void* d_src, int sampleFormat, long long srcStride, float* d_dst, int width, int rows,
                               int filter, int channel, float value, hipStream_t s);

// Copies rows of rowBytes bytes from any address and stride to rows dstStride apart
// (d_dst and dstStride multiples of 16; dstStride >= rowBytes rounded up to 16).  The
// source rows are those of images of rowsPerImage rows, srcImageStride apart; at most
// 65535 rows per launch.
hipError_t launch_realign_rows(const void* d_src, long long srcStride, long long srcImageStride, int rowsPerImage,
                               void* d_dst, long long dstStride, int rowBytes, int rows, hipStream_t s);

// Three packed float planes, planeFloats apart, to interleaved float pixels.
hipError_t launch_interleave_planes(const float* d_planes, long long planeFloats, float* d_dst, long long pixels,
                                    hipStream_t s);

// Symbol histograms (uint64[4][256] per image of the batch); d_hist is cleared first unless
// the caller has done that (launch_clear_histograms) for several launches at once.
// ex (single images only): the kernel also writes the finished histograms to ex->host, pinned
// host memory, so that the host needs no copy; ex->counter is a device word the workgroups
// count themselves off with, never reset, and ex->expected follows it from launch to launch.
constexpr int kHistogramMaxGrid = 2048;
struct HistogramExport {
    unsigned long long* counter;
    unsigned long long expected;
    unsigned long long* host;
    // Density of the image's densest tiles, for the choice of the bit writer (k_histogram_tiles
    // only; k_histogram leaves hostStats[3] at 0): stats is device memory, one word per workgroup
    // of the launch (kHistogramMaxGrid at most); hostStats (pinned) receives [0] the most non-empty groups in a
    // tile of 256 blocks, [1] the most non-zero AC coefficients in such a tile, [2] the most in a
    // run of 64 blocks (a quarter of a tile), [3] 1.  Both null: not wanted.
    unsigned long long* stats;
    unsigned long long* hostStats;
};
hipError_t launch_histogram(const KGeom& g, const int16_t* d_coef, unsigned long long* d_hist, hipStream_t s,
                            bool clear = true, HistogramExport* ex = nullptr);
hipError_t launch_clear_histograms(unsigned long long* d_hist, int images, hipStream_t s);

long long group_slot_bytes(const KGeom& g);   // EntropyWorkspace::tileSlots

// Single-pass bit writer (lengths, positions, bits) and byte stuffing.  d_tables:
// uint32[4][256], entry = (length << 16) | code.  Scan bytes go to d_out + prefixLen; EOI is appended when
// writeEoi is set.  ws.result[0] receives prefixLen + scan bytes (+ 2 with EOI);
// ws.result[1] is non-zero when outCapacity was too small.
hipError_t launch_entropy(const KGeom& g, const int16_t* d_coef, const uint32_t* d_tables, const EntropyWorkspace& ws,
                          uint8_t* d_out, long long prefixLen, long long outCapacity, int writeEoi, hipStream_t s);

// Batch form: d_tables holds one table set per image (uint32[batch][4][256]), the output
// goes to the slots of `bo`.  ws.result[1] is non-zero when a slot was too small; the
// per-image lengths follow from ws.ivOff, bo.ffStart and the header lengths.
hipError_t launch_entropy_batch(const KGeom& g, const int16_t* d_coef, const uint32_t* d_tables,
                                const EntropyWorkspace& ws, const BatchOutput& bo, hipStream_t s);

}  // namespace jpeghip

#endif
