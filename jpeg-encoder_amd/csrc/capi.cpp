// C ABI of libjpegenc_hip: context, workspace and orchestration of the kernels.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "context.h"

using namespace jpeghip;

namespace jpeghip {

static int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }
static int lcm_i(int a, int b) { return a / gcd_i(a, b) * b; }

int make_geometry(const jpeg_hip_params& p, Geometry& g) {
    g.integerDivisors = true;
    if (p.width < 1 || p.height < 1 || p.width > 65535 || p.height > 65535) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    for (int c = 0; c < 3; ++c) {
        g.wf[c] = p.sampling[4 * c];
        g.wm[c] = p.sampling[4 * c + 1];
        g.hf[c] = p.sampling[4 * c + 2];
        g.hm[c] = p.sampling[4 * c + 3];
        if (g.wf[c] < 1 || g.wf[c] > 4 || g.hf[c] < 1 || g.hf[c] > 4) return JPEG_HIP_ERR_INVALID_ARGUMENT;
        if ((g.wm[c] != JPEG_HIP_SUBSAMPLING && g.wm[c] != JPEG_HIP_AVERAGE) ||
            (g.hm[c] != JPEG_HIP_SUBSAMPLING && g.hm[c] != JPEG_HIP_AVERAGE))
            return JPEG_HIP_ERR_INVALID_ARGUMENT;
        // QTable takes any float (QuantizationTables.h:6-27) and the DQT segment stores (byte)q
        // (JPEGSegments.h:152-167), so a divisor has to lie in [1, 256).  Integers take the fast
        // kernels, anything else the kernel that really divides.
        for (int i = 0; i < 64; ++i) {
            const float q = p.qtable[c][i];
            if (!(q >= 1.0f && q < 256.0f)) return JPEG_HIP_ERR_INVALID_ARGUMENT;
            if (q != std::floor(q)) g.integerDivisors = false;
        }
    }
    if (p.restart_interval < 0 || p.restart_interval > 65535) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (p.standard_huffman != 0 && p.standard_huffman != 1) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    g.width = p.width;
    g.height = p.height;
    g.stepW = lcm_i(lcm_i(g.wf[0], g.wf[1]), g.wf[2]) * 8;   // SamplingScheme.h:36-38
    g.stepH = lcm_i(lcm_i(g.hf[0], g.hf[1]), g.hf[2]) * 8;
    g.simW = (p.width + g.stepW - 1) / g.stepW * g.stepW;    // Image.cpp:8-10
    g.simH = (p.height + g.stepH - 1) / g.stepH * g.stepH;
    g.mcusPerRow = g.simW / g.stepW;
    g.mcuRows = g.simH / g.stepH;
    g.blocksPerMcu = 0;
    for (int c = 0; c < 3; ++c) {
        g.hFactor[c] = g.stepW / g.wf[c] / 8;                 // SamplingScheme.h:39-42
        g.vFactor[c] = g.stepH / g.hf[c] / 8;
        if (g.hFactor[c] > 4 || g.vFactor[c] > 4) return JPEG_HIP_ERR_INVALID_ARGUMENT;
        g.chanOffset[c] = g.blocksPerMcu;
        g.chanBlocks[c] = g.hFactor[c] * g.vFactor[c];
        g.blocksPerMcu += g.chanBlocks[c];
    }
    int r0 = p.mcu_row_begin, r1 = p.mcu_row_end;
    if (r0 == 0 && r1 == 0) r1 = g.mcuRows;
    if (r0 < 0 || r1 > g.mcuRows || r0 >= r1) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    g.bandRow0 = r0;
    g.bandRows = r1 - r0;
    g.startsImage = r0 == 0;
    g.endsImage = r1 == g.mcuRows;
    g.nMcu = (int64_t)g.bandRows * g.mcusPerRow;
    g.totalBlocks = g.nMcu * g.blocksPerMcu;
    g.restartInterval = p.restart_interval;
    if (g.restartInterval > 0) {
        const int64_t firstMcu = (int64_t)r0 * g.mcusPerRow;
        if (firstMcu % g.restartInterval != 0) return JPEG_HIP_ERR_INVALID_ARGUMENT;
        g.firstInterval = firstMcu / g.restartInterval;
        g.intervals = (g.nMcu + g.restartInterval - 1) / g.restartInterval;
    } else {
        if (!g.startsImage || !g.endsImage) return JPEG_HIP_ERR_INVALID_ARGUMENT;   // bands need restart markers
        g.firstInterval = 0;
        g.intervals = 1;
    }
    return JPEG_HIP_OK;
}

}  // namespace jpeghip

namespace {

int fail(jpeg_hip_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return status;
}

int hip_fail(jpeg_hip_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, JPEG_HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                      \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);  \
    } while (0)

// Makes the context's GPU the calling thread's current device for the length of an entry
// point and puts the caller's device back on every way out.
struct DeviceScope {
    int previous = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        if (previous != device) err = hipSetDevice(device);
        else previous = -1;   // nothing to put back
    }
    ~DeviceScope() {
        if (previous >= 0) (void)hipSetDevice(previous);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

#define ENTER_DEVICE(ctx)                                                              \
    DeviceScope deviceScope_((ctx)->device);                                           \
    if (deviceScope_.err != hipSuccess) return hip_fail(ctx, deviceScope_.err, "hipSetDevice")

int bytes_per_sample(int fmt) { return fmt == JPEG_HIP_RGBF32 ? 4 : 1; }

int check_pixels(jpeg_hip_ctx* ctx, const void* p, int fmt, int64_t stride) {
    if (!p) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "pixel pointer is null");
    if (fmt != JPEG_HIP_RGB8 && fmt != JPEG_HIP_RGBF32) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "unknown sample format");
    if (stride < (int64_t)ctx->geo.width * 3 * bytes_per_sample(fmt))
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "row stride is smaller than one row of pixels");
    if (fmt == JPEG_HIP_RGBF32 && (stride % 4 != 0 || ((uintptr_t)p & 3) != 0))
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "float pixels must be 4-byte aligned");
    if (fmt == JPEG_HIP_RGBF32 && ctx->prm.standard_huffman)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the standard Huffman tables cover 8-bit pixels only");
    return JPEG_HIP_OK;
}

// Real pixel rows reachable from the band's RGB pointer.
int band_rows_available(const Geometry& g) { return g.height - g.bandRow0 * g.stepH; }

// Chooses where the blocks' halves lie for a buffer of `blocks` blocks (see fill_kgeom).
void set_coefficient_layout(KGeom& k, int64_t blocks) {
    const char* splitEnv = getenv("JPEGENC_SPLIT_MIN_BLOCKS");
    // (The kernels keep the distance between the planes in 32 bits.)
    // 4:2:0 only: its luma blocks leave the transform four in a row; the other schemes store single blocks,
    // bpm apart, and halves of 64 bytes cost their transform more than the passes behind it gain
    // (8192 x 8192 4:4:4: transform 131 -> 155 us).
    const bool is420 = k.bpm == 6 && k.hF[0] == 2 && k.vF[0] == 2;
    const bool split = is420 && blocks >= (splitEnv ? atoll(splitEnv) : 400000) && blocks < (1ll << 25);
    k.coefPitch = split ? 64 : 128;
    k.coefHigh = split ? (long long)blocks * 64 : 64;
}

void fill_kgeom(jpeg_hip_ctx* ctx) {
    const Geometry& g = ctx->geo;
    KGeom& k = ctx->kg;
    k.W = g.width;
    k.rowsAvail = band_rows_available(g);
    k.strideBytes = ctx->stride;
    k.sampleFormat = ctx->fmt;
    k.stages = JPEG_HIP_STAGE_ALL;
    k.trueDivision = g.integerDivisors ? 0 : 1;
    k.mcusPerRow = g.mcusPerRow;
    k.nMcu = g.nMcu;
    k.totalBlocks = g.totalBlocks;
    k.bpm = g.blocksPerMcu;
    k.off1 = g.chanOffset[1];
    k.off2 = g.chanOffset[2];
    for (int c = 0; c < 3; ++c) {
        k.hF[c] = g.hFactor[c];
        k.vF[c] = g.vFactor[c];
        k.wf[c] = g.wf[c];
        k.wm[c] = g.wm[c];
        k.hf[c] = g.hf[c];
        k.hm[c] = g.hm[c];
    }
    k.mcusPerInterval = g.restartInterval > 0 ? g.restartInterval : g.nMcu;
    k.intervals = g.intervals;
    k.firstInterval = g.firstInterval;
    k.padOnes = ctx->prm.pad_ones ? 1 : 0;
    k.writeEoi = 0;
    k.batch = 1;
    k.mcusPerImage = g.nMcu;
    k.blocksPerImage = g.totalBlocks;
    k.imageStrideBytes = 0;
    k.tableStride = kTabImageWords;
    k.nGains = 0;
    k.groupMask = (const unsigned char*)ctx->gmask.p;
    // Large images keep groups 0..3 of all blocks in the first half of the coefficient buffer
    // and groups 4..7 in the second: the histogram pass and the bit writer read non-empty groups
    // only, memory comes in lines of 128 bytes, and the second half of most blocks of a
    // photographic image is empty (16384 x 16384 4:2:0 q=90: 29 % of the blocks have anything
    // there), so whole lines of the second plane are never fetched: 857 -> 607 MB per pass, histogram
    // pass 183 -> 157 us.  Smaller images (3840 x 2160 4:4:4, 292 k blocks: 23.5 -> 24.5 us) and
    // small batches keep a block's 128 bytes together.  (Diagnostic switch JPEGENC_SPLIT_MIN_BLOCKS,
    // read at every call so that a test can set it.)
    set_coefficient_layout(k, g.totalBlocks);
}

// Workspace for `images` images of the current geometry (1 outside batch mode; in a batch
// every image is one interval of the combined stream).
int ensure_workspace(jpeg_hip_ctx* ctx, int images = 1) {
    // With borrowed buffers the unstuffed stream gets what its slice holds, not the worst case.
    const WorkspaceSizes sz = workspace_sizes(ctx->geo, images, ctx->staticWorkspace ? (int64_t)ctx->unstuffed.cap : -1);
    const int64_t unstuffedCap = sz.unstuffedCap, maxChunks = sz.maxChunks;
    // A batch pass runs its entropy stage in two halves, each in a slice of these buffers: two
    // slices need a little more than one whole (constants of the size formulas count twice).
    const size_t slack = images > 1 ? 16384 : 0;
    HIP_TRY(ctx, ctx->coef.ensure(sz.coef));
    HIP_TRY(ctx, ctx->gmask.ensure(sz.gmask));
    HIP_TRY(ctx, ctx->qtab.ensure(sz.qtab));
    HIP_TRY(ctx, ctx->hist.ensure(sz.hist));
    HIP_TRY(ctx, ctx->tables.ensure(sz.tables));
    HIP_TRY(ctx, ctx->ivOff.ensure(sz.ivOff + slack));
    HIP_TRY(ctx, ctx->ffCount.ensure(sz.ffCount + slack));
    HIP_TRY(ctx, ctx->groupSum.ensure(sz.groupSum + slack));
    HIP_TRY(ctx, ctx->result.ensure(sz.result + slack));
    HIP_TRY(ctx, ctx->unstuffed.ensure(sz.unstuffed + slack));
    HIP_TRY(ctx, ctx->tileState.ensure(sz.tileState + slack));
    EntropyWorkspace& ws = ctx->ws;
    ws.ivOff = (unsigned long long*)ctx->ivOff.p;
    ws.ffCount = (uint32_t*)ctx->ffCount.p;
    ws.groupSum = (unsigned long long*)ctx->groupSum.p;
    // Diagnostic switch: the chunk count from which the stuffing pass uses group sums.
    ws.groupSumsFrom = getenv("JPEGENC_GROUP_SUMS_FROM") ? atoll(getenv("JPEGENC_GROUP_SUMS_FROM")) : 0;
    ws.result = (unsigned long long*)ctx->result.p;
    ws.unstuffed = (uint8_t*)ctx->unstuffed.p;
    ws.unstuffedCap = unstuffedCap;
    ws.maxChunks = maxChunks;
    ws.tileState = (unsigned long long*)ctx->tileState.p;
    ws.streamBound = -1;
    ws.avgBlockBits = -1;
    return JPEG_HIP_OK;
}

// Host clock at the stations of an encode (jpeg_hip_get_host_timeline), while profiling is on.
void tick(jpeg_hip_ctx* ctx, int i) {
    if (ctx->profiling)
        ctx->hostTimeline[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void mark(jpeg_hip_ctx* ctx, int i) {
    if (ctx->profiling) (void)hipEventRecord(ctx->ev[i], ctx->stream);
}

// Bits that tables spend on the symbols counted in hist[4][256]: code length plus the
// amplitude bits the symbol announces (the category for DC, the low nibble for AC).
int64_t coded_bits(const uint64_t* hist, const uint8_t lengths[4][256]) {
    int64_t bits = 0;
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 256; ++s)
            if (hist[t * 256 + s]) bits += (int64_t)hist[t * 256 + s] * (lengths[t][s] + (t < 2 ? s : (s & 15)));
    return bits;
}

// Waits for the context's stream.  Polls first: a blocking wait costs tens of microseconds
// to wake up, which is as long as the kernels of a small image take.
hipError_t wait_stream(jpeg_hip_ctx* ctx) {
    static const bool blocking = getenv("JPEGENC_BLOCKING_WAIT") != nullptr;
    if (blocking) return hipStreamSynchronize(ctx->stream);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(ctx->stream);
    }
}

// The same for an event recorded on the stream.
hipError_t wait_event(hipEvent_t ev) {
    static const bool blocking = getenv("JPEGENC_BLOCKING_WAIT") != nullptr;
    if (blocking) return hipEventSynchronize(ev);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipEventSynchronize(ev);
    }
}

// Packs the tables for the kernels next to the headerLen bytes of file header already placed
// in the pinned block.
int install_tables(jpeg_hip_ctx* ctx, int headerLen = 0) {
    fill_table_image(ctx->lengths, ctx->codes, ctx->pin->tables);
    // They stay in the pinned block until the entropy stage runs: its first kernel fetches
    // them from there (run_entropy).
    ctx->tablesPending = true;
    ctx->pendingHeaderLen = headerLen;
    ctx->hasTables = true;
    ctx->stagedScanLen = -1;
    ctx->streamBits = -1;
    return JPEG_HIP_OK;
}

// Header bytes for the band (none unless it starts the image) into the pinned block.
int stage_headers(jpeg_hip_ctx* ctx, int64_t capacity, int* headerLen) {
    *headerLen = 0;
    if (!ctx->geo.startsImage) return JPEG_HIP_OK;
    const int n = write_headers(ctx->prm, ctx->geo, ctx->lengths, ctx->codes, ctx->pin->header, kHeaderCapacity);
    if (n < 0) return fail(ctx, n, "header assembly failed");
    if (n > capacity) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "output buffer is too small for the headers");
    *headerLen = n;
    return JPEG_HIP_OK;
}

// Tables from histograms (or the standard ones when hist is null), installed on the device.
// With stageHeader the band's file header goes along and *headerLen receives its length.
int build_tables(jpeg_hip_ctx* ctx, const uint64_t* hist, bool stageHeader = false, int64_t capacity = 0,
                 int* headerLen = nullptr) {
    auto t0 = std::chrono::steady_clock::now();
    if (hist) {
        for (int t = 0; t < 4; ++t) {
            uint64_t total = 0;
            for (int s = 0; s < 256; ++s) total += hist[t * 256 + s];
            if (total == 0) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "a symbol histogram is empty");
            huffman_from_histogram(hist + t * 256, ctx->lengths[t], ctx->codes[t]);
        }
    } else {
        standard_huffman(ctx->lengths, ctx->codes);
    }
    int hn = 0;
    if (stageHeader) {
        const int st = stage_headers(ctx, capacity, &hn);
        if (st != JPEG_HIP_OK) return st;
        *headerLen = hn;
    }
    auto t1 = std::chrono::steady_clock::now();
    ctx->times[3] = std::chrono::duration<float, std::milli>(t1 - t0).count();
    const int st = install_tables(ctx, hn);
    // hist covers at least this band's symbols, so this bounds the band's stream.
    ctx->streamBits = hist ? coded_bits(hist, ctx->lengths) : -1;
    return st;
}

// Runs the entropy kernels and reads the length back.  prefix: bytes already placed at
// the start of d_out (headers).
// Launches the entropy kernels; run_entropy_wait collects their result.
// Tile size of the bit writer on groups for a mean block length (bits), 0: the one-lane-per-block
// writer.  A tile's bits are staged in 22 Kibit of LDS (88 per block of the largest tile) and its
// non-empty groups listed in room for five per block; a tile that does not fit is reported by the
// kernel and the stage is run again one size down (16 blocks fit whatever they hold).
int group_tile_for(int64_t avgBlockBits) {
    if (const char* forced = getenv("JPEGENC_GROUP_TILE")) return atoi(forced);   // diagnostic switch: 256, 64, 16 or 0
    if (avgBlockBits < 0 || avgBlockBits <= 56) return 256;
    if (avgBlockBits <= 224) return 64;
    return 0;
}

// The same from what the histogram pass saw of the image's densest tiles (HistogramExport::hostStats),
// so that the form that fits is taken at once instead of being found by running the stage again:
// the image's mean says nothing about a band of dense texture in an otherwise smooth picture.
// A tile's bits are estimated as its symbols (non-zero AC coefficients, a DC term and an end of
// block per block) times the image's mean bits per symbol, plus a margin for symbols that are
// rarer -- and so longer -- in the dense places than on average.  Dense tiles go to the
// one-lane-per-block writer, not to tiles of 16 blocks (4096 x 4096 with a band of noise: 95 against 143 us).
int group_tile_from_stats(const jpeg_hip_ctx* ctx, int64_t streamBits, const uint64_t* hist) {
    if (const char* forced = getenv("JPEGENC_GROUP_TILE")) return atoi(forced);
    uint64_t symbols = 0;
    for (int i = 0; i < 1024; ++i) symbols += hist[i];
    if (symbols == 0 || streamBits <= 0) return 256;
    const double bitsPerSymbol = (double)streamBits / (double)symbols;
    const double room = 0.97 * 22528.0;   // kGrpStageBits, less the slack the kernel keeps behind a lane's bits
    const uint64_t items = ctx->pin->tileStats[0], nz256 = ctx->pin->tileStats[1], nz64 = ctx->pin->tileStats[2];
    // (interval ends are items too: at most one per 16 blocks)
    const bool list256 = items + 16 <= 1278;   // kGrpMaxItems
    // Margins: a quarter more for the large tiles; half as much again for the small ones, which
    // are chosen when the large ones do not fit, i.e. for the dense places themselves (measured: the
    // band of noise in a smooth image costs 6.2 bits per symbol where the image's mean is 4.5), and
    // lose nothing to the one-lane-per-block writer on dense content anyway.
    if (list256 && (double)(nz256 + 2 * 256) * bitsPerSymbol * 1.25 <= room) return 256;
    if ((double)(nz64 + 2 * 64) * bitsPerSymbol * 1.5 <= room) return 64;
    return 0;
}

int launch_entropy_kernels(jpeg_hip_ctx* ctx) {
    // Padding (< 1 byte) and a marker per interval on top of the coded bits, EOI, slack.
    // Without histograms the output capacity bounds the stream: a longer one fails anyway.
    const int64_t prefix = ctx->emitPrefix, capacity = ctx->emitCapacity;
    ctx->ws.streamBound = ctx->streamBits < 0 ? capacity - prefix : ctx->streamBits / 8 + 3 * ctx->geo.intervals + 2 + 64;
    // A workspace of fixed size may hold less than the worst case: the stream has to fit.
    if (ctx->ws.unstuffedCap < workspace_sizes(ctx->geo, 1).unstuffedCap &&
        ctx->ws.streamBound + kStuffChunk + 64 > ctx->ws.unstuffedCap)
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "the entropy-coded stream does not fit the workspace");
    // The kernels write length and error flags into the pinned record themselves.
    ctx->pin->result[0] = ctx->pin->result[1] = ctx->pin->result[2] = ctx->pin->result[3] = 0;
    ctx->ws.hostResult = (unsigned long long*)ctx->pin->result;
    ++ctx->sumEntropyRuns;
    HIP_TRY(ctx, launch_entropy(ctx->kg, (const int16_t*)ctx->coef.p, (const uint32_t*)ctx->tables.p, ctx->ws, ctx->emitOut,
                                prefix, capacity, ctx->kg.writeEoi, ctx->stream));
    return JPEG_HIP_OK;
}

int run_entropy_launch(jpeg_hip_ctx* ctx, uint8_t* d_out, int64_t prefix, int64_t capacity, int writeEoi) {
    mark(ctx, 3);
    ctx->kg.writeEoi = writeEoi;
    // The histograms (and so streamBits) cover the whole image even when this context encodes a band.
    ctx->ws.avgBlockBits = ctx->streamBits < 0 ? -1 : ctx->streamBits / ((int64_t)ctx->geo.mcuRows * ctx->geo.mcusPerRow * ctx->geo.blocksPerMcu);
    ctx->ws.groupTile = ctx->tileStatsValid && ctx->streamBits >= 0 ? group_tile_from_stats(ctx, ctx->streamBits, ctx->pin->hist)
                                                                    : group_tile_for(ctx->ws.avgBlockBits);
    ctx->ws.tileSlots = nullptr;
    // Deferred placement of the tiles' bits pays for large images (16384 x 16384: 363 -> 315 us for
    // the stage, 3840 x 2160: 49 -> 50).  (Diagnostic switch JPEGENC_SLOTS_MIN_BLOCKS: another threshold.)
    const char* slotsEnv = getenv("JPEGENC_SLOTS_MIN_BLOCKS");
    const int64_t slotsFrom = slotsEnv ? atoll(slotsEnv) : 500000;
    if (!ctx->staticWorkspace && ctx->ws.groupTile == 256 && ctx->geo.totalBlocks >= slotsFrom) {
        HIP_TRY(ctx, ctx->tileSlots.ensure((size_t)group_slot_bytes(ctx->kg)));
        ctx->ws.tileSlots = (uint32_t*)ctx->tileSlots.p;
    }
    ctx->ws.header = (const uint8_t*)ctx->tables.p + sizeof ctx->pin->tables;
    ctx->ws.headerLen = (int)prefix;
    // A header staged with the tables is fetched with them; a launch without header (a
    // repeated emit, a band inside the image) still has to bring the tables over.
    ctx->ws.pendingTables = ctx->tablesPending ? ctx->pin->tables : nullptr;
    if (ctx->tablesPending && (int64_t)ctx->pendingHeaderLen != prefix) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tables.p, ctx->pin->tables, sizeof ctx->pin->tables + (size_t)ctx->pendingHeaderLen,
                                    hipMemcpyHostToDevice, ctx->stream));
        ctx->ws.pendingTables = nullptr;
    }
    ctx->tablesPending = false;
    ctx->emitOut = d_out;
    ctx->emitPrefix = prefix;
    ctx->emitCapacity = capacity;
    const int st = launch_entropy_kernels(ctx);
    mark(ctx, 4);
    return st;
}

int run_entropy_wait(jpeg_hip_ctx* ctx, int64_t* outLen) {
    HIP_TRY(ctx, wait_stream(ctx));
    tick(ctx, 7);
    if (ctx->profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->times[0] = ms;
        if (hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]) == hipSuccess) ctx->times[1] = ms;
        if (hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]) == hipSuccess) ctx->times[2] = ms;
        // Sums over the images since profiling was switched on, so that a caller in a tight loop
        // reads them once at the end instead of after every image.
        for (int i = 0; i < 4; ++i) ctx->sumTimes[i] += ctx->times[i];
        for (int i = 0; i < 7; ++i) ctx->sumTimeline[i] += ctx->hostTimeline[i + 1] - ctx->hostTimeline[i];
        if (ctx->lastResultSeen > 0) ctx->sumTimeline[7] += ctx->hostTimeline[0] - ctx->lastResultSeen;
        ctx->lastResultSeen = ctx->hostTimeline[7];
        ++ctx->sumCount;
    }
    // What the bit writer reports beside its bytes: 2 the chain of tiles broke, 4 the stream ran
    // past the part of the buffer that was cleared for it, 8 a tile's bits did not fit its
    // stage, 16 a symbol occurred that the tables have no code for.  4 and 8 are cured by running
    // the stage again with a full clear / with smaller tiles.
    for (int attempt = 0; attempt < 5 && (ctx->pin->result[3] & (4 | 8)) != 0 && (ctx->pin->result[3] & (2 | 16)) == 0; ++attempt) {
        if (ctx->pin->result[3] & 4) ctx->streamBits = -1;
        if (ctx->pin->result[3] & 8) ctx->ws.groupTile = ctx->ws.groupTile > 64 ? 64 : 0;   // (tiles of 16 blocks lose to the one-lane-per-block writer on dense content)
        ctx->ws.pendingTables = nullptr;   // tables and header are on the device by now
        const int st = launch_entropy_kernels(ctx);
        if (st != JPEG_HIP_OK) return st;
        HIP_TRY(ctx, wait_stream(ctx));
    }
    if (ctx->pin->result[3] & 16) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the Huffman tables have no code for a symbol of this image");
    if (ctx->pin->result[3] & (4 | 8)) return fail(ctx, JPEG_HIP_ERR_HIP, "the bit writer could not place its stream");
    if (ctx->pin->result[2] != 0 || (ctx->pin->result[3] & 2)) return fail(ctx, JPEG_HIP_ERR_HIP, "the bit writer's tiles lost their chain");
    if (ctx->pin->result[1] != 0) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "output buffer is too small for the scan");
    *outLen = (int64_t)ctx->pin->result[0];
    return JPEG_HIP_OK;
}

int run_entropy(jpeg_hip_ctx* ctx, uint8_t* d_out, int64_t prefix, int64_t capacity, int writeEoi, int64_t* outLen) {
    const int st = run_entropy_launch(ctx, d_out, prefix, capacity, writeEoi);
    return st != JPEG_HIP_OK ? st : run_entropy_wait(ctx, outLen);
}

// Byte pixels of `images` images (rowsPerImage rows each, imageStride apart) copied into
// ctx->repacked with rows on 16-byte boundaries and the images packed; *dstStride receives
// the new row stride.
int realign_pixels(jpeg_hip_ctx* ctx, const void* src, int64_t stride, int64_t imageStride, int rowsPerImage, int images,
                   int width, int64_t* dstStride) {
    const size_t rowBytes = (size_t)width * 3, pitch = (rowBytes + 15) / 16 * 16;
    HIP_TRY(ctx, ctx->repacked.ensure(pitch * rowsPerImage * images));
    const int perLaunch = std::max(1, 65535 / rowsPerImage);
    for (int first = 0; first < images; first += perLaunch) {
        const int cnt = std::min(perLaunch, images - first);
        HIP_TRY(ctx, launch_realign_rows((const uint8_t*)src + (int64_t)first * imageStride, stride, imageStride, rowsPerImage,
                                         (uint8_t*)ctx->repacked.p + (size_t)first * rowsPerImage * pitch, (long long)pitch,
                                         (int)rowBytes, cnt * rowsPerImage, ctx->stream));
    }
    *dstStride = (int64_t)pitch;
    return JPEG_HIP_OK;
}

// staged: the call comes from jpeg_hip_transform, which honours jpeg_hip_channel_gain.
// clearHist: histograms the caller will count into right afterwards (do_histograms with
// cleared = true).
int do_transform(jpeg_hip_ctx* ctx, int stages, bool staged = false, uint64_t* clearHist = nullptr) {
    if (!ctx->begun || !ctx->hasPixels) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "transform needs begin() and pixels");
    const Geometry& g = ctx->geo;
    const bool subsampled = g.blocksPerMcu != 3;
    if (subsampled && !(stages & JPEG_HIP_STAGE_SUBSAMPLE))
        return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "a subsampled scheme needs the subsample stage");
    fill_kgeom(ctx);
    ctx->kg.stages = stages;
    if (staged) {
        ctx->kg.nGains = ctx->nGains;
        for (int i = 0; i < ctx->nGains; ++i) {
            ctx->kg.gains[i].channel = ctx->gains[i].channel;
            ctx->kg.gains[i].afterSubsampling = ctx->gains[i].afterSubsampling;
            ctx->kg.gains[i].value = ctx->gains[i].value;
        }
    }
    // The fast kernels read byte pixels in dwords.  Caller-owned rows that do not start on
    // 4-byte boundaries (an odd width, packed) are realigned by a copy kernel first, a
    // fraction of what the byte-wise kernel would cost.
    const void* rgb = ctx->rgb;
    // (Not with a borrowed workspace, which has no room for the copy: the byte-wise kernel
    // takes such rows as they are.)
    if (ctx->fmt == JPEG_HIP_RGB8 && stages == JPEG_HIP_STAGE_ALL && ctx->kg.nGains == 0 && !ctx->staticWorkspace &&
        ((ctx->stride & 3) != 0 || ((uintptr_t)rgb & 3) != 0)) {
        const int rows = std::min(band_rows_available(g), g.bandRows * g.stepH);
        int64_t dstStride = 0;
        const int st = realign_pixels(ctx, rgb, ctx->stride, 0, rows, 1, g.width, &dstStride);
        if (st != JPEG_HIP_OK) return st;
        rgb = ctx->repacked.p;
        ctx->kg.strideBytes = (long long)dstStride;
    }
    mark(ctx, 0);
    HIP_TRY(ctx, launch_transform(ctx->kg, rgb, (const float*)ctx->qtab.p, (int16_t*)ctx->coef.p, (unsigned char*)ctx->gmask.p,
                                  ctx->stream, (unsigned long long*)clearHist));
    mark(ctx, 1);
    ctx->transformed = true;
    ctx->tileStatsValid = ctx->tileStatsPending = false;
    ctx->stagedScanLen = -1;
    // Whatever histograms the installed tables came from, they did not count these
    // coefficients: the bit count derived from them bounds nothing any more.
    ctx->streamBits = -1;
    return JPEG_HIP_OK;
}

// toHost: the kernel also leaves the finished histograms in the pinned block (pin->hist), for
// a finish step that is told to use them as they are.
int do_histograms(jpeg_hip_ctx* ctx, uint64_t* d_hist, bool cleared = false, bool toHost = false) {
    if (!ctx->transformed) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "histograms need the transform");
    ctx->histOnHost = false;
    HistogramExport ex = {nullptr, 0, nullptr, nullptr, nullptr};
    if (toHost) {
        if (!ctx->histDone.p) {   // the workgroups' counter starts at zero and is never reset; behind it a word per workgroup for the tile statistics
            HIP_TRY(ctx, ctx->histDone.ensure(64 + 8 * kHistogramMaxGrid));
            HIP_TRY(ctx, hipMemsetAsync(ctx->histDone.p, 0, 64, ctx->stream));
            ctx->histExpected = 0;
        }
        ex.counter = (unsigned long long*)ctx->histDone.p;
        ex.expected = ctx->histExpected;
        ex.host = (unsigned long long*)ctx->pin->hist;
        // (Diagnostic switch JPEGENC_TILE_STATS=0: the pass keeps no statistics and the bit writer's form follows the mean block length.)
        static const bool noStats = getenv("JPEGENC_TILE_STATS") != nullptr && getenv("JPEGENC_TILE_STATS")[0] == '0';
        ex.stats = noStats ? nullptr : (unsigned long long*)ctx->histDone.p + 8;
        ex.hostStats = (unsigned long long*)ctx->pin->tileStats;
    }
    HIP_TRY(ctx, launch_histogram(ctx->kg, (const int16_t*)ctx->coef.p, (unsigned long long*)d_hist, ctx->stream, !cleared,
                                  toHost ? &ex : nullptr));
    ctx->tileStatsValid = false;
    ctx->tileStatsPending = toHost;   // the kernel leaves them in the pinned block
    if (toHost) {
        ctx->histExpected = ex.expected;
        ctx->histOnHost = true;
    }
    mark(ctx, 2);
    return JPEG_HIP_OK;
}

// Histograms to the host (not for the standard tables), tables and header to the device.
int fetch_hist_and_build(jpeg_hip_ctx* ctx, const uint64_t* d_hist, int64_t capacity, int* headerLen) {
    if (ctx->prm.standard_huffman) return build_tables(ctx, nullptr, true, capacity, headerLen);
    if (!d_hist) {   // the band's own histograms, which the kernel has put into the pinned block
        if (!ctx->histOnHost) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "no histograms of this band: pass the summed ones");
        HIP_TRY(ctx, wait_stream(ctx));
        tick(ctx, 3);
        ctx->tileStatsValid = ctx->pin->tileStats[3] == 1;
        return build_tables(ctx, ctx->pin->hist, true, capacity, headerLen);
    }
    ctx->histOnHost = false;   // the pinned block is about to hold the caller's histograms
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pin->hist, d_hist, sizeof ctx->pin->hist, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    tick(ctx, 3);
    // (the statistics are the band's own even when the histograms are the whole image's)
    ctx->tileStatsValid = ctx->tileStatsPending && ctx->pin->tileStats[3] == 1;
    return build_tables(ctx, ctx->pin->hist, true, capacity, headerLen);
}

}  // namespace

extern "C" {

const char* jpeg_hip_status_string(int status) {
    switch (status) {
        case JPEG_HIP_OK: return "ok";
        case JPEG_HIP_ERR_INVALID_ARGUMENT: return "invalid argument";
        case JPEG_HIP_ERR_HIP: return "HIP runtime error";
        case JPEG_HIP_ERR_BUFFER_TOO_SMALL: return "buffer too small";
        case JPEG_HIP_ERR_BAD_STATE: return "entry points called out of order";
        case JPEG_HIP_ERR_NO_DEVICE: return "no usable GPU";
        case JPEG_HIP_ERR_PEER_FAILED: return "another rank of the sharded encode failed";
        default: return "unknown status";
    }
}

int jpeg_hip_create(int device_id, jpeg_hip_ctx** out_ctx) {
    if (!out_ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return JPEG_HIP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    DeviceScope scope(device_id);
    if (scope.err != hipSuccess) return JPEG_HIP_ERR_HIP;
    if (static_workspace_init(device_id) != hipSuccess) return JPEG_HIP_ERR_HIP;
    jpeg_hip_ctx* ctx = new (std::nothrow) jpeg_hip_ctx();
    if (!ctx) return JPEG_HIP_ERR_HIP;
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&ctx->pin, sizeof(Pinned), hipHostMallocDefault) != hipSuccess) {
        jpeg_hip_destroy(ctx);
        return JPEG_HIP_ERR_HIP;
    }
    for (auto& e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            jpeg_hip_destroy(ctx);
            return JPEG_HIP_ERR_HIP;
        }
    *out_ctx = ctx;
    return JPEG_HIP_OK;
}

void jpeg_hip_destroy(jpeg_hip_ctx* ctx) {
    if (!ctx) return;
    DeviceScope scope(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->pixels, &ctx->coef, &ctx->gmask, &ctx->qtab, &ctx->hist, &ctx->tables, &ctx->ivOff, &ctx->ffCount, &ctx->groupSum, &ctx->result,
                      &ctx->unstuffed, &ctx->staging, &ctx->batchHdr, &ctx->batchHdrLen, &ctx->batchFfStart,
                      &ctx->tileState, &ctx->decCoef, &ctx->decPlanes, &ctx->decTables, &ctx->decRgb, &ctx->decSum, &ctx->pixelsF, &ctx->pipeIn[0], &ctx->pipeIn[1],
                      &ctx->pipeOut, &ctx->histDone, &ctx->gather, &ctx->joined, &ctx->repacked, &ctx->tileSlots, &ctx->exchange};
    delete ctx->decoded;
    delete ctx->workers;
    delete ctx->packers;
    for (DevBuf* b : bufs) b->release();
    HostBuf* hbufs[] = {&ctx->hostHist, &ctx->hostTables, &ctx->hostHdr, &ctx->hostHdrLen, &ctx->hostOffsets, &ctx->uploadStage[0], &ctx->uploadStage[1],
                        &ctx->pipeStage[0], &ctx->pipeStage[1]};
    for (auto& e : ctx->pipeEvent)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->uploadEvent)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->partEvent)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->partCounted)
        if (e) (void)hipEventDestroy(e);
    for (auto& grp : ctx->groupTime)
        for (auto& e : grp)
            if (e) (void)hipEventDestroy(e);
    if (ctx->histStream) (void)hipStreamDestroy(ctx->histStream);
    if (ctx->tableStream) (void)hipStreamDestroy(ctx->tableStream);
    if (ctx->copyStream) (void)hipStreamDestroy(ctx->copyStream);
    for (HostBuf* b : hbufs) b->release();
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->orderEvent) (void)hipEventDestroy(ctx->orderEvent);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* jpeg_hip_last_error(const jpeg_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void* jpeg_hip_stream(jpeg_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int jpeg_hip_wait_for_stream(jpeg_hip_ctx* ctx, void* hip_stream) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if ((hipStream_t)hip_stream == ctx->stream) return JPEG_HIP_OK;
    ENTER_DEVICE(ctx);
    if (!ctx->orderEvent) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->orderEvent, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->orderEvent, (hipStream_t)hip_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->orderEvent, 0));
    return JPEG_HIP_OK;
}

int jpeg_hip_stream_wait_for_ctx(jpeg_hip_ctx* ctx, void* hip_stream) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if ((hipStream_t)hip_stream == ctx->stream) return JPEG_HIP_OK;
    ENTER_DEVICE(ctx);
    if (!ctx->orderEvent) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->orderEvent, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->orderEvent, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)hip_stream, ctx->orderEvent, 0));
    return JPEG_HIP_OK;
}

int jpeg_hip_compute_layout(const jpeg_hip_params* params, jpeg_hip_layout* out) {
    if (!params || !out) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    Geometry g;
    const int st = make_geometry(*params, g);
    if (st != JPEG_HIP_OK) return st;
    out->padded_width = g.simW;
    out->padded_height = g.simH;
    out->mcu_width = g.stepW;
    out->mcu_height = g.stepH;
    out->mcus_per_row = g.mcusPerRow;
    out->mcu_rows = g.mcuRows;
    out->band_mcu_rows = g.bandRows;
    out->blocks_per_mcu = g.blocksPerMcu;
    out->total_blocks = g.totalBlocks;
    for (int c = 0; c < 3; ++c) {
        out->h_factor[c] = g.hFactor[c];
        out->v_factor[c] = g.vFactor[c];
        out->blocks[c] = g.nMcu * g.chanBlocks[c];
    }
    out->intervals = g.intervals;
    // Headers, every scan byte stuffed, one marker per interval, EOI.
    out->max_jpeg_bytes = 2048 + 2 * worst_case_scan_bytes(g.totalBlocks) + 2 * g.intervals + 2;
    return JPEG_HIP_OK;
}

int jpeg_hip_begin(jpeg_hip_ctx* ctx, const jpeg_hip_params* params) {
    if (!ctx || !params) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    ENTER_DEVICE(ctx);
    Geometry g;
    const int st = make_geometry(*params, g);
    if (st != JPEG_HIP_OK) return fail(ctx, st, "invalid image parameters");
    ctx->prm = *params;
    ctx->geo = g;
    ctx->begun = true;
    ctx->hasPixels = ctx->transformed = ctx->hasTables = false;
    ctx->histOnHost = false;
    ctx->nGains = 0;
    const int ws = ensure_workspace(ctx);
    if (ws != JPEG_HIP_OK) return ws;
    // The divisors stay on the device from one image to the next while they do not change.
    if (ctx->qtabOnDevice && memcmp(ctx->pin->qtab, params->qtable, 192 * sizeof(float)) == 0) return JPEG_HIP_OK;
    // Earlier work on the stream may still read the pinned staging block.
    HIP_TRY(ctx, wait_stream(ctx));
    for (int i = 0; i < 192; ++i) {
        const float q = params->qtable[i / 64][i % 64];
        const double inv = 1.0 / (double)q;
        const float hi = (float)inv;
        ctx->pin->qtab[i] = q;
        ctx->pin->qtab[192 + i] = hi;
        ctx->pin->qtab[384 + i] = (float)(inv - (double)hi);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->qtab.p, ctx->pin->qtab, sizeof ctx->pin->qtab, hipMemcpyHostToDevice, ctx->stream));
    ctx->qtabOnDevice = true;
    return JPEG_HIP_OK;
}

int jpeg_hip_bind_pixels(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "call jpeg_hip_begin first");
    const int st = check_pixels(ctx, d_rgb, sample_format, stride_bytes);
    if (st != JPEG_HIP_OK) return st;
    ctx->rgb = d_rgb;
    ctx->fmt = sample_format;
    ctx->stride = stride_bytes;
    ctx->hasPixels = true;
    ctx->transformed = false;
    return JPEG_HIP_OK;
}

// Rows of rowBytes bytes, stride_bytes apart in host memory, to d_dst with rows dstStride apart.
// A two-dimensional copy out of pageable memory goes row by row, ten microseconds each
// (1080 rows: 10 ms).  Instead the rows are packed into pinned blocks of about 32 MiB by
// a few host threads and sent as one piece each, two blocks taking turns.  Returns with
// the copies queued on the stream; the caller waits.
static int upload_rows(jpeg_hip_ctx* ctx, const void* h_src, int64_t stride_bytes, size_t rowBytes, size_t dstStride,
                       int rows, uint8_t* d_dst) {
    const int rowsPerPiece = (int)std::max<size_t>(1, std::min<size_t>((size_t)rows, (32u << 20) / dstStride));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, ctx->uploadStage[i].ensure((size_t)rowsPerPiece * dstStride));
        if (!ctx->uploadEvent[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->uploadEvent[i], hipEventDisableTiming));
    }
    if (!ctx->packers) ctx->packers = new WorkerPool(3);
    // The blocks may still be on their way for an earlier call's rows.
    for (int i = 0; i < 2; ++i) HIP_TRY(ctx, hipEventSynchronize(ctx->uploadEvent[i]));
    int piece = 0;
    for (int row0 = 0; row0 < rows; row0 += rowsPerPiece, ++piece) {
        const int n = std::min(rowsPerPiece, rows - row0);
        uint8_t* stage = (uint8_t*)ctx->uploadStage[piece & 1].p;
        if (piece >= 2) HIP_TRY(ctx, hipEventSynchronize(ctx->uploadEvent[piece & 1]));   // its last copy has left
        const int slices = n >= 64 ? 16 : 1;
        ctx->packers->run(slices, [&](int t) {
            const int a = n * t / slices, b = n * (t + 1) / slices;
            const uint8_t* src = (const uint8_t*)h_src + (int64_t)(row0 + a) * stride_bytes;
            uint8_t* dst = stage + (size_t)a * dstStride;
            if ((size_t)stride_bytes == dstStride) {
                // the source's last row may end with its pixels, not with a full stride
                memcpy(dst, src, (size_t)(b - a - 1) * dstStride + rowBytes);
            } else {
                for (int y = a; y < b; ++y, src += stride_bytes, dst += dstStride) memcpy(dst, src, rowBytes);
            }
        });
        HIP_TRY(ctx, hipMemcpyAsync(d_dst + (size_t)row0 * dstStride, stage, (size_t)(n - 1) * dstStride + rowBytes,
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->uploadEvent[piece & 1], ctx->stream));
    }
    return JPEG_HIP_OK;
}

int jpeg_hip_upload_pixels(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "call jpeg_hip_begin first");
    int st = check_pixels(ctx, h_rgb, sample_format, stride_bytes);
    if (st != JPEG_HIP_OK) return st;
    ENTER_DEVICE(ctx);
    const int rows = std::min(band_rows_available(ctx->geo), ctx->geo.bandRows * ctx->geo.stepH);
    const size_t rowBytes = (size_t)ctx->geo.width * 3 * bytes_per_sample(sample_format);
    const size_t dstStride = (rowBytes + 3) / 4 * 4;
    HIP_TRY(ctx, ctx->pixels.ensure(dstStride * rows));
    st = upload_rows(ctx, h_rgb, stride_bytes, rowBytes, dstStride, rows, (uint8_t*)ctx->pixels.p);
    if (st != JPEG_HIP_OK) return st;
    HIP_TRY(ctx, wait_stream(ctx));
    ctx->rgb = ctx->pixels.p;
    ctx->fmt = sample_format;
    ctx->stride = (int64_t)dstStride;
    ctx->hasPixels = true;
    ctx->transformed = false;
    return JPEG_HIP_OK;
}

int jpeg_hip_upload_planes(jpeg_hip_ctx* ctx, const float* h_r, const float* h_g, const float* h_b, int64_t row_stride_bytes) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "call jpeg_hip_begin first");
    if (!h_r || !h_g || !h_b) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "plane pointer is null");
    const size_t rowBytes = (size_t)ctx->geo.width * sizeof(float);
    if (row_stride_bytes < (int64_t)rowBytes) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "row stride is smaller than one row of samples");
    if (row_stride_bytes % 4 != 0 || (((uintptr_t)h_r | (uintptr_t)h_g | (uintptr_t)h_b) & 3) != 0)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "float planes must be 4-byte aligned");
    if (ctx->prm.standard_huffman)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the standard Huffman tables cover 8-bit pixels only");
    ENTER_DEVICE(ctx);
    const int rows = std::min(band_rows_available(ctx->geo), ctx->geo.bandRows * ctx->geo.stepH);
    const size_t planeBytes = rowBytes * rows;
    HIP_TRY(ctx, ctx->pixels.ensure(3 * planeBytes));
    HIP_TRY(ctx, ctx->pixelsF.ensure(3 * planeBytes));
    const float* planes[3] = {h_r, h_g, h_b};
    for (int c = 0; c < 3; ++c) {
        const int st = upload_rows(ctx, planes[c], row_stride_bytes, rowBytes, rowBytes, rows, (uint8_t*)ctx->pixels.p + c * planeBytes);
        if (st != JPEG_HIP_OK) return st;
    }
    HIP_TRY(ctx, launch_interleave_planes((const float*)ctx->pixels.p, (long long)(planeBytes / sizeof(float)),
                                          (float*)ctx->pixelsF.p, (long long)rows * ctx->geo.width, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    ctx->rgb = ctx->pixelsF.p;
    ctx->fmt = JPEG_HIP_RGBF32;
    ctx->stride = (int64_t)ctx->geo.width * 3 * sizeof(float);
    ctx->hasPixels = true;
    ctx->transformed = false;
    return JPEG_HIP_OK;
}

// Rows of pixels the band's RGB pointer covers.
static int band_pixel_rows(const jpeg_hip_ctx* ctx) {
    return std::min(band_rows_available(ctx->geo), ctx->geo.bandRows * ctx->geo.stepH);
}

int jpeg_hip_filter_pixels(jpeg_hip_ctx* ctx, int filter, int channel, float value) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun || !ctx->hasPixels) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "filters need begin() and pixels");
    if ((filter != JPEG_HIP_FILTER_SEPIA && filter != JPEG_HIP_FILTER_MULTIPLY) || channel < 0 || channel > 2)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "unknown filter or channel");
    if (ctx->prm.standard_huffman)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the standard Huffman tables cover 8-bit pixels only");
    ENTER_DEVICE(ctx);
    const int rows = band_pixel_rows(ctx), width = ctx->geo.width;
    HIP_TRY(ctx, ctx->pixelsF.ensure((size_t)rows * width * 3 * sizeof(float)));
    HIP_TRY(ctx, launch_pixel_filter(ctx->rgb, ctx->fmt, ctx->stride, (float*)ctx->pixelsF.p, width, rows, filter, channel,
                                     value, ctx->stream));
    ctx->rgb = ctx->pixelsF.p;
    ctx->fmt = JPEG_HIP_RGBF32;
    ctx->stride = (int64_t)width * 3 * sizeof(float);
    ctx->transformed = false;
    return JPEG_HIP_OK;
}

int jpeg_hip_channel_gain(jpeg_hip_ctx* ctx, int channel, float value, int after_subsampling) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "channel gains need begin()");
    if (channel < 0 || channel > 2) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "unknown channel");
    if (ctx->prm.standard_huffman)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the standard Huffman tables cover 8-bit pixels only");
    if (ctx->nGains >= JPEG_HIP_MAX_GAINS) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "too many channel gains");
    ctx->gains[ctx->nGains].channel = channel;
    ctx->gains[ctx->nGains].afterSubsampling = after_subsampling ? 1 : 0;
    ctx->gains[ctx->nGains].value = value;
    ++ctx->nGains;
    ctx->transformed = false;
    return JPEG_HIP_OK;
}

int jpeg_hip_get_pixels(jpeg_hip_ctx* ctx, float* h_rgb, int64_t capacity_floats) {
    if (!ctx || !h_rgb) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->begun || !ctx->hasPixels) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "no pixels");
    ENTER_DEVICE(ctx);
    const int rows = band_pixel_rows(ctx), width = ctx->geo.width;
    const int64_t n = (int64_t)rows * width * 3;
    if (capacity_floats < n) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "pixel buffer too small");
    if (ctx->rgb != ctx->pixelsF.p) {   // convert a copy; the pixels in use stay as they are
        HIP_TRY(ctx, ctx->pixelsF.ensure((size_t)n * sizeof(float)));
        HIP_TRY(ctx, launch_pixel_filter(ctx->rgb, ctx->fmt, ctx->stride, (float*)ctx->pixelsF.p, width, rows, -1, 0, 0.0f,
                                         ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(h_rgb, ctx->pixelsF.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    return JPEG_HIP_OK;
}

int jpeg_hip_transform(jpeg_hip_ctx* ctx, int stages) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    ENTER_DEVICE(ctx);
    return do_transform(ctx, stages, true);
}

int jpeg_hip_histograms(jpeg_hip_ctx* ctx, uint64_t* d_hist) {
    if (!ctx || !d_hist) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    ENTER_DEVICE(ctx);
    return do_histograms(ctx, d_hist);
}

int jpeg_hip_histograms_host(jpeg_hip_ctx* ctx, uint64_t* h_hist) {
    if (!ctx || !h_hist) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    ENTER_DEVICE(ctx);
    const int st = do_histograms(ctx, (uint64_t*)ctx->hist.p);
    if (st != JPEG_HIP_OK) return st;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pin->hist, ctx->hist.p, sizeof ctx->pin->hist, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    memcpy(h_hist, ctx->pin->hist, sizeof ctx->pin->hist);
    return JPEG_HIP_OK;
}

int jpeg_hip_huffman_from_histogram(const uint64_t counts[256], uint8_t lengths[256], uint16_t codes[256]) {
    if (!counts || !lengths || !codes) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    huffman_from_histogram(counts, lengths, codes);
    return JPEG_HIP_OK;
}

int jpeg_hip_huffman_from_histogram_limit(const uint64_t counts[256], int code_word_length, uint8_t lengths[256],
                                          uint64_t codes[256]) {
    if (!counts || !lengths || !codes) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    // HuffmanCodingByte.h:105 (no longer than the 64-bit code word) and :121 (ceil(log2(256 + 1)) = 9 at least).
    if (code_word_length < 9 || code_word_length > 64) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    huffman_from_histogram_limit(counts, code_word_length, lengths, codes);
    return JPEG_HIP_OK;
}

int jpeg_hip_standard_huffman(uint8_t lengths[4][256], uint16_t codes[4][256]) {
    if (!lengths || !codes) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    standard_huffman(lengths, codes);
    return JPEG_HIP_OK;
}

int jpeg_hip_write_dht(const uint8_t lengths[256], const uint16_t codes[256], int table_id, int coefficient_type,
                       uint8_t* h_out, int capacity) {
    if (!lengths || !codes || !h_out) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    return write_dht(lengths, codes, table_id, coefficient_type, h_out, capacity);
}

int jpeg_hip_write_headers(const jpeg_hip_params* params, const uint8_t lengths[4][256], const uint16_t codes[4][256],
                           uint8_t* h_out, int capacity) {
    if (!params || !lengths || !codes || !h_out) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    Geometry g;
    const int st = make_geometry(*params, g);
    if (st != JPEG_HIP_OK) return st;
    return write_headers(*params, g, lengths, codes, h_out, capacity);
}

int jpeg_hip_set_huffman(jpeg_hip_ctx* ctx, const uint8_t lengths[4][256], const uint16_t codes[4][256]) {
    if (!ctx || !lengths || !codes) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "call jpeg_hip_begin first");
    for (int t = 0; t < 4; ++t)
        if (!huffman_table_is_sane(lengths[t], codes[t]))
            return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "a Huffman table has a code longer than 16 bits, wider than its length, or lengths no prefix code can have");
    ENTER_DEVICE(ctx);
    memcpy(ctx->lengths, lengths, sizeof ctx->lengths);
    memcpy(ctx->codes, codes, sizeof ctx->codes);
    return install_tables(ctx);
}

int jpeg_hip_build_huffman(jpeg_hip_ctx* ctx, const uint64_t h_hist[4][256]) {
    if (!ctx || !h_hist) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->begun) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "call jpeg_hip_begin first");
    ENTER_DEVICE(ctx);
    return build_tables(ctx, &h_hist[0][0]);
}

int jpeg_hip_get_huffman(jpeg_hip_ctx* ctx, uint8_t lengths[4][256], uint16_t codes[4][256]) {
    if (!ctx || !lengths || !codes) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->hasTables) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "no Huffman tables installed");
    memcpy(lengths, ctx->lengths, sizeof ctx->lengths);
    memcpy(codes, ctx->codes, sizeof ctx->codes);
    return JPEG_HIP_OK;
}

int jpeg_hip_emit_scan(jpeg_hip_ctx* ctx, uint8_t* d_out, int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !d_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->transformed || !ctx->hasTables) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "emit needs coefficients and tables");
    ENTER_DEVICE(ctx);
    return run_entropy(ctx, d_out, 0, out_capacity, 0, out_len);
}

int jpeg_hip_emit_scan_host(jpeg_hip_ctx* ctx, uint8_t* h_out, int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    if (!ctx->transformed || !ctx->hasTables) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "emit needs coefficients and tables");
    ENTER_DEVICE(ctx);
    if (ctx->stagedScanLen < 0) {
        jpeg_hip_layout lay;
        int st = jpeg_hip_compute_layout(&ctx->prm, &lay);
        if (st != JPEG_HIP_OK) return st;
        HIP_TRY(ctx, ctx->staging.ensure((size_t)lay.max_jpeg_bytes));
        int64_t n = 0;
        st = run_entropy(ctx, (uint8_t*)ctx->staging.p, 0, lay.max_jpeg_bytes, 0, &n);
        if (st != JPEG_HIP_OK) return st;
        ctx->stagedScanLen = n;
    }
    *out_len = ctx->stagedScanLen;
    if (!h_out) return JPEG_HIP_OK;
    if (out_capacity < ctx->stagedScanLen) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "host scan buffer too small");
    HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->staging.p, (size_t)ctx->stagedScanLen, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    return JPEG_HIP_OK;
}

int jpeg_hip_get_coefficients(jpeg_hip_ctx* ctx, int channel, int16_t* h_out, int64_t capacity_values) {
    if (!ctx || !h_out || channel < 0 || channel > 2) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->transformed) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "no coefficients yet");
    ENTER_DEVICE(ctx);
    const Geometry& g = ctx->geo;
    const int64_t perMcu = g.chanBlocks[channel];
    if (capacity_values < g.nMcu * perMcu * 64) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "coefficient buffer too small");
    // Scan order interleaves the channels per MCU: this channel's blocks are gathered on the
    // device and leave in one piece (a strided copy to pageable memory goes row by row).
    const size_t bytes = (size_t)g.nMcu * perMcu * 128;
    HIP_TRY(ctx, ctx->gather.ensure(bytes));
    HIP_TRY(ctx, launch_gather_blocks(ctx->kg, (const int16_t*)ctx->coef.p, g.chanOffset[channel], (int)perMcu, ctx->gather.p, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->gather.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    return JPEG_HIP_OK;
}

const int16_t* jpeg_hip_coefficients_device(jpeg_hip_ctx* ctx) {
    if (!ctx || !ctx->transformed) return nullptr;
    if (ctx->kg.coefPitch == 128) return (const int16_t*)ctx->coef.p;
    // The two halves of the blocks lie apart (fill_kgeom): the caller gets a joined copy.
    DeviceScope deviceScope((ctx)->device);
    if (deviceScope.err != hipSuccess) return nullptr;
    const size_t bytes = (size_t)ctx->geo.totalBlocks * 128;
    if (ctx->joined.ensure(bytes) != hipSuccess ||
        launch_gather_blocks(ctx->kg, (const int16_t*)ctx->coef.p, 0, ctx->geo.blocksPerMcu, ctx->joined.p, ctx->stream) != hipSuccess ||
        wait_stream(ctx) != hipSuccess)
        return nullptr;
    return (const int16_t*)ctx->joined.p;
}

int jpeg_hip_analyze_band(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                          const jpeg_hip_params* params, uint64_t* d_hist) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    tick(ctx, 0);
    int st = jpeg_hip_begin(ctx, params);
    if (st != JPEG_HIP_OK) return st;
    st = jpeg_hip_bind_pixels(ctx, d_rgb, sample_format, stride_bytes);
    if (st != JPEG_HIP_OK) return st;
    // A null d_hist keeps the histograms in the context's own buffer.
    uint64_t* hist = d_hist ? d_hist : (uint64_t*)ctx->hist.p;
    const bool count = !ctx->prm.standard_huffman;   // fixed tables: nothing to count
    st = do_transform(ctx, JPEG_HIP_STAGE_ALL, false, count ? hist : nullptr);
    if (st != JPEG_HIP_OK) return st;
    if (!count) {
        mark(ctx, 2);
        tick(ctx, 1);
        return JPEG_HIP_OK;
    }
    st = do_histograms(ctx, hist, true, true);
    tick(ctx, 1);
    return st;
}

int jpeg_hip_finish_band(jpeg_hip_ctx* ctx, const uint64_t* d_hist_sum, uint8_t* d_out, int64_t out_capacity,
                         int64_t* out_len) {
    if (!ctx || !d_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    const int st = jpeg_hip_finish_band_start(ctx, d_hist_sum, d_out, out_capacity);
    if (st != JPEG_HIP_OK) return st;
    return jpeg_hip_finish_band_wait(ctx, out_len);
}

int jpeg_hip_finish_band_start(jpeg_hip_ctx* ctx, const uint64_t* d_hist_sum, uint8_t* d_out, int64_t out_capacity) {
    if (!ctx || !d_out) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->transformed) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "finish needs analyze");
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    ENTER_DEVICE(ctx);
    tick(ctx, 2);
    int headerLen = 0;
    int st = fetch_hist_and_build(ctx, d_hist_sum, out_capacity, &headerLen);
    if (st != JPEG_HIP_OK) return st;
    tick(ctx, 4);
    st = run_entropy_launch(ctx, d_out, headerLen, out_capacity, ctx->geo.endsImage ? 1 : 0);
    if (st != JPEG_HIP_OK) return st;
    tick(ctx, 5);
    ctx->scanInFlight = true;
    return JPEG_HIP_OK;
}

int jpeg_hip_finish_band_wait(jpeg_hip_ctx* ctx, int64_t* out_len) {
    if (!ctx || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "no jpeg_hip_finish_band_start to wait for");
    ENTER_DEVICE(ctx);
    ctx->scanInFlight = false;
    tick(ctx, 6);
    return run_entropy_wait(ctx, out_len);
}

int jpeg_hip_encode(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                    const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !d_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int st = jpeg_hip_analyze_band(ctx, d_rgb, sample_format, stride_bytes, params, nullptr);
    if (st != JPEG_HIP_OK) return st;
    return jpeg_hip_finish_band(ctx, nullptr, d_out, out_capacity, out_len);
}

int jpeg_hip_encode_host(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes,
                         const jpeg_hip_params* params, uint8_t* h_out, int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !h_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int st = jpeg_hip_begin(ctx, params);
    if (st != JPEG_HIP_OK) return st;
    st = jpeg_hip_upload_pixels(ctx, h_rgb, sample_format, stride_bytes);
    if (st != JPEG_HIP_OK) return st;
    st = do_transform(ctx, JPEG_HIP_STAGE_ALL, false, params->standard_huffman ? nullptr : (uint64_t*)ctx->hist.p);
    if (st != JPEG_HIP_OK) return st;
    if (!params->standard_huffman) {
        st = do_histograms(ctx, (uint64_t*)ctx->hist.p, true, true);
        if (st != JPEG_HIP_OK) return st;
    } else {
        mark(ctx, 2);
    }
    jpeg_hip_layout lay;
    st = jpeg_hip_compute_layout(params, &lay);
    if (st != JPEG_HIP_OK) return st;
    HIP_TRY(ctx, ctx->staging.ensure((size_t)lay.max_jpeg_bytes));
    ctx->stagedScanLen = -1;
    int64_t len = 0;
    st = jpeg_hip_finish_band(ctx, nullptr, (uint8_t*)ctx->staging.p, lay.max_jpeg_bytes, &len);
    if (st != JPEG_HIP_OK) return st;
    if (len > out_capacity) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "host output buffer too small");
    HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->staging.p, (size_t)len, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    *out_len = len;
    return JPEG_HIP_OK;
}

int jpeg_hip_set_profiling(jpeg_hip_ctx* ctx, int enabled) {
    if (!ctx) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ctx->profiling = enabled != 0;
    for (auto& v : ctx->sumTimes) v = 0;
    for (auto& v : ctx->sumTimeline) v = 0;
    ctx->sumCount = 0;
    ctx->sumEntropyRuns = 0;
    ctx->lastResultSeen = 0;
    return JPEG_HIP_OK;
}

int jpeg_hip_get_host_timeline(jpeg_hip_ctx* ctx, double out_us[8]) {
    if (!ctx || !out_us) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    memcpy(out_us, ctx->hostTimeline, sizeof ctx->hostTimeline);
    return JPEG_HIP_OK;
}

int jpeg_hip_get_profile_sums(jpeg_hip_ctx* ctx, double stage_ms[4], double host_us[8], int64_t* images, int64_t* entropy_runs) {
    if (!ctx || !stage_ms || !host_us || !images || !entropy_runs) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    *entropy_runs = ctx->sumEntropyRuns;
    memcpy(stage_ms, ctx->sumTimes, sizeof ctx->sumTimes);
    memcpy(host_us, ctx->sumTimeline, sizeof ctx->sumTimeline);
    *images = ctx->sumCount;
    return JPEG_HIP_OK;
}

int jpeg_hip_get_timings(jpeg_hip_ctx* ctx, float out_ms[4]) {
    if (!ctx || !out_ms) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    memcpy(out_ms, ctx->times, sizeof ctx->times);
    return JPEG_HIP_OK;
}

}  // extern "C"

// ---- batch of equally shaped images ------------------------------------------------------
namespace {

constexpr int kBatchHeaderPitch = 1024;          // bytes reserved per header in the staging blocks
constexpr int64_t kBatchMaxBlocks = 12 << 20;    // blocks per pass: 1.5 GiB of coefficients

// Huffman tables and headers of images [k0, k1) of a pass, from their histograms.
void batch_tables_worker(const jpeg_hip_ctx* ctx, int k0, int k1, const uint64_t* hist, uint32_t* tables, uint8_t* hdr,
                         uint32_t* hdrLen, int* status, int64_t* bits) {
    uint8_t lengths[4][256];
    uint16_t codes[4][256];
    *bits = 0;
    for (int k = k0; k < k1; ++k) {
        const uint64_t* h = hist + (size_t)k * 1024;
        for (int t = 0; t < 4; ++t) {
            uint64_t total = 0;
            for (int s = 0; s < 256; ++s) total += h[t * 256 + s];
            if (total == 0) {
                *status = JPEG_HIP_ERR_INVALID_ARGUMENT;
                return;
            }
            huffman_from_histogram(h + t * 256, lengths[t], codes[t]);
        }
        fill_table_image(lengths, codes, tables + (size_t)k * kTabImageWords);
        const int n = write_headers(ctx->prm, ctx->geo, lengths, codes, hdr + (size_t)k * kBatchHeaderPitch, kBatchHeaderPitch);
        if (n < 0) {
            *status = n;
            return;
        }
        hdrLen[k] = (uint32_t)n;
        *bits += coded_bits(h, lengths);
    }
}

// The entropy stage of images [a, a + cnt) of a batch pass, in its own slice of the workspace.
struct BatchHalf {
    int a = 0, cnt = 0;
    KGeom k;
    EntropyWorkspace ws;
    BatchOutput bo;
    uint64_t* off = nullptr;   // host: [cnt + 1] stream offsets, [cnt + 1] stuffing offsets, [4] result
    const int16_t* coef = nullptr;
    const uint32_t* tables = nullptr;
};

int launch_batch_half(jpeg_hip_ctx* ctx, BatchHalf& h) {
    HIP_TRY(ctx, launch_entropy_batch(h.k, h.coef, h.tables, h.ws, h.bo, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h.off, h.ws.ivOff, (size_t)(h.cnt + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h.off + h.cnt + 1, h.bo.ffStart, (size_t)(h.cnt + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h.off + 2 * (h.cnt + 1), h.ws.result, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
    return JPEG_HIP_OK;
}

// One pass: `n` images starting at d_rgb, output slots starting at d_out.
int encode_batch_pass(jpeg_hip_ctx* ctx, const uint8_t* d_rgb, int64_t imageStride, int n, uint8_t* d_out,
                      int64_t slotBytes, int64_t* lens) {
    const Geometry& g = ctx->geo;
    fill_kgeom(ctx);
    KGeom& k = ctx->kg;
    k.batch = n;
    k.nMcu = g.nMcu * n;
    k.totalBlocks = g.totalBlocks * n;
    // The planes (if any) span the whole pass; a part of it starts coefPitch bytes per block into
    // the first plane and finds its second halves the same distance behind.
    set_coefficient_layout(k, k.totalBlocks);
    // Frames whose rows do not start on 4-byte boundaries are realigned first (see do_transform).
    if (ctx->fmt == JPEG_HIP_RGB8 && ((ctx->stride & 3) != 0 || (imageStride & 3) != 0 || ((uintptr_t)d_rgb & 3) != 0)) {
        int64_t dstStride = 0;
        const int st = realign_pixels(ctx, d_rgb, ctx->stride, imageStride, g.height, n, g.width, &dstStride);
        if (st != JPEG_HIP_OK) return st;
        d_rgb = (const uint8_t*)ctx->repacked.p;
        imageStride = dstStride * g.height;
        k.strideBytes = (long long)dstStride;
    }
    k.imageStrideBytes = imageStride;
    k.mcusPerInterval = g.nMcu;
    k.intervals = n;
    k.firstInterval = 0;
    k.writeEoi = 1;
    const bool standard = ctx->prm.standard_huffman != 0;
    int parts = 1;
    // The entropy stage runs in two halves when the pass is cut into parts: the first half
    // while the host still builds the tables of the second.
    int nHalves = 1, tasksOfFirstHalf = 0;
    int64_t halfBits[2] = {0, 0};
    BatchHalf half[2];
    bool marked3 = false;
    auto start_half = [&](int hIdx, int64_t bitsOfHalf) -> int {
        BatchHalf& h = half[hIdx];
        h.ws.streamBound = bitsOfHalf < 0 ? (int64_t)h.cnt * slotBytes : bitsOfHalf / 8 + 3 * (int64_t)h.cnt + 2 + 64;
        h.ws.avgBlockBits = bitsOfHalf < 0 ? -1 : bitsOfHalf / ((int64_t)h.cnt * g.totalBlocks);
        h.ws.groupTile = group_tile_for(h.ws.avgBlockBits);
        if (!marked3) mark(ctx, 3);
        marked3 = true;
        return launch_batch_half(ctx, h);
    };
    // Room for the staged bits of every tile (deferred placement, see run_entropy_launch).
    uint32_t* slots = nullptr;
    if (!ctx->staticWorkspace && ctx->tileSlots.ensure((size_t)group_slot_bytes(k)) == hipSuccess) slots = (uint32_t*)ctx->tileSlots.p;
    auto set_halves = [&](int firstCount) {
        nHalves = firstCount > 0 && firstCount < n ? 2 : 1;
        const WorkspaceSizes w0 = workspace_sizes(g, nHalves == 2 ? firstCount : n);
        uint64_t* off = (uint64_t*)ctx->hostOffsets.p;
        for (int i = 0; i < nHalves; ++i) {
            BatchHalf& h = half[i];
            h.a = i == 0 ? 0 : firstCount;
            h.cnt = nHalves == 1 ? n : (i == 0 ? firstCount : n - firstCount);
            h.k = k;
            h.k.batch = h.cnt;
            h.k.intervals = h.cnt;
            h.k.nMcu = g.nMcu * h.cnt;
            h.k.totalBlocks = g.totalBlocks * h.cnt;
            h.k.groupMask = (const unsigned char*)ctx->gmask.p + (int64_t)h.a * g.totalBlocks;
            h.coef = (const int16_t*)ctx->coef.p + (int64_t)h.a * g.totalBlocks * (k.coefPitch / 2);
            h.tables = (const uint32_t*)ctx->tables.p + (h.k.tableStride ? (size_t)h.a * kTabImageWords : 0);
            h.ws = ctx->ws;
            h.ws.tileSlots = slots ? slots + (i == 1 ? group_slot_bytes(half[0].k) / 4 : 0) : nullptr;
            if (i == 1) {   // behind the first half's slices
                h.ws.ivOff += firstCount + 1;
                h.ws.ffCount += w0.maxChunks;
                h.ws.groupSum += w0.maxChunks / 64 + 2;
                h.ws.result += 4;
                h.ws.unstuffed += w0.unstuffedCap;
                h.ws.unstuffedCap = (int64_t)ctx->unstuffed.cap - w0.unstuffedCap;
                h.ws.unstuffedCap = h.ws.unstuffedCap / 16 * 16;
                h.ws.tileState += w0.tiles + 1;
            } else if (nHalves == 2) {
                h.ws.unstuffedCap = w0.unstuffedCap;
            }
            h.ws.maxChunks = (h.ws.unstuffedCap + kStuffChunk - 1) / kStuffChunk;
            if (i == 1 || nHalves == 2) {
                const WorkspaceSizes wi = workspace_sizes(g, h.cnt);
                if (h.ws.unstuffedCap > wi.unstuffedCap) h.ws.unstuffedCap = wi.unstuffedCap;
                h.ws.maxChunks = wi.maxChunks < h.ws.maxChunks ? wi.maxChunks : h.ws.maxChunks;
            }
            h.bo.out = d_out + (int64_t)h.a * slotBytes;
            h.bo.slotBytes = slotBytes;
            h.bo.hdr = (const uint8_t*)ctx->batchHdr.p + (size_t)h.a * kBatchHeaderPitch;
            h.bo.hdrLen = (const uint32_t*)ctx->batchHdrLen.p + h.a;
            h.bo.hdrPitch = kBatchHeaderPitch;
            h.bo.ffStart = (unsigned long long*)ctx->batchFfStart.p + (i == 0 ? 0 : firstCount + 1);
            h.off = off + (i == 0 ? 0 : 2 * (firstCount + 1) + 4);
        }
    };
    if (standard) {
        // One table set and one header serve every image; nothing waits for the host.
        mark(ctx, 0);
        HIP_TRY(ctx, launch_transform(k, d_rgb, (const float*)ctx->qtab.p, (int16_t*)ctx->coef.p, (unsigned char*)ctx->gmask.p,
                                      ctx->stream));
        mark(ctx, 1);
        mark(ctx, 2);
        k.tableStride = 0;
        uint8_t lengths[4][256];
        uint16_t codes[4][256];
        standard_huffman(lengths, codes);
        fill_table_image(lengths, codes, (uint32_t*)ctx->hostTables.p);
        uint8_t* hdr = (uint8_t*)ctx->hostHdr.p;
        const int hn = write_headers(ctx->prm, ctx->geo, lengths, codes, hdr, kBatchHeaderPitch);
        if (hn < 0) return fail(ctx, hn, "header assembly failed");
        uint32_t* hdrLen = (uint32_t*)ctx->hostHdrLen.p;
        for (int i = 0; i < n; ++i) {
            if (i > 0) memcpy(hdr + (size_t)i * kBatchHeaderPitch, hdr, (size_t)hn);
            hdrLen[i] = (uint32_t)hn;
        }
        halfBits[0] = -1;
        set_halves(0);
    } else {
    // The histogram pass runs in up to four parts, each followed (on a stream of its own, so
    // that the parts stay back to back) by the copy of its histograms: the host builds the
    // tables of one part, and sends them back on a third stream, while the device works on
    // the next ones.  Sixteen host threads need about as long for the tables of a pass as
    // the device for its histograms.
    if (!ctx->histStream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->histStream, hipStreamNonBlocking));
    if (!ctx->tableStream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->tableStream, hipStreamNonBlocking));
    // Four parts measure best for 256 frames of 1080p (1.72 ms per pass; two parts 1.78, eight 1.77: a part is a
    // launch of its own, and the fewer there are the later the host starts).  Diagnostic switch
    // JPEGENC_BATCH_PARTS: another upper limit.
    static const int partsLimit = getenv("JPEGENC_BATCH_PARTS") ? atoi(getenv("JPEGENC_BATCH_PARTS")) : 4;
    const int maxParts = partsLimit < 1 ? 1 : (partsLimit > kBatchParts ? kBatchParts : partsLimit);
    parts = n / 16 < 1 ? 1 : (n / 16 > maxParts ? maxParts : n / 16);
    // The transform runs in two launches: one would let the host start later, one per part
    // costs the transform a tenth of its speed.
    const int groups = parts >= 4 ? 2 : 1;
    int partBegin[kBatchParts + 1];
    for (int p = 0; p <= parts; ++p) partBegin[p] = (int)((int64_t)n * p / parts);
    set_halves(parts >= 2 ? partBegin[parts / 2] : 0);
    HIP_TRY(ctx, launch_clear_histograms((unsigned long long*)ctx->hist.p, n, ctx->stream));
    auto part_geometry = [&](int a, int cnt) {
        KGeom kp = k;
        kp.batch = cnt;
        kp.intervals = cnt;
        kp.nMcu = g.nMcu * cnt;
        kp.totalBlocks = g.totalBlocks * cnt;
        kp.groupMask = (const unsigned char*)ctx->gmask.p + (int64_t)a * g.totalBlocks;
        return kp;
    };
    for (int grp = 0; grp < groups; ++grp) {
        const int p0 = parts * grp / groups, p1 = parts * (grp + 1) / groups;
        const int ga = partBegin[p0], gcnt = partBegin[p1] - ga;
        // Timing events sit at the group's edges only: an event between two kernels costs
        // the stream more than ten microseconds.
        if (ctx->profiling) {
            for (auto& e : ctx->groupTime[grp])
                if (!e) HIP_TRY(ctx, hipEventCreate(&e));
            HIP_TRY(ctx, hipEventRecord(ctx->groupTime[grp][0], ctx->stream));
        }
        HIP_TRY(ctx, launch_transform(part_geometry(ga, gcnt), d_rgb + (int64_t)ga * imageStride, (const float*)ctx->qtab.p,
                                      (int16_t*)ctx->coef.p + (int64_t)ga * g.totalBlocks * (k.coefPitch / 2),
                                      (unsigned char*)ctx->gmask.p + (int64_t)ga * g.totalBlocks, ctx->stream));
        if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->groupTime[grp][1], ctx->stream));
        for (int p = p0; p < p1; ++p) {
            const int a = partBegin[p], cnt = partBegin[p + 1] - a;
            HIP_TRY(ctx, launch_histogram(part_geometry(a, cnt), (const int16_t*)ctx->coef.p + (int64_t)a * g.totalBlocks * (k.coefPitch / 2),
                                          (unsigned long long*)ctx->hist.p + (size_t)a * 1024, ctx->stream, false));
            if (!ctx->partCounted[p]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->partCounted[p], hipEventDisableTiming));
            if (!ctx->partEvent[p]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->partEvent[p], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ctx->partCounted[p], ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->histStream, ctx->partCounted[p], 0));
            HIP_TRY(ctx, hipMemcpyAsync((uint64_t*)ctx->hostHist.p + (size_t)a * 1024, (const uint64_t*)ctx->hist.p + (size_t)a * 1024,
                                        (size_t)cnt * 1024 * 8, hipMemcpyDeviceToHost, ctx->histStream));
            HIP_TRY(ctx, hipEventRecord(ctx->partEvent[p], ctx->histStream));
        }
        if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->groupTime[grp][2], ctx->stream));
    }
    if (!ctx->workers) {
        const unsigned hw = std::thread::hardware_concurrency();
        ctx->workers = new WorkerPool((int)(hw == 0 ? 1 : (hw > 16 ? 16 : hw)) - 1);
    }
    // Two images per task: enough tasks to balance the threads, few enough to keep the
    // claim counter quiet.
    const int per = 2, tasks = (n + per - 1) / per + parts;
    std::vector<int> status((size_t)tasks, JPEG_HIP_OK);
    std::vector<int64_t> bits((size_t)tasks, 0);
    int taskBase = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < parts; ++p) {
        HIP_TRY(ctx, wait_event(ctx->partEvent[p]));
        if (p == parts - 1) t0 = std::chrono::steady_clock::now();   // what follows is no longer hidden behind the device
        const int a0 = partBegin[p], a1 = partBegin[p + 1];
        const int partTasks = (a1 - a0 + per - 1) / per;
        ctx->workers->run(partTasks, [&](int t) {
            const int a = a0 + t * per, b = a + per < a1 ? a + per : a1;
            batch_tables_worker(ctx, a, b, (const uint64_t*)ctx->hostHist.p, (uint32_t*)ctx->hostTables.p,
                                (uint8_t*)ctx->hostHdr.p, (uint32_t*)ctx->hostHdrLen.p, &status[(size_t)(taskBase + t)],
                                &bits[(size_t)(taskBase + t)]);
        });
        taskBase += partTasks;
        // The part's tables and headers travel while the device is still busy with later parts.
        const size_t cnt = (size_t)(a1 - a0);
        HIP_TRY(ctx, hipMemcpyAsync((uint32_t*)ctx->tables.p + (size_t)a0 * kTabImageWords, (const uint32_t*)ctx->hostTables.p + (size_t)a0 * kTabImageWords,
                                    cnt * kTabImageWords * 4, hipMemcpyHostToDevice, ctx->tableStream));
        HIP_TRY(ctx, hipMemcpyAsync((uint8_t*)ctx->batchHdr.p + (size_t)a0 * kBatchHeaderPitch,
                                    (const uint8_t*)ctx->hostHdr.p + (size_t)a0 * kBatchHeaderPitch, cnt * kBatchHeaderPitch,
                                    hipMemcpyHostToDevice, ctx->tableStream));
        HIP_TRY(ctx, hipMemcpyAsync((uint32_t*)ctx->batchHdrLen.p + a0, (const uint32_t*)ctx->hostHdrLen.p + a0, cnt * 4,
                                    hipMemcpyHostToDevice, ctx->tableStream));
        if (p + 1 == parts / 2 && nHalves == 2) {
            // The first half's tables are on their way: its entropy stage goes behind the last
            // histogram pass and runs while the host builds the tables of the second half.
            for (int t = 0; t < taskBase; ++t) {
                if (status[(size_t)t] != JPEG_HIP_OK) return fail(ctx, status[(size_t)t], "Huffman tables or headers of a batch image could not be built");
                halfBits[0] += bits[(size_t)t];
            }
            tasksOfFirstHalf = taskBase;
            HIP_TRY(ctx, hipEventRecord(ctx->partCounted[0], ctx->tableStream));   // free again: its part was counted long ago
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->partCounted[0], 0));
            const int st = start_half(0, halfBits[0]);
            if (st != JPEG_HIP_OK) return st;
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->partCounted[1 % parts], ctx->tableStream));   // free again: its part was counted long ago
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->partCounted[1 % parts], 0));
    for (int st : status)
        if (st != JPEG_HIP_OK) return fail(ctx, st, "Huffman tables or headers of a batch image could not be built");
    ctx->times[3] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (size_t t = (size_t)tasksOfFirstHalf; t < bits.size(); ++t) halfBits[nHalves - 1] += bits[t];
    }

    if (standard) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tables.p, ctx->hostTables.p, kTabImageWords * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->batchHdr.p, ctx->hostHdr.p, (size_t)n * kBatchHeaderPitch, hipMemcpyHostToDevice,
                                    ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->batchHdrLen.p, ctx->hostHdrLen.p, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    // The last (or only) half.
    {
        const int st = start_half(nHalves - 1, halfBits[nHalves - 1]);
        if (st != JPEG_HIP_OK) return st;
    }
    mark(ctx, 4);
    HIP_TRY(ctx, wait_stream(ctx));
    uint64_t flagsSeen = 0;
    for (int i = 0; i < nHalves; ++i) {
        BatchHalf& h = half[i];
        // Flags of the bit writer as in run_entropy_wait: a stage that was too small (8) or a
        // stream beyond the cleared bytes (4) are cured by a second run.
        for (int attempt = 0; attempt < 3; ++attempt) {
            const uint64_t flags = h.off[2 * (h.cnt + 1) + 1];
            if ((flags & (4 | 8)) == 0 || (flags & (2 | 16)) != 0) break;
            if (flags & 4) h.ws.streamBound = -1;
            if (flags & 8) h.ws.groupTile = h.ws.groupTile > 64 ? 64 : 0;
            const int st = launch_batch_half(ctx, h);
            if (st != JPEG_HIP_OK) return st;
            HIP_TRY(ctx, wait_stream(ctx));
        }
        flagsSeen |= h.off[2 * (h.cnt + 1) + 1];
    }
    if (ctx->profiling) {
        float ms = 0;
        if (standard) {
            if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->times[0] += ms;
        } else {
            for (int grp = 0; grp < (parts >= 4 ? 2 : 1); ++grp) {
                if (hipEventElapsedTime(&ms, ctx->groupTime[grp][0], ctx->groupTime[grp][1]) == hipSuccess) ctx->times[0] += ms;
                if (hipEventElapsedTime(&ms, ctx->groupTime[grp][1], ctx->groupTime[grp][2]) == hipSuccess) ctx->times[1] += ms;
            }
        }
        if (hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]) == hipSuccess) ctx->times[2] += ms;
    }
    if (flagsSeen & 2) return fail(ctx, JPEG_HIP_ERR_HIP, "the bit writer's tiles lost their chain");
    if (flagsSeen & 16) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the Huffman tables have no code for a symbol of an image");
    if (flagsSeen & (4 | 8)) return fail(ctx, JPEG_HIP_ERR_HIP, "the bit writer could not place its stream");
    bool overflow = (flagsSeen & 1) != 0;
    const uint32_t* hdrLen = (const uint32_t*)ctx->hostHdrLen.p;
    for (int i = 0; i < nHalves; ++i) {
        const BatchHalf& h = half[i];
        for (int j = 0; j < h.cnt; ++j) {
            const int img = h.a + j;
            lens[img] = (int64_t)hdrLen[img] + (int64_t)(h.off[j + 1] - h.off[j]) +
                        (int64_t)(h.off[h.cnt + 1 + j + 1] - h.off[h.cnt + 1 + j]);
            if (lens[img] > slotBytes) overflow = true;
        }
    }
    if (overflow) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "an output slot is too small for its image");
    return JPEG_HIP_OK;
}

}  // namespace

namespace {

// Checks a batch request, binds the first image's pixels for the checks that need them and
// sizes the workspace; *passFrames receives the number of frames a pass may hold.
int prepare_batch(jpeg_hip_ctx* ctx, const void* rgb, int sample_format, int64_t stride_bytes, int64_t image_stride_bytes,
                  int count, const jpeg_hip_params* params, int64_t slot_bytes, int maxPass, int* passFrames) {
    if (count < 1) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "batch count must be at least 1");
    if (params->restart_interval != 0 || params->mcu_row_begin != 0 || params->mcu_row_end != 0)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "batch images are whole images without restart intervals");
    if (slot_bytes < kBatchHeaderPitch) return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "output slots are smaller than the headers");
    int st = jpeg_hip_begin(ctx, params);
    if (st != JPEG_HIP_OK) return st;
    st = jpeg_hip_bind_pixels(ctx, rgb, sample_format, stride_bytes);
    if (st != JPEG_HIP_OK) return st;
    const Geometry& g = ctx->geo;
    const int64_t minImage = stride_bytes * (int64_t)(g.height - 1) + (int64_t)g.width * 3 * bytes_per_sample(sample_format);
    if (count > 1 && image_stride_bytes < minImage)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "image stride is smaller than one image");
    if (sample_format == JPEG_HIP_RGBF32 && image_stride_bytes % 4 != 0)
        return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "float pixels must be 4-byte aligned");
    int64_t perPass = kBatchMaxBlocks / g.totalBlocks;
    if (perPass > maxPass) perPass = maxPass;
    if (perPass < 1) perPass = 1;
    if (perPass > count) perPass = count;
    const int pass = (int)perPass;
    st = ensure_workspace(ctx, pass);
    if (st != JPEG_HIP_OK) return st;
    HIP_TRY(ctx, ctx->batchHdr.ensure((size_t)pass * kBatchHeaderPitch));
    HIP_TRY(ctx, ctx->batchHdrLen.ensure((size_t)pass * 4));
    HIP_TRY(ctx, ctx->batchFfStart.ensure((size_t)(pass + 2) * 8));
    HIP_TRY(ctx, ctx->hostHist.ensure((size_t)pass * 1024 * 8));
    HIP_TRY(ctx, ctx->hostTables.ensure((size_t)pass * kTabImageWords * 4));
    HIP_TRY(ctx, ctx->hostHdr.ensure((size_t)pass * kBatchHeaderPitch));
    HIP_TRY(ctx, ctx->hostHdrLen.ensure((size_t)pass * 4));
    HIP_TRY(ctx, ctx->hostOffsets.ensure((size_t)(2 * (pass + 2) + 8) * 8));
    ctx->times[0] = ctx->times[1] = ctx->times[2] = ctx->times[3] = 0;
    *passFrames = pass;
    return JPEG_HIP_OK;
}

void leave_batch(jpeg_hip_ctx* ctx) {   // the context no longer describes a single image
    ctx->hasPixels = false;
    ctx->transformed = false;
    ctx->hasTables = false;
    ctx->stagedScanLen = -1;
}

}  // namespace

extern "C" int jpeg_hip_encode_batch(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                                     int64_t image_stride_bytes, int count, const jpeg_hip_params* params, uint8_t* d_out,
                                     int64_t slot_bytes, int64_t* h_lens) {
    if (!ctx || !params || !d_out || !h_lens) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int pass = 0;
    int st = prepare_batch(ctx, d_rgb, sample_format, stride_bytes, image_stride_bytes, count, params, slot_bytes, 1 << 30,
                           &pass);
    if (st != JPEG_HIP_OK) return st;
    for (int first = 0; first < count; first += pass) {
        const int n = count - first < pass ? count - first : pass;
        st = encode_batch_pass(ctx, (const uint8_t*)d_rgb + (int64_t)first * image_stride_bytes, image_stride_bytes, n,
                               d_out + (int64_t)first * slot_bytes, slot_bytes, h_lens + first);
        if (st != JPEG_HIP_OK) return st;
    }
    leave_batch(ctx);
    return JPEG_HIP_OK;
}

// Frames and files in host memory.  Passes of a few dozen frames go through two pinned
// staging blocks and two device blocks: a helper thread packs and uploads pass i + 1 on a
// second stream while pass i is being encoded, so the GPU only waits for the very first
// upload.
extern "C" int jpeg_hip_encode_batch_host(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes,
                                          int64_t image_stride_bytes, int count, const jpeg_hip_params* params,
                                          uint8_t* h_out, int64_t slot_bytes, int64_t* h_lens) {
    if (!ctx || !params || !h_rgb || !h_out || !h_lens) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ENTER_DEVICE(ctx);
    const int64_t rowBytes = (int64_t)params->width * 3 * bytes_per_sample(sample_format);
    const int64_t packedStride = (rowBytes + 3) / 4 * 4;
    const int64_t packedImage = packedStride * params->height;
    int64_t maxPass = (192LL << 20) / (packedImage > 0 ? packedImage : 1);   // staging of about 192 MiB per block
    if (maxPass < 1) maxPass = 1;
    int pass = 0;
    // The geometry checks want a pixel pointer; the real (device) one is bound pass by pass.
    int st = prepare_batch(ctx, h_rgb, sample_format, stride_bytes, image_stride_bytes, count, params, slot_bytes,
                           (int)(maxPass > (1 << 20) ? (1 << 20) : maxPass), &pass);
    if (st != JPEG_HIP_OK) return st;
    if (!ctx->copyStream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
    for (auto& e : ctx->pipeEvent)
        if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, ctx->pipeStage[i].ensure((size_t)(pass * packedImage)));
        HIP_TRY(ctx, ctx->pipeIn[i].ensure((size_t)(pass * packedImage)));
    }
    HIP_TRY(ctx, ctx->pipeOut.ensure((size_t)(pass * slot_bytes)));
    const int passes = (count + pass - 1) / pass;
    if (!ctx->packers) ctx->packers = new WorkerPool(3);   // with the uploader itself: four threads pack
    std::atomic<int> uploaded(0), encoded(0), failed(0);
    std::thread uploader([&]() {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            failed = 1;
            return;
        }
        for (int i = 0; i < passes && !failed; ++i) {
            while (encoded.load() < i - 1 && !failed) std::this_thread::yield();   // blocks i % 2 are free again
            const int first = i * pass, n = count - first < pass ? count - first : pass;
            uint8_t* stage = (uint8_t*)ctx->pipeStage[i & 1].p;
            // Packing a pass into pinned memory is what bounds this entry point (one thread
            // copies about half as fast as the link transfers): several threads pack, and
            // the pass goes up in pieces so that the link works while the next piece is packed.
            const int pieces = n >= 8 ? 4 : 1;
            bool sent = true;
            for (int piece = 0; piece < pieces && sent; ++piece) {
                const int k0 = n * piece / pieces, k1 = n * (piece + 1) / pieces;
                ctx->packers->run(k1 - k0, [&](int t) {
                    const int k = k0 + t;
                    const uint8_t* src = (const uint8_t*)h_rgb + (int64_t)(first + k) * image_stride_bytes;
                    uint8_t* dst = stage + (int64_t)k * packedImage;
                    if (stride_bytes == packedStride) {
                        memcpy(dst, src, (size_t)(packedStride * (params->height - 1) + rowBytes));
                    } else {
                        for (int y = 0; y < params->height; ++y)
                            memcpy(dst + y * packedStride, src + y * stride_bytes, (size_t)rowBytes);
                    }
                });
                sent = hipMemcpyAsync((uint8_t*)ctx->pipeIn[i & 1].p + (int64_t)k0 * packedImage, stage + (int64_t)k0 * packedImage,
                                      (size_t)((k1 - k0) * packedImage), hipMemcpyHostToDevice, ctx->copyStream) == hipSuccess;
            }
            if (!sent || hipEventRecord(ctx->pipeEvent[i & 1], ctx->copyStream) != hipSuccess) {
                failed = 1;
                return;
            }
            uploaded = i + 1;
        }
    });
    for (int i = 0; i < passes && st == JPEG_HIP_OK; ++i) {
        while (uploaded.load() < i + 1 && !failed) std::this_thread::yield();
        if (failed) {
            st = fail(ctx, JPEG_HIP_ERR_HIP, "upload of a pass failed");
            break;
        }
        const int first = i * pass, n = count - first < pass ? count - first : pass;
        hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->pipeEvent[i & 1], 0);
        if (e != hipSuccess) {
            st = hip_fail(ctx, e, "hipStreamWaitEvent");
            break;
        }
        ctx->rgb = ctx->pipeIn[i & 1].p;
        ctx->stride = packedStride;
        st = encode_batch_pass(ctx, (const uint8_t*)ctx->pipeIn[i & 1].p, packedImage, n, (uint8_t*)ctx->pipeOut.p, slot_bytes,
                               h_lens + first);
        if (st == JPEG_HIP_OK) {
            int64_t longest = 0;
            for (int k = 0; k < n; ++k) longest = std::max(longest, h_lens[first + k]);
            e = hipMemcpy2DAsync(h_out + (int64_t)first * slot_bytes, (size_t)slot_bytes, ctx->pipeOut.p, (size_t)slot_bytes,
                                 (size_t)longest, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) st = hip_fail(ctx, e, "download of a pass");
        }
        if (st != JPEG_HIP_OK) failed = 1;
        encoded = i + 1;
    }
    failed = failed.load() ? 1 : 0;
    if (st != JPEG_HIP_OK) failed = 1;   // lets the uploader leave its wait loops
    uploader.join();
    (void)hipStreamSynchronize(ctx->copyStream);
    leave_batch(ctx);
    return st;
}

// ---- verification decoder ------------------------------------------------------------------
namespace {

void fill_info(const DecodedStream& d, jpeg_hip_image_info* info) {
    info->width = d.width;
    info->height = d.height;
    for (int c = 0; c < 3; ++c) {
        info->h_samp[c] = d.hSamp[c];
        info->v_samp[c] = d.vSamp[c];
    }
    info->restart_interval = d.restartInterval;
    info->total_blocks = d.totalBlocks;
}

// Decodes the file's entropy-coded data on the host and runs the inverse kernels; the
// RGB8 image goes to d_rgb.
int decode_to_device(jpeg_hip_ctx* ctx, const uint8_t* h_jpeg, int64_t len, uint8_t* d_rgb, int64_t stride,
                     int64_t capacity, jpeg_hip_image_info* info) {
    if (!ctx->decoded) ctx->decoded = new (std::nothrow) DecodedStream();
    if (!ctx->decoded) return fail(ctx, JPEG_HIP_ERR_HIP, "out of host memory");
    DecodedStream& d = *ctx->decoded;
    std::string err;
    int st = decode_entropy(h_jpeg, len, d, err);
    if (st != JPEG_HIP_OK) {
        d.coef.clear();
        return fail(ctx, st, err);
    }
    if (info) fill_info(d, info);
    if (stride < (int64_t)d.width * 3 || capacity < stride * (int64_t)(d.height - 1) + (int64_t)d.width * 3)
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "RGB buffer too small for the decoded image");
    DGeom g;
    g.width = d.width;
    g.height = d.height;
    g.mcusPerRow = d.mcusPerRow;
    g.mcuRows = d.mcuRows;
    g.bpm = d.blocksPerMcu;
    g.off1 = d.chanOffset[1];
    g.off2 = d.chanOffset[2];
    g.totalBlocks = d.totalBlocks;
    long long offset = 0;
    for (int c = 0; c < 3; ++c) {
        g.hS[c] = d.hSamp[c];
        g.vS[c] = d.vSamp[c];
        g.wf[c] = d.mcuW / 8 / d.hSamp[c];
        g.hf[c] = d.mcuH / 8 / d.vSamp[c];
        g.planeW[c] = d.mcusPerRow * d.hSamp[c] * 8;
        g.planeH[c] = d.mcuRows * d.vSamp[c] * 8;
        g.planeOffset[c] = offset;
        offset += (long long)g.planeW[c] * g.planeH[c];
    }
    // Divisors per component in zig-zag order, then the cosine table of DCT::directIDCT.
    float tables[192 + 64];
    for (int c = 0; c < 3; ++c)
        for (int z = 0; z < 64; ++z) tables[c * 64 + z] = (float)d.qtables[d.tq[c]][z];
    for (int x = 0; x < 8; ++x)
        for (int i = 0; i < 8; ++i) tables[192 + x * 8 + i] = (float)std::cos(3.14159265358979323846 * (2 * x + 1) * i / 16.0);
    HIP_TRY(ctx, ctx->decCoef.ensure(d.coef.size() * sizeof(int16_t)));
    HIP_TRY(ctx, ctx->decPlanes.ensure((size_t)offset * sizeof(float)));
    HIP_TRY(ctx, ctx->decTables.ensure(sizeof tables));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->decCoef.p, d.coef.data(), d.coef.size() * sizeof(int16_t), hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->decTables.p, tables, sizeof tables, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_inverse_blocks(g, (const int16_t*)ctx->decCoef.p, (const float*)ctx->decTables.p,
                                       (const float*)ctx->decTables.p + 192, (float*)ctx->decPlanes.p, ctx->stream));
    HIP_TRY(ctx, launch_planes_to_rgb(g, (const float*)ctx->decPlanes.p, d_rgb, stride, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // `tables` and d.coef are pageable host memory
    return JPEG_HIP_OK;
}

}  // namespace

extern "C" {

int jpeg_hip_decode_info(const uint8_t* h_jpeg, int64_t len, jpeg_hip_image_info* info) {
    if (!h_jpeg || !info) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    DecodedStream d;
    std::string err;
    const int st = parse_headers(h_jpeg, len, d, err);
    if (st != JPEG_HIP_OK) return st;
    fill_info(d, info);
    return JPEG_HIP_OK;
}

int jpeg_hip_decode(jpeg_hip_ctx* ctx, const uint8_t* h_jpeg, int64_t len, uint8_t* d_rgb, int64_t stride_bytes,
                    int64_t capacity_bytes, jpeg_hip_image_info* info) {
    if (!ctx || !h_jpeg || !d_rgb) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ENTER_DEVICE(ctx);
    return decode_to_device(ctx, h_jpeg, len, d_rgb, stride_bytes, capacity_bytes, info);
}

int jpeg_hip_decode_host(jpeg_hip_ctx* ctx, const uint8_t* h_jpeg, int64_t len, uint8_t* h_rgb, int64_t stride_bytes,
                         int64_t capacity_bytes, jpeg_hip_image_info* info) {
    if (!ctx || !h_jpeg || !h_rgb) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ENTER_DEVICE(ctx);
    jpeg_hip_image_info local;
    int st = jpeg_hip_decode_info(h_jpeg, len, &local);
    if (st != JPEG_HIP_OK) return fail(ctx, st, "unsupported or damaged JPEG headers");
    if (stride_bytes < (int64_t)local.width * 3 ||
        capacity_bytes < stride_bytes * (int64_t)(local.height - 1) + (int64_t)local.width * 3)
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "RGB buffer too small for the decoded image");
    const int64_t tight = (int64_t)local.width * 3;
    HIP_TRY(ctx, ctx->decRgb.ensure((size_t)(tight * local.height)));
    st = decode_to_device(ctx, h_jpeg, len, (uint8_t*)ctx->decRgb.p, tight, tight * local.height, info);
    if (st != JPEG_HIP_OK) return st;
    if (stride_bytes == tight)   // one piece: a two-dimensional copy to pageable memory goes row by row
        HIP_TRY(ctx, hipMemcpyAsync(h_rgb, ctx->decRgb.p, (size_t)(tight * local.height), hipMemcpyDeviceToHost, ctx->stream));
    else
        HIP_TRY(ctx, hipMemcpy2DAsync(h_rgb, (size_t)stride_bytes, ctx->decRgb.p, (size_t)tight, (size_t)tight,
                                      (size_t)local.height, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_HIP_OK;
}

int jpeg_hip_decoded_coefficients(jpeg_hip_ctx* ctx, int16_t* h_out, int64_t capacity_values) {
    if (!ctx || !h_out) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!ctx->decoded || ctx->decoded->coef.empty()) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "nothing decoded yet");
    if (capacity_values < (int64_t)ctx->decoded->coef.size())
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "coefficient buffer too small");
    memcpy(h_out, ctx->decoded->coef.data(), ctx->decoded->coef.size() * sizeof(int16_t));
    return JPEG_HIP_OK;
}

int jpeg_hip_synthetic_images(jpeg_hip_ctx* ctx, int kind, int32_t width, int32_t y0, int32_t y1, uint32_t seed,
                              uint8_t* d_out, int64_t stride_bytes, int64_t image_stride_bytes, int count) {
    if (!ctx || !d_out || (kind != 0 && kind != 1) || width < 1 || y0 < 0 || y1 <= y0 || count < 1 ||
        stride_bytes < (int64_t)width * 3)
        return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, launch_synthetic_images(d_out, kind, width, y1 - y0, y0, seed, stride_bytes, image_stride_bytes, count,
                                         ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    return JPEG_HIP_OK;
}

int jpeg_hip_psnr(jpeg_hip_ctx* ctx, const uint8_t* d_a, int64_t stride_a, const uint8_t* d_b, int64_t stride_b,
                  int32_t width, int32_t height, double* mse, double* psnr) {
    if (!ctx || !d_a || !d_b || width < 1 || height < 1 || stride_a < (int64_t)width * 3 || stride_b < (int64_t)width * 3)
        return JPEG_HIP_ERR_INVALID_ARGUMENT;
    ENTER_DEVICE(ctx);
    HIP_TRY(ctx, ctx->decSum.ensure(8));
    HIP_TRY(ctx, launch_squared_error(d_a, stride_a, d_b, stride_b, width, height, (unsigned long long*)ctx->decSum.p,
                                      ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pin->result, ctx->decSum.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, wait_stream(ctx));
    const double m = (double)ctx->pin->result[0] / ((double)width * height * 3.0);
    if (mse) *mse = m;
    if (psnr) *psnr = m == 0.0 ? INFINITY : 10.0 * std::log10(255.0 * 255.0 / m);
    return JPEG_HIP_OK;
}

}  // extern "C"
