// The encoder context and its buffers, shared by the host translation units of
// libjpegenc_hip.  Not part of the public ABI.
#ifndef JPEGHIP_CONTEXT_H
#define JPEGHIP_CONTEXT_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "decode.h"
#include "internal.h"
#include "kernels.h"

namespace jpeghip {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;   // p is a slice of memory owned elsewhere: never freed, never grown
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (borrowed) return hipErrorOutOfMemory;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        // Grow with headroom so a stream of slightly different images does not reallocate.
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
    }
};

// Growable pinned host block.
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

constexpr int kHeaderCapacity = 2048;

struct Pinned {
    float qtab[3 * 192];   // divisors, then the high and low parts of their reciprocals
    uint64_t hist[4 * 256];
    uint32_t tables[kTabImageWords];    // table image (kernels.h) and header travel to the device in one copy
    uint8_t header[kHeaderCapacity];
    uint64_t result[4];
    uint64_t exchangeFlag;              // failure flags of a sharded encode, to and from the device
    uint64_t tileStats[4];              // HistogramExport::hostStats of the last histogram pass
};

// Host threads that stay around between batch passes: starting sixteen threads costs more
// than the table work they are started for.
class WorkerPool {
public:
    explicit WorkerPool(int threads) {
        for (int i = 0; i < threads; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
            stopFlag_.store(true);
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    // Calls fn(i) for every i in [0, n), on the pool's threads and the caller's, and returns
    // when all calls have returned.  One run at a time.
    void run(int n, const std::function<void(int)>& fn) {
        bool asleep;
        {
            std::lock_guard<std::mutex> lock(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            busy_.store((int)threads_.size());
            asleep = sleepers_ > 0;
            generation_.fetch_add(1);
        }
        if (asleep) wake_.notify_all();
        for (int i; (i = next_.fetch_add(1)) < n;) fn(i);
        while (busy_.load() != 0) std::this_thread::yield();
        fn_ = nullptr;
    }

private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // Runs of one batch pass follow each other within a fraction of a millisecond, and
            // waking a sleeping thread takes about as long as its share of a run: stay awake
            // for a while after a run before going to sleep.
            const auto idleSince = std::chrono::steady_clock::now();
            while (generation_.load() == seen && !stopFlag_.load() &&
                   std::chrono::steady_clock::now() - idleSince < std::chrono::microseconds(500))
                std::this_thread::yield();
            const std::function<void(int)>* fn;
            int n;
            {
                std::unique_lock<std::mutex> lock(m_);
                ++sleepers_;
                wake_.wait(lock, [&] { return stop_ || generation_.load() != seen; });
                --sleepers_;
                if (stop_) return;
                seen = generation_.load();
                fn = fn_;
                n = n_;
            }
            for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
            busy_.fetch_sub(1);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, sleepers_ = 0;
    std::atomic<int> next_{0}, busy_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stopFlag_{false};
    bool stop_ = false;
};

// Bytes of every buffer of the device workspace for `images` images of geometry g (1 outside
// batch mode; in a batch every image is one interval of the combined stream).
// unstuffedLimit >= 0 caps the unstuffed stream below its worst case (a workspace of fixed
// size); the caller then has to check the stream's real bound against unstuffedCap.
struct WorkspaceSizes {
    int64_t unstuffedCap, maxChunks, tiles;
    size_t coef, gmask, qtab, hist, tables, ivOff, ffCount, groupSum, result, unstuffed, tileState;
};

inline WorkspaceSizes workspace_sizes(const Geometry& g, int images, int64_t unstuffedLimit = -1) {
    WorkspaceSizes w;
    const int64_t nb = g.totalBlocks * images;
    const int64_t intervals = images > 1 ? images : g.intervals;
    // Entropy-coded bits plus the RSTn/EOI markers the bit writer places in the stream.
    w.unstuffedCap = (worst_case_scan_bytes(nb) + 2 * intervals + 2 + kStuffChunk + 63) / 16 * 16;
    if (unstuffedLimit >= 0 && unstuffedLimit / 16 * 16 < w.unstuffedCap) w.unstuffedCap = unstuffedLimit / 16 * 16;
    w.maxChunks = (w.unstuffedCap + kStuffChunk - 1) / kStuffChunk;
    // One state word per tile of the bit writer; its smallest tiles are 16 blocks, in fours per image.
    // (and room for what the dense bit writer keeps behind the states of its tiles of 256 blocks)
    w.tiles = (g.totalBlocks + 15) / 16 * images;
    const int64_t denseWords = dense_state_words((g.totalBlocks + 255) / 256 * images);
    if (w.tiles < denseWords) w.tiles = denseWords;
    w.tiles += 4 * images;
    w.coef = (size_t)nb * 128;
    w.gmask = (size_t)nb + 64;   // one byte per block, read in dwords
    w.qtab = sizeof(Pinned::qtab);
    w.hist = 1024 * sizeof(uint64_t) * (size_t)images;
    w.tables = kTabImageWords * sizeof(uint32_t) * (size_t)images + kHeaderCapacity;
    w.ivOff = (size_t)(intervals + 1) * 8;
    w.ffCount = (size_t)w.maxChunks * 4;
    w.groupSum = (size_t)(w.maxChunks / 64 + 2) * 8;
    w.result = 4 * 8;
    w.unstuffed = (size_t)w.unstuffedCap;
    w.tileState = (size_t)(w.tiles + 1) * 8;
    return w;
}

// This GPU's copy of the static workspace array (static_workspace.hip); called with `device` current.
hipError_t static_workspace_init(int device);

constexpr int kBatchParts = 8;   // pieces a batch pass is cut into to hide the host's table building

}  // namespace jpeghip

struct jpeg_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err = "";
    jpeg_hip_params prm;
    jpeghip::Geometry geo;
    jpeghip::KGeom kg;
    bool begun = false, hasPixels = false, transformed = false, hasTables = false, qtabOnDevice = false;
    bool scanInFlight = false;                   // between jpeg_hip_finish_band_start and _wait
    bool staticWorkspace = false;                // the workspace buffers are borrowed slices of fixed size: nothing may be allocated
    int64_t stagedScanLen = -1;   // scan of the current coefficients and tables held in `staging`, or -1
    uint8_t* emitOut = nullptr;                  // arguments of the entropy stage in flight, for a second run
    int64_t emitPrefix = 0, emitCapacity = 0;
    int64_t streamBits = -1;      // entropy-coded bits the installed tables spend on the histograms they came from, or -1
    const void* rgb = nullptr;
    int fmt = JPEG_HIP_RGB8;
    int64_t stride = 0;
    jpeghip::DevBuf repacked;                             // byte pixels whose rows did not start on 4-byte boundaries, rows realigned
    jpeghip::DevBuf pixels, coef, gmask, qtab, hist, tables, ivOff, ffCount, groupSum, result, unstuffed,
        staging, batchHdr, batchHdrLen, batchFfStart, tileState, decCoef, decPlanes, decTables, decRgb, decSum, pixelsF, pipeIn[2], pipeOut;
    jpeghip::DecodedStream* decoded = nullptr;   // last file given to the verification decoder
    jpeghip::WorkerPool* workers = nullptr;               // created by the first batch pass
    jpeghip::WorkerPool* packers = nullptr;               // pack host pixels into pinned memory (uploads, jpeg_hip_encode_batch_host)
    jpeghip::HostBuf uploadStage[2];                      // pinned blocks jpeg_hip_upload_pixels sends its pieces from
    hipEvent_t uploadEvent[2] = {nullptr, nullptr};
    jpeghip::DevBuf joined;                               // all blocks with their halves joined (jpeg_hip_coefficients_device)
    jpeghip::DevBuf gather;                               // one channel's coefficients, contiguous (jpeg_hip_get_coefficients)
    jpeghip::DevBuf tileSlots;                            // staged bits of the bit writer's tiles (EntropyWorkspace::tileSlots)
    jpeghip::DevBuf exchange;                             // histograms + failure flags a sharded encode all-reduces (sharded.cpp)
    hipEvent_t orderEvent = nullptr;             // orders the context's stream against a caller's (jpeg_hip_wait_for_stream)
    jpeghip::DevBuf histDone;                             // counter of the histogram kernel's host export
    unsigned long long histExpected = 0;         // value of that counter once every launch so far has finished
    bool histOnHost = false;                     // pin->hist holds the current band's histograms as counted
    bool tileStatsPending = false;               // the histogram pass in flight (or done) leaves pin->tileStats
    bool tileStatsValid = false;                 // pin->tileStats describe the current coefficients and have arrived
    bool tablesPending = false;                  // pin->tables (+ pendingHeaderLen header bytes) are not on the device yet
    int pendingHeaderLen = 0;
    int nGains = 0;                              // jpeg_hip_channel_gain calls since begin()
    struct {
        int channel, afterSubsampling;
        float value;
    } gains[JPEG_HIP_MAX_GAINS];
    jpeghip::HostBuf hostHist, hostTables, hostHdr, hostHdrLen, hostOffsets, pipeStage[2];
    hipStream_t copyStream = nullptr;
    hipEvent_t pipeEvent[2] = {nullptr, nullptr};
    hipStream_t histStream = nullptr;            // carries a batch pass's histograms to the host
    hipStream_t tableStream = nullptr;           // ... and its tables and headers to the device
    hipEvent_t partCounted[jpeghip::kBatchParts] = {};    // a part of a batch pass has its histograms
    hipEvent_t partEvent[jpeghip::kBatchParts] = {};      // ... and they are on the host
    hipEvent_t groupTime[2][3] = {};             // profiling: before a group's transform, behind it, behind its histogram parts
    jpeghip::EntropyWorkspace ws;
    jpeghip::Pinned* pin = nullptr;
    uint8_t lengths[4][256];
    uint16_t codes[4][256];
    bool profiling = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float times[4] = {0, 0, 0, 0};
    double hostTimeline[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // jpeg_hip_get_host_timeline
    double sumTimes[4] = {0, 0, 0, 0};                   // jpeg_hip_get_profile_sums: `times` added up
    double sumTimeline[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // ... the seven intervals of hostTimeline, and the gap between images
    double lastResultSeen = 0;                           // hostTimeline[7] of the image before
    int64_t sumCount = 0;
    int64_t sumEntropyRuns = 0;                          // launches of the entropy stage (more than sumCount: some images needed a second run)
};

#endif
