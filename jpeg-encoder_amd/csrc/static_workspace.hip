// jpeg_hip_encode_static: jpeg_hip_encode for callers that cannot allocate device memory.
//
// The device workspace of an encode (coefficients, histograms, Huffman tables, tile states,
// the unstuffed stream and its 0xFF counts) normally lives in buffers the context allocates
// and grows on demand.  Here it is carved out of one array that is part of the library's
// code object, so the entry point never calls hipMalloc or hipFree.  Every GPU the library
// is used on holds its own copy of the array, at its own address, until the process ends.
#include <mutex>

#include "context.h"

using namespace jpeghip;

namespace {

constexpr size_t kStaticWorkspaceBytes = 32u << 20;   // per GPU; see jpeg_hip_static_workspace_bytes()

// In host code this name is only a handle for hipGetSymbolAddress: the array's device
// address is different on every GPU and is looked up once per device below.
static __device__ __align__(256) unsigned char scratch[kStaticWorkspaceBytes];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
unsigned char* g_scratchOnDevice[kMaxDevices] = {};
// One call at a time per GPU: every context on a GPU borrows the same array.
std::mutex g_inUse[kMaxDevices];

}  // namespace

namespace jpeghip {

// Looks up this GPU's copy of the array.  Called by jpeg_hip_create with `device` current;
// loading the code object on that device is what reserves the array's memory there.
hipError_t static_workspace_init(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_lock);
    if (g_scratchOnDevice[device] != nullptr) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (p == nullptr) return hipErrorInvalidSymbol;
    g_scratchOnDevice[device] = (unsigned char*)p;
    return hipSuccess;
}

}  // namespace jpeghip

namespace {

unsigned char* scratch_on(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_lock);
    return g_scratchOnDevice[device];
}

int fail(jpeg_hip_ctx* ctx, int status, const char* msg) {
    ctx->err = msg;
    return status;
}

// The context's workspace buffers, swapped for slices of the array and put back on the way out.
struct Loan {
    static constexpr int kCount = 12;
    jpeg_hip_ctx* ctx;
    DevBuf* slot[kCount];
    DevBuf saved[kCount];
    unsigned long long savedExpected;
    explicit Loan(jpeg_hip_ctx* c) : ctx(c) {
        DevBuf* all[kCount] = {&c->coef, &c->qtab, &c->hist, &c->tables, &c->ivOff, &c->ffCount, &c->groupSum, &c->result,
                               &c->unstuffed, &c->tileState, &c->histDone, &c->gmask};
        for (int i = 0; i < kCount; ++i) {
            slot[i] = all[i];
            saved[i] = *all[i];
        }
        savedExpected = c->histExpected;
    }
    void lend(DevBuf& b, unsigned char*& cursor, size_t bytes) {
        b.p = cursor;
        b.cap = bytes;
        b.borrowed = true;
        cursor += (bytes + 255) / 256 * 256;
    }
    ~Loan() {
        for (int i = 0; i < kCount; ++i) *slot[i] = saved[i];
        ctx->histExpected = savedExpected;
        ctx->staticWorkspace = false;
        // The context's own buffers hold nothing of the image just encoded (and its own copy
        // of the divisors may be older than the pinned one): the next call starts afresh.
        ctx->qtabOnDevice = false;
        ctx->begun = ctx->hasPixels = ctx->transformed = ctx->hasTables = false;
        ctx->histOnHost = false;
        ctx->tablesPending = false;
        ctx->stagedScanLen = -1;
        ctx->streamBits = -1;
    }
};

size_t round256(size_t n) { return (n + 255) / 256 * 256; }

}  // namespace

extern "C" {

int64_t jpeg_hip_static_workspace_bytes(void) { return (int64_t)kStaticWorkspaceBytes; }

int jpeg_hip_encode_static(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                           const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !params || !d_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->scanInFlight) return fail(ctx, JPEG_HIP_ERR_BAD_STATE, "jpeg_hip_finish_band_wait has not been called");
    unsigned char* base = scratch_on(ctx->device);
    if (!base) return fail(ctx, JPEG_HIP_ERR_HIP, "the static workspace was not found on this GPU");
    Geometry g;
    const int gs = make_geometry(*params, g);
    if (gs != JPEG_HIP_OK) return fail(ctx, gs, "invalid image parameters");

    // Everything but the unstuffed stream has a size fixed by the geometry; the stream (with
    // the 0xFF counts that go with it, 4 bytes per 4 KiB chunk and 8 per 64 chunks) gets the rest.
    const WorkspaceSizes fixed = workspace_sizes(g, 1, 0);
    const size_t fixedBytes = round256(fixed.coef) + round256(fixed.gmask) + round256(fixed.qtab) + round256(fixed.hist) + round256(fixed.tables) +
                              round256(fixed.ivOff) + round256(fixed.result) + round256(fixed.tileState) + round256(64 + 8 * kHistogramMaxGrid) /* histDone */;
    const size_t margin = 64u << 10;   // rounding of the slices, the counts of a partial chunk
    if (fixedBytes + margin + (1u << 20) > kStaticWorkspaceBytes)
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL,
                    "image too large for the static workspace (32 MiB: 128 bytes per 8x8 block plus the entropy-coded stream)");
    const size_t rest = kStaticWorkspaceBytes - fixedBytes - margin;
    const int64_t streamLimit = (int64_t)((double)rest / (1.0 + 4.0 / kStuffChunk + 8.0 / (64.0 * kStuffChunk)));
    const WorkspaceSizes sz = workspace_sizes(g, 1, streamLimit);
    if (params->standard_huffman && sz.unstuffedCap < workspace_sizes(g, 1).unstuffedCap)
        return fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL,
                    "with the standard Huffman tables the static workspace has to hold the worst-case stream (216 bytes per block)");

    hipError_t he = hipSuccess;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    if (previous != ctx->device) {
        he = hipSetDevice(ctx->device);
        if (he != hipSuccess) return fail(ctx, JPEG_HIP_ERR_HIP, "hipSetDevice failed");
    }

    int st;
    {
        // Held until the call's kernels have finished (the entry point waits for them): a second
        // caller on this GPU, from any context or thread, waits here instead of racing on the array.
        std::lock_guard<std::mutex> inUse(g_inUse[ctx->device]);
        Loan loan(ctx);
        unsigned char* cursor = base;
        loan.lend(ctx->coef, cursor, sz.coef);
        loan.lend(ctx->gmask, cursor, sz.gmask);
        loan.lend(ctx->qtab, cursor, sz.qtab);
        loan.lend(ctx->hist, cursor, sz.hist);
        loan.lend(ctx->tables, cursor, sz.tables);
        loan.lend(ctx->ivOff, cursor, sz.ivOff);
        loan.lend(ctx->result, cursor, sz.result);
        loan.lend(ctx->tileState, cursor, sz.tileState);
        loan.lend(ctx->histDone, cursor, 64 + 8 * kHistogramMaxGrid);
        loan.lend(ctx->ffCount, cursor, sz.ffCount);
        loan.lend(ctx->groupSum, cursor, sz.groupSum);
        loan.lend(ctx->unstuffed, cursor, sz.unstuffed);
        if ((size_t)(cursor - base) > kStaticWorkspaceBytes) {
            st = fail(ctx, JPEG_HIP_ERR_BUFFER_TOO_SMALL, "image too large for the static workspace");
        } else {
            ctx->staticWorkspace = true;
            ctx->qtabOnDevice = false;   // the borrowed table slice holds the last call's divisors, whoever made it
            // The array keeps what the last call left.  The histogram kernel's count-off word must
            // start at zero (the histograms, tile states, result record and the stream's reachable
            // part are cleared by the kernels that use them).
            ctx->histExpected = 0;
            he = hipMemsetAsync(ctx->histDone.p, 0, 64, ctx->stream);
            if (he != hipSuccess) {
                st = fail(ctx, JPEG_HIP_ERR_HIP, "clearing the static workspace failed");
            } else {
                st = jpeg_hip_analyze_band(ctx, d_rgb, sample_format, stride_bytes, params, nullptr);
                if (st == JPEG_HIP_OK) st = jpeg_hip_finish_band(ctx, nullptr, d_out, out_capacity, out_len);
                // Nothing of this call may still be running on the array when the next call starts.
                if (st != JPEG_HIP_OK) (void)hipStreamSynchronize(ctx->stream);
            }
        }
    }
    if (previous >= 0 && previous != ctx->device) (void)hipSetDevice(previous);
    return st;
}

}  // extern "C"
