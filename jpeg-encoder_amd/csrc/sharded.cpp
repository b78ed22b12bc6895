// Sharded encode over RCCL: one band of MCU rows per GPU, one all-reduce of the four symbol
// histograms (8 KiB) plus a word of failure flags as the only exchange between the ranks.
// RCCL is bound at run time (dlopen of librccl.so.1 on first use): the library and everything
// that needs only one GPU load without it.  No framework sits between the library and the
// communicator.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; nothing of it is linked

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "context.h"

using namespace jpeghip;

namespace {

int fail(jpeg_hip_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return status;
}

// The six RCCL entry points the library uses, looked up once.
struct Rccl {
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclCommCuDevice) commCuDevice = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
    std::string error;   // why it is not usable, or empty
};

const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // A copy the process has already loaded (e.g. the one a framework ships) is found by its
        // soname; otherwise the dynamic linker's search path, then the ROCm installation.
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char* root = getenv("ROCM_PATH");
            const std::string path = std::string(root && *root ? root : "/opt/rocm") + "/lib/librccl.so.1";
            h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) {
            const char* why = dlerror();
            r.error = std::string("librccl is not available: ") + (why ? why : "dlopen failed");
            return;
        }
        r.getUniqueId = (decltype(r.getUniqueId))dlsym(h, "ncclGetUniqueId");
        r.commInitRank = (decltype(r.commInitRank))dlsym(h, "ncclCommInitRank");
        r.commDestroy = (decltype(r.commDestroy))dlsym(h, "ncclCommDestroy");
        r.allReduce = (decltype(r.allReduce))dlsym(h, "ncclAllReduce");
        r.commCuDevice = (decltype(r.commCuDevice))dlsym(h, "ncclCommCuDevice");
        r.getErrorString = (decltype(r.getErrorString))dlsym(h, "ncclGetErrorString");
        if (!r.getUniqueId || !r.commInitRank || !r.commDestroy || !r.allReduce || !r.commCuDevice || !r.getErrorString)
            r.error = "librccl lacks an entry point the library needs";
    });
    return r;
}

int nccl_fail(jpeg_hip_ctx* ctx, ncclResult_t r, const char* what) {
    return fail(ctx, JPEG_HIP_ERR_HIP, std::string(what) + ": " + rccl().getErrorString(r));
}

// The communicator has to live on the context's GPU.
int check_comm(jpeg_hip_ctx* ctx, ncclComm_t comm) {
    if (!rccl().error.empty()) return fail(ctx, JPEG_HIP_ERR_HIP, rccl().error);
    int device = -1;
    const ncclResult_t r = rccl().commCuDevice(comm, &device);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclCommCuDevice");
    if (device != ctx->device) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the communicator belongs to another GPU than the context");
    return JPEG_HIP_OK;
}

struct DeviceGuard {
    int previous = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        if (previous != device) ok = hipSetDevice(device) == hipSuccess;
        else previous = -1;
    }
    ~DeviceGuard() {
        if (previous >= 0) (void)hipSetDevice(previous);
    }
};

// Words exchanged per image: the four histograms and, behind them, the number of ranks whose
// local step failed.
constexpr size_t kExchangeWords = 4 * 256 + 1;

// The exchange buffer belongs to the context, not to the per-image workspace: a rank whose image
// cannot even be set up still has something to send.  Its last word (the failure flags) is zero
// whenever no exchange is under way.
int ensure_exchange(jpeg_hip_ctx* ctx) {
    if (ctx->exchange.p) return JPEG_HIP_OK;
    if (ctx->exchange.ensure(kExchangeWords * 8) != hipSuccess ||
        hipMemsetAsync(ctx->exchange.p, 0, kExchangeWords * 8, ctx->stream) != hipSuccess)
        return fail(ctx, JPEG_HIP_ERR_HIP, "no device memory for the exchange buffer");
    return JPEG_HIP_OK;
}

}  // namespace

extern "C" {

int jpeg_hip_comm_unique_id(uint8_t id[128]) {
    if (!id) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    if (!rccl().error.empty()) return JPEG_HIP_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
    ncclUniqueId uid;
    if (rccl().getUniqueId(&uid) != ncclSuccess) return JPEG_HIP_ERR_HIP;
    memcpy(id, &uid, sizeof uid);
    return JPEG_HIP_OK;
}

int jpeg_hip_comm_create(const uint8_t id[128], int rank, int world_size, int device_id, void** out_comm) {
    if (!id || !out_comm || world_size < 1 || rank < 0 || rank >= world_size) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    *out_comm = nullptr;
    if (!rccl().error.empty()) return JPEG_HIP_ERR_HIP;
    DeviceGuard guard(device_id);
    if (!guard.ok) return JPEG_HIP_ERR_HIP;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = rccl().commInitRank(&comm, world_size, uid, rank);
    if (r != ncclSuccess) return JPEG_HIP_ERR_HIP;
    *out_comm = comm;
    return JPEG_HIP_OK;
}

void jpeg_hip_comm_destroy(void* comm) {
    if (comm && rccl().error.empty()) (void)rccl().commDestroy((ncclComm_t)comm);
}

int jpeg_hip_allreduce_histograms(jpeg_hip_ctx* ctx, void* rccl_comm, uint64_t* d_hist) {
    if (!ctx || !rccl_comm || !d_hist) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    const int st = check_comm(ctx, (ncclComm_t)rccl_comm);
    if (st != JPEG_HIP_OK) return st;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(ctx, JPEG_HIP_ERR_HIP, "hipSetDevice failed");
    // In place, on the context's stream: ordered behind the kernel that counted and in front of
    // whatever the context launches next.
    const ncclResult_t r = rccl().allReduce(d_hist, d_hist, 4 * 256, ncclUint64, ncclSum, (ncclComm_t)rccl_comm, ctx->stream);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclAllReduce");
    ctx->histOnHost = false;   // the band's own counts are not the image's
    return JPEG_HIP_OK;
}

int jpeg_hip_allreduce_status(jpeg_hip_ctx* ctx, void* rccl_comm, int local_status, int* failed_ranks) {
    if (!ctx || !rccl_comm || !failed_ranks) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int st = check_comm(ctx, (ncclComm_t)rccl_comm);
    if (st != JPEG_HIP_OK) return st;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(ctx, JPEG_HIP_ERR_HIP, "hipSetDevice failed");
    st = ensure_exchange(ctx);
    if (st != JPEG_HIP_OK) return st;
    uint64_t* word = (uint64_t*)ctx->exchange.p + 4 * 256;
    ctx->pin->exchangeFlag = local_status == JPEG_HIP_OK ? 0 : 1;
    if (hipMemcpyAsync(word, &ctx->pin->exchangeFlag, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(ctx, JPEG_HIP_ERR_HIP, "hipMemcpyAsync failed");
    const ncclResult_t r = rccl().allReduce(word, word, 1, ncclUint64, ncclSum, (ncclComm_t)rccl_comm, ctx->stream);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclAllReduce");
    if (hipMemcpyAsync(&ctx->pin->exchangeFlag, word, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, JPEG_HIP_ERR_HIP, "reading the agreed status failed");
    *failed_ranks = (int)ctx->pin->exchangeFlag;
    (void)hipMemsetAsync(word, 0, 8, ctx->stream);
    return JPEG_HIP_OK;
}

int jpeg_hip_encode_sharded(jpeg_hip_ctx* ctx, void* rccl_comm, const void* d_rgb_band, int sample_format,
                            int64_t stride_bytes, const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity,
                            int64_t* out_len) {
    // Without a context or a communicator there is nothing to take part in the exchange with.
    if (!ctx || !rccl_comm) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int st = check_comm(ctx, (ncclComm_t)rccl_comm);
    if (st != JPEG_HIP_OK) return st;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(ctx, JPEG_HIP_ERR_HIP, "hipSetDevice failed");
    st = ensure_exchange(ctx);
    if (st != JPEG_HIP_OK) return st;
    uint64_t* ex = (uint64_t*)ctx->exchange.p;

    // This rank's half: transform and histograms of its band, counted into the exchange buffer.
    int local = (!params || !d_out || !out_len) ? JPEG_HIP_ERR_INVALID_ARGUMENT : JPEG_HIP_OK;
    if (local == JPEG_HIP_OK) local = jpeg_hip_analyze_band(ctx, d_rgb_band, sample_format, stride_bytes, params, ex);
    const std::string localError = ctx->err;
    if (local != JPEG_HIP_OK) {
        // A rank that fails still takes part in the collective -- the others would wait for it
        // for ever -- with empty histograms and its flag raised.
        ctx->pin->exchangeFlag = 1;
        if (hipMemsetAsync(ex, 0, 4 * 256 * 8, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ex + 4 * 256, &ctx->pin->exchangeFlag, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return fail(ctx, JPEG_HIP_ERR_HIP, "could not prepare the exchange after a local failure: " + localError);
    }
    const ncclResult_t r = rccl().allReduce(ex, ex, kExchangeWords, ncclUint64, ncclSum, (ncclComm_t)rccl_comm, ctx->stream);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclAllReduce");
    // Sums and flags to the host in one copy; the tables are built from there.
    ctx->histOnHost = false;
    if (hipMemcpyAsync(ctx->pin->hist, ex, 4 * 256 * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(&ctx->pin->exchangeFlag, ex + 4 * 256, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, JPEG_HIP_ERR_HIP, "reading the summed histograms failed");
    const uint64_t failedRanks = ctx->pin->exchangeFlag;
    if (failedRanks != 0) {
        (void)hipMemsetAsync(ex + 4 * 256, 0, 8, ctx->stream);   // the flag word starts the next image at zero
        if (local != JPEG_HIP_OK) return fail(ctx, local, localError);
        return fail(ctx, JPEG_HIP_ERR_PEER_FAILED, std::to_string(failedRanks) + " other rank(s) of the sharded encode failed; no piece was written");
    }
    // Identical tables on every rank, then this band's piece of the file.
    ctx->histOnHost = true;   // pin->hist holds what the tables are to be built from
    return jpeg_hip_finish_band(ctx, nullptr, d_out, out_capacity, out_len);
}

}  // extern "C"
