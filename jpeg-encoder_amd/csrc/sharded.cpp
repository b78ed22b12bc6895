// Sharded encode over RCCL: one band of MCU rows per GPU, one all-reduce of the four symbol
// histograms (8 KiB) as the only exchange between the ranks.  RCCL is linked directly; no
// framework sits between the library and the communicator.
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "context.h"

using namespace jpeghip;

namespace {

int fail(jpeg_hip_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return status;
}

int nccl_fail(jpeg_hip_ctx* ctx, ncclResult_t r, const char* what) {
    return fail(ctx, JPEG_HIP_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r));
}

// The communicator has to live on the context's GPU.
int check_comm(jpeg_hip_ctx* ctx, ncclComm_t comm) {
    int device = -1;
    const ncclResult_t r = ncclCommCuDevice(comm, &device);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclCommCuDevice");
    if (device != ctx->device) return fail(ctx, JPEG_HIP_ERR_INVALID_ARGUMENT, "the communicator belongs to another GPU than the context");
    return JPEG_HIP_OK;
}

}  // namespace

extern "C" {

int jpeg_hip_comm_unique_id(uint8_t id[128]) {
    if (!id) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
    ncclUniqueId uid;
    if (ncclGetUniqueId(&uid) != ncclSuccess) return JPEG_HIP_ERR_HIP;
    memcpy(id, &uid, sizeof uid);
    return JPEG_HIP_OK;
}

int jpeg_hip_comm_create(const uint8_t id[128], int rank, int world_size, int device_id, void** out_comm) {
    if (!id || !out_comm || world_size < 1 || rank < 0 || rank >= world_size) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    *out_comm = nullptr;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    if (hipSetDevice(device_id) != hipSuccess) return JPEG_HIP_ERR_HIP;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world_size, uid, rank);
    if (previous >= 0 && previous != device_id) (void)hipSetDevice(previous);
    if (r != ncclSuccess) return JPEG_HIP_ERR_HIP;
    *out_comm = comm;
    return JPEG_HIP_OK;
}

void jpeg_hip_comm_destroy(void* comm) {
    if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

int jpeg_hip_allreduce_histograms(jpeg_hip_ctx* ctx, void* rccl_comm, uint64_t* d_hist) {
    if (!ctx || !rccl_comm || !d_hist) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    const int st = check_comm(ctx, (ncclComm_t)rccl_comm);
    if (st != JPEG_HIP_OK) return st;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    if (previous != ctx->device && hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, JPEG_HIP_ERR_HIP, "hipSetDevice failed");
    // In place, on the context's stream: ordered behind the kernel that counted and in front of
    // whatever the context launches next.
    const ncclResult_t r = ncclAllReduce(d_hist, d_hist, 4 * 256, ncclUint64, ncclSum, (ncclComm_t)rccl_comm, ctx->stream);
    if (previous >= 0 && previous != ctx->device) (void)hipSetDevice(previous);
    if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclAllReduce");
    ctx->histOnHost = false;   // the band's own counts are not the image's
    return JPEG_HIP_OK;
}

int jpeg_hip_encode_sharded(jpeg_hip_ctx* ctx, void* rccl_comm, const void* d_rgb_band, int sample_format,
                            int64_t stride_bytes, const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity,
                            int64_t* out_len) {
    if (!ctx || !rccl_comm || !params || !d_out || !out_len) return JPEG_HIP_ERR_INVALID_ARGUMENT;
    int st = check_comm(ctx, (ncclComm_t)rccl_comm);
    if (st != JPEG_HIP_OK) return st;
    // Transform and histograms of this band, counted into the context's own buffer.
    st = jpeg_hip_analyze_band(ctx, d_rgb_band, sample_format, stride_bytes, params, nullptr);
    if (st != JPEG_HIP_OK) return st;
    uint64_t* hist = (uint64_t*)ctx->hist.p;
    if (!params->standard_huffman) {   // fixed tables: nothing was counted, nothing to exchange
        st = jpeg_hip_allreduce_histograms(ctx, rccl_comm, hist);
        if (st != JPEG_HIP_OK) return st;
    }
    // Identical tables on every rank, then this band's piece of the file.
    return jpeg_hip_finish_band(ctx, params->standard_huffman ? nullptr : hist, d_out, out_capacity, out_len);
}

}  // extern "C"
