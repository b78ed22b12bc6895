/*
 * jpegenc_hip.h -- C ABI of the MI355X (gfx950) baseline-JPEG encode hot path.
 *
 * The library replaces what sits behind the reference's Encoder / Image / ImageLoader
 * classes (Irame/JPEG-Encoder, tree "JPEG-Encoder-SIMD/"): colour conversion, chroma
 * subsampling, 8x8 AAN DCT, quantization, zig-zag, DC/AC symbol generation, per-image
 * length-limited Huffman tables and the escaped bit stream.  Plain C types only, caller
 * owned buffers, integer status codes, no exceptions across the boundary.  Every entry
 * point names the reference interface it stands in for.
 *
 * Pointer naming: d_* is device (HBM) memory of the context's GPU, h_* is host memory.
 * All device work of a context runs on one HIP stream; entry points that return data
 * to the host synchronise that stream before they return.
 */
#ifndef JPEGENC_HIP_H
#define JPEGENC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jpeg_hip_ctx jpeg_hip_ctx;

enum {
    JPEG_HIP_OK = 0,
    JPEG_HIP_ERR_INVALID_ARGUMENT = -1,
    JPEG_HIP_ERR_HIP = -2,              /* a HIP runtime call failed; see jpeg_hip_last_error */
    JPEG_HIP_ERR_BUFFER_TOO_SMALL = -3,
    JPEG_HIP_ERR_BAD_STATE = -4,        /* staged entry points called out of order */
    JPEG_HIP_ERR_NO_DEVICE = -5,
    JPEG_HIP_ERR_PEER_FAILED = -6       /* sharded encode: this rank was fine, another rank of the job failed */
};

/* ReductionMethod, SamplingScheme.h:4-7 */
enum { JPEG_HIP_SUBSAMPLING = 0, JPEG_HIP_AVERAGE = 1 };

/* Encoder::CoefficientType, Encoder.h:9-12 */
enum { JPEG_HIP_DC = 0, JPEG_HIP_AC = 1 };

/* Sample layout of the interleaved RGB input. */
enum {
    JPEG_HIP_RGB8 = 0,   /* bytes; value/255.0f is applied as ImageLoader::LoadPNG does (ImageLoader.cpp:190-193) */
    JPEG_HIP_RGBF32 = 1  /* floats in [0,1], the layout Image::setRawPixelData takes (Image.h:89) */
};

/* Transform stages, one bit per reference call that the fused kernel absorbs. */
enum {
    JPEG_HIP_STAGE_YCBCR = 1,     /* Encoder::convertToYCbCr, Encoder.cpp:15-23 */
    JPEG_HIP_STAGE_SUBSAMPLE = 2, /* Encoder::reduceResolutionBySchema, Encoder.cpp:157-166 */
    JPEG_HIP_STAGE_ALL = 3
};

/*
 * Geometry and tables of one image (or one band of a large image).
 *
 * sampling: SamplingDefinition (SamplingScheme.h:25-44) -- per channel Y, Cb, Cr the four
 *   ChannelReductionOptions fields {widthFactor, widthMethod, heightFactor, heightMethod}.
 * qtable:   QTableSet (QuantizationTables.h:11-27) -- three tables of 64 floats in natural
 *   (row-major) order, each in [1, 256): DQT stores (byte)q (JPEGSegments.h:152-167).  Integer
 *   divisors take the fast kernels; any other value is divided by as it is, in a slower kernel.
 * restart_interval: MCUs per restart interval; 0 reproduces the reference (one scan, no
 *   DRI/RSTn).  No counterpart in the reference.
 * pad_ones: 0 pads the last byte of the scan (and of every restart interval) with zero
 *   bits as BitBuffer::fillToByteBorder does (BitBuffer.cpp:12-16); 1 pads with one bits.
 * mcu_row_begin / mcu_row_end: the band of MCU rows this context encodes, for images
 *   sharded over several GPUs; both 0 selects the whole image.  A band must start on a
 *   restart-interval boundary.  The RGB pointer handed to the encode calls addresses
 *   image row mcu_row_begin * (MCU height).  No counterpart in the reference.
 * standard_huffman:
 *   0: Huffman tables optimised per image from its symbol histograms, as the reference
 *   always does (Encoder::createHuffmanTable, Encoder.cpp:368-416).  1: the typical tables
 *   of ITU-T T.81 Annex K.3 (K.3-K.6); the histogram pass, its read-back and the host table
 *   build are skipped, so an encode never waits for the host.  RGB8 input only (the tables
 *   have codes for 8-bit precision symbols only).  No counterpart in the reference.
 */
typedef struct jpeg_hip_params {
    int32_t width;
    int32_t height;
    int32_t sampling[12];
    float qtable[3][64];
    int32_t restart_interval;
    int32_t pad_ones;
    int32_t mcu_row_begin;
    int32_t mcu_row_end;
    int32_t standard_huffman;
} jpeg_hip_params;

/* Derived sizes (Image::Image, Image.cpp:6-16; Encoder::createBlocks, Encoder.cpp:168-227). */
typedef struct jpeg_hip_layout {
    int32_t padded_width, padded_height;  /* "simulated" size */
    int32_t mcu_width, mcu_height;        /* SamplingDefinition::stepSize */
    int32_t mcus_per_row, mcu_rows;       /* whole image */
    int32_t band_mcu_rows;                /* rows encoded by this context */
    int32_t h_factor[3], v_factor[3];     /* SamplingDefinition::inverseFactor */
    int32_t blocks_per_mcu;
    int64_t blocks[3];                    /* 8x8 blocks per channel in the band */
    int64_t total_blocks;
    int64_t intervals;                    /* restart intervals in the band (1 without restarts) */
    int64_t max_jpeg_bytes;               /* safe output capacity for jpeg_hip_encode */
} jpeg_hip_layout;

/* ---- context ------------------------------------------------------------------------ */

/* Opens a context on a GPU.  Stands in for constructing the reference's Encoder
 * (Encoder.cpp:7-13); the device workspace grows on demand and is reused across images. */
int jpeg_hip_create(int device_id, jpeg_hip_ctx** out_ctx);
void jpeg_hip_destroy(jpeg_hip_ctx* ctx);
/* Text of the last error recorded on the context (never NULL). */
const char* jpeg_hip_last_error(const jpeg_hip_ctx* ctx);
const char* jpeg_hip_status_string(int status);
/* HIP stream the context launches on (a hipStream_t), for callers that time or order work. */
void* jpeg_hip_stream(jpeg_hip_ctx* ctx);

/* Ordering against the caller's own HIP streams.  The context works on a stream of its own
 * (jpeg_hip_stream) that does not synchronise with any other, the default stream included, so
 * a caller that fills a device buffer on its stream and hands the pointer over must say so:
 * jpeg_hip_wait_for_stream makes everything the context launches from now on wait for the work
 * queued on hip_stream (a hipStream_t, NULL for the default stream) up to this call;
 * jpeg_hip_stream_wait_for_ctx is the other direction, for results of entry points that return
 * before their kernels have finished (jpeg_hip_transform, jpeg_hip_histograms,
 * jpeg_hip_analyze_band, jpeg_hip_allreduce_histograms, jpeg_hip_finish_band_start).  Neither
 * blocks the host.  They stand in for the ownership hand-over of Encoder(ImagePtr, ...)
 * (Encoder.cpp:11-13), which on one CPU thread needs no ordering. */
int jpeg_hip_wait_for_stream(jpeg_hip_ctx* ctx, void* hip_stream);
int jpeg_hip_stream_wait_for_ctx(jpeg_hip_ctx* ctx, void* hip_stream);

/* Pure host helper: sizes for a parameter block (no GPU needed). */
int jpeg_hip_compute_layout(const jpeg_hip_params* params, jpeg_hip_layout* out_layout);

/* ---- one-shot encode ------------------------------------------------------------------ */

/* The whole of EncodeJPEG between "pixels in memory" and "file bytes in memory"
 * (JPEG-Encoder-SIMD.cpp:421-437 + Encoder::serialize, Encoder.cpp:439-488), input and
 * output resident in HBM.  d_rgb: interleaved RGB, row stride in bytes.  On success
 * *out_len receives the JFIF length. */
int jpeg_hip_encode(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                    const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity, int64_t* out_len);

/* jpeg_hip_encode for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, processes whose allocator is owned by someone else).  No
 * counterpart in the reference.  Same arguments, same bytes written to d_out, but the device
 * workspace (coefficients, histograms, tables, tile states, the unstuffed stream) is a fixed
 * array of jpeg_hip_static_workspace_bytes() bytes (32 MiB) that is part of the library and that
 * every GPU the library runs on holds until the process ends; the call makes no hipMalloc /
 * hipFree.  Like jpeg_hip_encode it waits for its kernels and reads the result back, so it is
 * not a call to make while a stream is being captured into a graph.  Limits: 128 bytes per
 * 8x8 block plus the entropy-coded stream of the image must fit the array (1920x1080 in any
 * sampling scheme does with per-image tables; with standard_huffman the array has to hold the
 * worst-case stream of 216 bytes per block instead, which limits it to about 95 000 blocks:
 * 1920x1080 in 4:2:0 fits, in 4:4:4 it does not); rows of byte pixels that do not start on 4-byte
 * boundaries take a slower kernel instead of a realigned copy.  Larger images fail with
 * JPEG_HIP_ERR_BUFFER_TOO_SMALL and a message.  All calls on one GPU, from any context or
 * thread, share that GPU's array: the library lets one of them in at a time (the others wait
 * inside the call).  The context has to be re-begun (jpeg_hip_begin or a one-call entry
 * point) before staged calls. */
int jpeg_hip_encode_static(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                           const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity, int64_t* out_len);
int64_t jpeg_hip_static_workspace_bytes(void);

/* Same with host buffers (copies in and out; PCIe time included). */
int jpeg_hip_encode_host(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes,
                         const jpeg_hip_params* params, uint8_t* h_out, int64_t out_capacity, int64_t* out_len);

/* The same for `count` equally shaped images in one call: the driver's loop over files
 * (JPEG-Encoder-SIMD.cpp:397-441 once per image) for frames that share size, sampling scheme
 * and quantization tables.  Image k starts at d_rgb + k * image_stride_bytes; its file --
 * byte-identical to what jpeg_hip_encode produces for that image, with its own optimized
 * Huffman tables -- is written to d_out + k * slot_bytes and is h_lens[k] bytes long.
 * params->restart_interval and the band rows must be 0.  Every kernel runs once over the
 * whole batch (large batches are cut into passes of about 1.5 GiB of coefficients).
 * JPEG_HIP_ERR_BUFFER_TOO_SMALL if a file does not fit its slot. */
int jpeg_hip_encode_batch(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                          int64_t image_stride_bytes, int count, const jpeg_hip_params* params, uint8_t* d_out,
                          int64_t slot_bytes, int64_t* h_lens);

/* The same with frames and files in host memory (PCIe time included): the frames travel in
 * passes through pinned staging, and the upload of one pass overlaps the encode of the
 * pass before it.  This is the ingest half of the reference's loop (ImageLoader pixels ->
 * Encoder, JPEG-Encoder-SIMD.cpp:404-437) for callers that hold decoded frames. */
int jpeg_hip_encode_batch_host(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes,
                               int64_t image_stride_bytes, int count, const jpeg_hip_params* params, uint8_t* h_out,
                               int64_t slot_bytes, int64_t* h_lens);

/* ---- staged entry points (what the host Encoder class drives) --------------------------- */

/* Image(w, h, scheme) + Encoder(image, qtables): fixes geometry and tables, sizes the
 * workspace (Image.cpp:6-16, Encoder.cpp:11-13). */
int jpeg_hip_begin(jpeg_hip_ctx* ctx, const jpeg_hip_params* params);

/* Image::setRawPixelData / ImageLoader::LoadPNG hand-over (Image.cpp:63-139,
 * ImageLoader.cpp:190-196): copies host pixels into the context's own device buffer. */
int jpeg_hip_upload_pixels(jpeg_hip_ctx* ctx, const void* h_rgb, int sample_format, int64_t stride_bytes);
/* The same hand-over for a caller that holds the reference's own Image: three planes of
 * floats in 0..1 (ImageData's r, g, b, Image.h:30-61, filled by Image::setRawPixelData /
 * transposeFloat, Image.cpp:63-139), rows row_stride_bytes apart (the reference's simulated
 * width times 4).  Only the image's own width x height samples are read -- the replicated
 * border the reference keeps around them is rebuilt by the kernels.  The planes are
 * interleaved on the device; the context then holds float pixels. */
int jpeg_hip_upload_planes(jpeg_hip_ctx* ctx, const float* h_r, const float* h_g, const float* h_b,
                           int64_t row_stride_bytes);
/* Zero-copy variant: uses caller-owned device pixels, which must stay valid until the
 * transform has run. */
int jpeg_hip_bind_pixels(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes);

/* Pixel filters on the context's RGB pixels, before the transform: Image::applySepia
 * (Image.cpp:212-219, matrix and operation order of applySepiaFilterAVXImpl,
 * SIMD256.h:218-244) and Encoder::multiplyColorChannelBy while the channels still are R, G, B
 * (Encoder.cpp:35-38, multiplyAVX, SIMD256.h:246-258).  The result is kept as float pixels
 * in the context (byte pixels are divided by 255.0f first, ImageLoader.cpp:190-193) and is
 * what the next jpeg_hip_transform reads.  channel: 0 R, 1 G, 2 B (ignored for sepia). */
enum { JPEG_HIP_FILTER_SEPIA = 0, JPEG_HIP_FILTER_MULTIPLY = 1 };
int jpeg_hip_filter_pixels(jpeg_hip_ctx* ctx, int filter, int channel, float value);
/* Encoder::multiplyColorChannelBy called after convertToYCbCr (Encoder.cpp:35-38, multiplyAVX,
 * SIMD256.h:246-258): the next jpeg_hip_transform multiplies every sample of component
 * `channel` (0 Y, 1 Cb, 2 Cr) by `value`, before the resolution is reduced
 * (after_subsampling = 0: the call came between convertToYCbCr and
 * reduceResolutionBySchema) or after it (1).  Up to JPEG_HIP_MAX_GAINS calls, applied in
 * call order; jpeg_hip_begin forgets them.  One-call entry points (encode, batch, bands)
 * do not take gains. */
enum { JPEG_HIP_MAX_GAINS = 4 };
int jpeg_hip_channel_gain(jpeg_hip_ctx* ctx, int channel, float value, int after_subsampling);
/* The context's current pixels as floats on the host (three per pixel, rows of the band),
 * e.g. to look at a filter's result the way ImageLoader::SavePPM would. */
int jpeg_hip_get_pixels(jpeg_hip_ctx* ctx, float* h_rgb, int64_t capacity_floats);

/* convertToYCbCr + reduceResolutionBySchema + applyDCT(Y|Cb|Cr) + zig-zag as one fused
 * kernel (Encoder.cpp:15-23, 157-166, 229-258; SIMD256.h:155-186, 260-304, 470-522;
 * ZigZag.h:10-49).  stages: JPEG_HIP_STAGE_* bits for the two calls a caller may have
 * skipped.  Leaves quantized int16 coefficients in HBM: blocks in scan (MCU interleaved)
 * order, 64 zig-zag ordered values per block. */
int jpeg_hip_transform(jpeg_hip_ctx* ctx, int stages);

/* calculateDCValues / calculateACValues symbol statistics (Encoder.cpp:260-352) reduced
 * to the four histograms createHuffmanTable counts (Encoder.cpp:368-416): DC-Y, DC-CbCr,
 * AC-Y, AC-CbCr, 256 bins each.  d_hist (uint64[4][256]) is overwritten. */
int jpeg_hip_histograms(jpeg_hip_ctx* ctx, uint64_t* d_hist);
/* Same, copied to the host. */
int jpeg_hip_histograms_host(jpeg_hip_ctx* ctx, uint64_t* h_hist);

/* HuffmanTable<byte>::create(16, symbols) (HuffmanCodingByte.h:102-210) from a histogram:
 * package-merge code lengths (<= 16) and right-growing codes.  Pure host function.
 * lengths[s] == 0 marks an unused symbol; codes are right aligned. */
int jpeg_hip_huffman_from_histogram(const uint64_t counts[256], uint8_t lengths[256], uint16_t codes[256]);

/* HuffmanTable<byte>::create(codeWordLength, symbols) (HuffmanCodingByte.h:102-210) for any limit
 * the reference accepts: 9 (its lower bound ceil(log2(256 + 1)), :121) to 64 (the width of its
 * code word, :105); anything else is JPEG_HIP_ERR_INVALID_ARGUMENT where the reference returns a
 * null table.  Codes are right aligned in 64 bits.  Pure host function; only tables of at most
 * 16 bits can go on to jpeg_hip_set_huffman / jpeg_hip_write_dht. */
int jpeg_hip_huffman_from_histogram_limit(const uint64_t counts[256], int code_word_length, uint8_t lengths[256],
                                          uint64_t codes[256]);

/* The typical Huffman tables of ITU-T T.81 Annex K.3 in the layout of jpeg_hip_get_huffman:
 * [0] DC luminance (table K.3), [1] DC chrominance (K.4), [2] AC luminance (K.5), [3] AC
 * chrominance (K.6), with the canonical codes of Annex C.  No counterpart in the reference
 * (which always optimises); see jpeg_hip_params.standard_huffman. */
int jpeg_hip_standard_huffman(uint8_t lengths[4][256], uint16_t codes[4][256]);

/* DefineHuffmannTable segment (JPEGSegments.h:102-140, 203-207) for a table.  Returns the
 * number of bytes written, or a negative status. */
int jpeg_hip_write_dht(const uint8_t lengths[256], const uint16_t codes[256], int table_id, int coefficient_type,
                       uint8_t* h_out, int capacity);

/* Everything Encoder::serialize emits before the scan (Encoder.cpp:473-483): SOI, APP0,
 * SOF0, DHT x4, DQT x3, [DRI], SOS.  Pure host function.  Returns bytes written or a
 * negative status. */
int jpeg_hip_write_headers(const jpeg_hip_params* params, const uint8_t lengths[4][256], const uint16_t codes[4][256],
                           uint8_t* h_out, int capacity);

/* Encoder::getHuffmanTable x4 (Encoder.cpp:418-437) for tables built elsewhere (e.g. from
 * histograms summed over several GPUs).  Order: DC-Y, DC-CbCr, AC-Y, AC-CbCr. */
int jpeg_hip_set_huffman(jpeg_hip_ctx* ctx, const uint8_t lengths[4][256], const uint16_t codes[4][256]);
/* Builds the four tables from host histograms and installs them (create x4 + set). */
int jpeg_hip_build_huffman(jpeg_hip_ctx* ctx, const uint64_t h_hist[4][256]);
/* Copies the installed tables out (Encoder::getHuffmanTable). */
int jpeg_hip_get_huffman(jpeg_hip_ctx* ctx, uint8_t lengths[4][256], uint16_t codes[4][256]);

/* Encoder::serializeScanData (Encoder.cpp:490-516) with BitBuffer::pushBitsEscaped
 * semantics (BitBuffer.cpp:128-204): writes the entropy coded segment of this band
 * (RSTn markers between its restart intervals, and in front of its first interval when
 * the band does not start the image) to d_out.  *out_len receives its length. */
int jpeg_hip_emit_scan(jpeg_hip_ctx* ctx, uint8_t* d_out, int64_t out_capacity, int64_t* out_len);

/* Same for a host buffer.  The scan is produced once into a device buffer owned by the
 * context and kept until the coefficients or tables change; *out_len always receives its
 * length.  With h_out == NULL nothing is copied (size query); a too small buffer returns
 * JPEG_HIP_ERR_BUFFER_TOO_SMALL. */
int jpeg_hip_emit_scan_host(jpeg_hip_ctx* ctx, uint8_t* h_out, int64_t out_capacity, int64_t* out_len);

/* Quantized coefficients of one channel copied to the host: blocks in the channel's MCU
 * order (Encoder::createBlocks), 64 zig-zag ordered int16 per block -- the values
 * Encoder::calculateDCValues/ACValues read (Encoder.cpp:268-269, 308-309). */
int jpeg_hip_get_coefficients(jpeg_hip_ctx* ctx, int channel, int16_t* h_out, int64_t capacity_values);
/* Device pointer of the coefficients in scan order (int16, total_blocks * 64: a block's 64 zig-zag
 * ordered values are contiguous).  For large images the library keeps the two halves of the
 * blocks in two planes: the pointer is then that of a joined copy made by this call (valid
 * until the context's next transform or call of this function); null when that copy fails. */
const int16_t* jpeg_hip_coefficients_device(jpeg_hip_ctx* ctx);

/* ---- sharded images (one band per GPU) ---------------------------------------------- */

/* First half of a sharded encode: transform + histograms of this band.  The caller sums
 * d_hist over all bands (one all-reduce) before calling jpeg_hip_finish_band. */
int jpeg_hip_analyze_band(jpeg_hip_ctx* ctx, const void* d_rgb, int sample_format, int64_t stride_bytes,
                          const jpeg_hip_params* params, uint64_t* d_hist);
/* Second half: tables from the summed histograms (host copy made internally), then this
 * band's piece of the file: headers if the band starts the image, the band's scan bytes,
 * EOI if the band ends the image.  Concatenating the pieces in band order gives the file.
 * d_hist_sum may be null when the band is the whole image: the histograms are then taken
 * as jpeg_hip_analyze_band counted them (the kernel leaves a copy with the host, so no
 * transfer is needed). */
int jpeg_hip_finish_band(jpeg_hip_ctx* ctx, const uint64_t* d_hist_sum, uint8_t* d_out, int64_t out_capacity,
                         int64_t* out_len);

/* jpeg_hip_finish_band in two halves, for a caller that keeps several images in flight on
 * several contexts (a band of a sharded image is a fraction of a millisecond of kernels;
 * a host that sits out every wait leaves the GPU idle most of the time): _start builds the
 * tables and launches the kernels that write the piece, _wait returns when they are done
 * and gives the length.  Between the two the context accepts no other call that starts
 * new work (jpeg_hip_begin and the one-call entry points return JPEG_HIP_ERR_BAD_STATE).
 * The reference, which encodes one image at a time on the host, has no counterpart. */
int jpeg_hip_finish_band_start(jpeg_hip_ctx* ctx, const uint64_t* d_hist_sum, uint8_t* d_out, int64_t out_capacity);
int jpeg_hip_finish_band_wait(jpeg_hip_ctx* ctx, int64_t* out_len);

/* The whole sharded encode of one rank in one call, for a host that holds an RCCL
 * communicator (an ncclComm_t, passed as void* so that this header needs no RCCL header):
 * jpeg_hip_analyze_band, one ncclAllReduce (sum, 4 x 256 uint64) of the histograms on the
 * context's stream, jpeg_hip_finish_band.  It stands in for Encoder::serialize
 * (Encoder.cpp:439-488) run on an image that several GPUs hold one band each of: every rank
 * derives the same tables from the summed histograms (Encoder::createHuffmanTable,
 * Encoder.cpp:368-416) and writes its piece of the file; the pieces concatenate in rank order.
 * params->mcu_row_begin / mcu_row_end name this rank's band, restart_interval must be set
 * (bands start on interval boundaries).  Collective: every rank of the communicator has to call
 * it, in the same order as the others.  RCCL is loaded when the first of these entry points
 * is called (librccl.so.1); without it they return JPEG_HIP_ERR_HIP and the rest of the library
 * works as before. */
int jpeg_hip_encode_sharded(jpeg_hip_ctx* ctx, void* rccl_comm, const void* d_rgb_band, int sample_format,
                            int64_t stride_bytes, const jpeg_hip_params* params, uint8_t* d_out, int64_t out_capacity,
                            int64_t* out_len);
/* Failure handling: every rank takes part in the all-reduce whatever happened to it before
 * (bad arguments, a refused workspace, a failed launch) -- its histograms then count as empty and a
 * flag word that travels with them says so -- and when any flag is up no rank writes a piece: the
 * failing rank returns its own status, every other rank JPEG_HIP_ERR_PEER_FAILED, and the
 * communicator stays usable.  (The standard tables go through the same exchange, for the flags
 * alone.)  Only a null context or communicator returns before the collective.  A failure
 * behind the exchange (an output buffer too small for the piece) is local: the other ranks'
 * pieces are complete, and a job that wants one verdict calls jpeg_hip_allreduce_status. */

/* One more collective for jobs that want every rank to know whether all pieces were written:
 * sums a 0/1 word (local_status != JPEG_HIP_OK) over the ranks, on the context's stream, and
 * waits for it.  *failed_ranks receives the number of ranks that reported a failure. */
int jpeg_hip_allreduce_status(jpeg_hip_ctx* ctx, void* rccl_comm, int local_status, int* failed_ranks);

/* The exchange step alone, for callers that keep several images in flight with
 * jpeg_hip_analyze_band / jpeg_hip_finish_band_start / _wait: sums d_hist (uint64[4][256], as
 * jpeg_hip_analyze_band filled it) over the ranks in place, on the context's stream. */
int jpeg_hip_allreduce_histograms(jpeg_hip_ctx* ctx, void* rccl_comm, uint64_t* d_hist);
/* Communicator set-up for hosts without RCCL bindings of their own (ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy): rank 0 makes the 128-byte id and hands it to the other
 * ranks by whatever channel the job has; every rank then creates its communicator on its GPU. */
int jpeg_hip_comm_unique_id(uint8_t id[128]);
int jpeg_hip_comm_create(const uint8_t id[128], int rank, int world_size, int device_id, void** out_comm);
void jpeg_hip_comm_destroy(void* comm);

/* Timing of the last jpeg_hip_encode / analyze / finish in milliseconds, measured with HIP
 * events on the context's stream: [0] fused transform kernel, [1] histogram kernel,
 * [2] entropy kernels (clears, bit writer, stuffing), [3] host table build.
 * Only filled while profiling is enabled. */
int jpeg_hip_set_profiling(jpeg_hip_ctx* ctx, int enabled);
int jpeg_hip_get_timings(jpeg_hip_ctx* ctx, float out_ms[4]);
/* Where the host was, and when, during the last jpeg_hip_analyze_band / jpeg_hip_finish_band
 * (or the one-call entry points built from them): microseconds of the steady clock at [0]
 * analyze entered, [1] transform and histogram kernels launched, [2] finish entered (behind the
 * caller's exchange step, if any), [3] histograms seen on the host, [4] tables and header built,
 * [5] entropy kernels launched, [6] wait for the result entered, [7] result seen.  The gaps
 * the kernel timings do not explain are here.  Only filled while profiling is enabled. */
int jpeg_hip_get_host_timeline(jpeg_hip_ctx* ctx, double out_us[8]);
/* The same figures added up over the single images encoded since jpeg_hip_set_profiling(ctx, 1)
 * was last called (which clears the sums): stage_ms as jpeg_hip_get_timings, host_us[0..6] the
 * seven intervals between the stations of jpeg_hip_get_host_timeline, host_us[7] the time from
 * one image's result to the next image's entry, *images the number of images, *entropy_runs
 * the number of times the entropy stage was launched for them (one per image unless a bit writer
 * had to be run again in another form).  For loops in which asking after every image would cost
 * more than the answer is worth. */
int jpeg_hip_get_profile_sums(jpeg_hip_ctx* ctx, double stage_ms[4], double host_us[8], int64_t* images,
                              int64_t* entropy_runs);

/* ---- verification decoder ---------------------------------------------------------------- */

/* The way back, for checking files on the box that made them: baseline Huffman decoding
 * (restart intervals included) on the host, then on the GPU dequantization, the inverse
 * transform of DCT::directIDCT (DCT.cpp:46-79), sample replication for subsampled chroma,
 * the matrix of convertYCbCrToRGBAVXImpl (SIMD256.h:188-216; Encoder::convertToRGB,
 * Encoder.cpp:25-33) and the byte rounding of ImageLoader::SavePPM (ImageLoader.cpp:168-172).
 * Accepts 8-bit, three-component, single-scan baseline files with integer sampling ratios --
 * everything this encoder writes.  A checker, not a fast path. */
typedef struct jpeg_hip_image_info {
    int32_t width, height;
    int32_t h_samp[3], v_samp[3];   /* sampling factors of the frame header */
    int32_t restart_interval;       /* MCUs, 0 = none */
    int64_t total_blocks;
} jpeg_hip_image_info;

/* Header parse only.  Pure host function. */
int jpeg_hip_decode_info(const uint8_t* h_jpeg, int64_t len, jpeg_hip_image_info* info);
/* File in host memory -> interleaved RGB8 in device memory (row stride in bytes). */
int jpeg_hip_decode(jpeg_hip_ctx* ctx, const uint8_t* h_jpeg, int64_t len, uint8_t* d_rgb, int64_t stride_bytes,
                    int64_t capacity_bytes, jpeg_hip_image_info* info);
/* Same, RGB8 copied to the host. */
int jpeg_hip_decode_host(jpeg_hip_ctx* ctx, const uint8_t* h_jpeg, int64_t len, uint8_t* h_rgb, int64_t stride_bytes,
                         int64_t capacity_bytes, jpeg_hip_image_info* info);
/* Quantized coefficients of the last decoded file: blocks in scan order, 64 zig-zag ordered
 * values each -- the layout of jpeg_hip_coefficients_device, so a file can be compared with
 * the coefficients it was written from. */
int jpeg_hip_decoded_coefficients(jpeg_hip_ctx* ctx, int16_t* h_out, int64_t capacity_values);
/* Mean squared error and PSNR (dB, 8-bit peak; +inf for identical images) between two
 * RGB8 images in device memory. */
int jpeg_hip_psnr(jpeg_hip_ctx* ctx, const uint8_t* d_a, int64_t stride_a, const uint8_t* d_b, int64_t stride_b,
                  int32_t width, int32_t height, double* mse, double* psnr);

/* ---- harness utility ------------------------------------------------------------------------ */

/* Fills device memory with the synthetic test images of the benchmark and the tests (no
 * counterpart in the reference, whose driver reads files): rows y0..y1 of `count` RGB8 images of
 * `width` pixels, image k (seed + k) starting at d_out + k * image_stride_bytes.  kind 0: uniform
 * noise, byte(x, y, c) = mix32(seed ^ (3 * (y * width + x) + c)) >> 24; kind 1: triangle-wave
 * colour ramps plus a 4-bit hashed dither ("natural-like").  Integer arithmetic only: the
 * same bytes as the host generators of the tests. */
int jpeg_hip_synthetic_images(jpeg_hip_ctx* ctx, int kind, int32_t width, int32_t y0, int32_t y1, uint32_t seed,
                              uint8_t* d_out, int64_t stride_bytes, int64_t image_stride_bytes, int count);

#ifdef __cplusplus
}
#endif
#endif /* JPEGENC_HIP_H */
