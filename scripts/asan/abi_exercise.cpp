// Exercises the C ABI's host code (batches with their worker pools and pipelines, uploads
// in pieces, the decoder, staged calls) under the host-side AddressSanitizer build.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jpegenc_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int st_ = (call);                                                              \
        if (st_ != 0) {                                                                      \
            fprintf(stderr, "%s -> %d (%s)\n", #call, st_, ctx ? jpeg_hip_last_error(ctx) : ""); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static jpeg_hip_params make_params(int w, int h, bool s420) {
    jpeg_hip_params p;
    memset(&p, 0, sizeof p);
    p.width = w;
    p.height = h;
    const int f = s420 ? 2 : 1;
    const int s[12] = {1, 0, 1, 0, f, 1, f, 1, f, 1, f, 1};   // {widthFactor, widthMethod, heightFactor, heightMethod} x 3, 1 = Average
    memcpy(p.sampling, s, sizeof s);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 64; ++i) p.qtable[c][i] = (float)(2 + (i + c) % 7);
    return p;
}

static void fill(std::vector<uint8_t>& v, unsigned seed) {
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < v.size(); ++i) {
        x = x * 1664525u + 1013904223u;
        v[i] = (uint8_t)((x >> 24) & ((i / 97) % 3 ? 0xff : 0x0f));   // noisy and quiet stretches
    }
}

int main() {
    jpeg_hip_ctx* ctx = nullptr;
    CHECK(jpeg_hip_create(0, &ctx));
    // 1. host batch: several passes, strides with padding, then every file decoded again
    {
        const int w = 642, h = 358, n = 70, rowPad = 5, imgPad = 13;
        const int64_t stride = w * 3 + rowPad, imageStride = stride * h + imgPad;
        std::vector<uint8_t> frames((size_t)(imageStride * n));
        fill(frames, 1);
        jpeg_hip_params p = make_params(w, h, true);
        const int64_t slot = 1 << 20;
        std::vector<uint8_t> out((size_t)(slot * n));
        std::vector<int64_t> lens(n);
        for (int rep = 0; rep < 3; ++rep)
            CHECK(jpeg_hip_encode_batch_host(ctx, frames.data(), JPEG_HIP_RGB8, stride, imageStride, n, &p, out.data(), slot,
                                             lens.data()));
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (int k = 0; k < n; k += 9) {
            jpeg_hip_image_info info;
            CHECK(jpeg_hip_decode_host(ctx, out.data() + k * slot, lens[k], rgb.data(), w * 3, (int64_t)rgb.size(), &info));
            if (info.width != w || info.height != h) return 2;
        }
        p.standard_huffman = 1;
        CHECK(jpeg_hip_encode_batch_host(ctx, frames.data(), JPEG_HIP_RGB8, stride, imageStride, n, &p, out.data(), slot,
                                         lens.data()));
    }
    // 2. single images from host memory: odd sizes, restart intervals, 4:4:4, coefficients
    for (int k = 0; k < 6; ++k) {
        const int w = 31 + 173 * k, h = 17 + 97 * k;
        std::vector<uint8_t> img((size_t)w * h * 3);
        fill(img, 10 + k);
        jpeg_hip_params p = make_params(w, h, k % 2 == 0);
        p.restart_interval = k % 3 == 0 ? 0 : 3 + k;
        p.pad_ones = k & 1;
        jpeg_hip_layout lay;
        CHECK(jpeg_hip_compute_layout(&p, &lay));
        std::vector<uint8_t> file((size_t)lay.max_jpeg_bytes);
        int64_t len = 0;
        CHECK(jpeg_hip_encode_host(ctx, img.data(), JPEG_HIP_RGB8, w * 3, &p, file.data(), (int64_t)file.size(), &len));
        for (int c = 0; c < 3; ++c) {
            std::vector<int16_t> coef((size_t)lay.blocks[c] * 64);
            CHECK(jpeg_hip_get_coefficients(ctx, c, coef.data(), (int64_t)coef.size()));
        }
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        jpeg_hip_image_info info;
        CHECK(jpeg_hip_decode_host(ctx, file.data(), len, rgb.data(), w * 3, (int64_t)rgb.size(), &info));
    }
    // 3. staged calls with gains and filters, device pixels through the band entry points
    {
        const int w = 500, h = 300;
        std::vector<uint8_t> img((size_t)w * h * 3);
        fill(img, 99);
        jpeg_hip_params p = make_params(w, h, true);
        CHECK(jpeg_hip_begin(ctx, &p));
        CHECK(jpeg_hip_upload_pixels(ctx, img.data(), JPEG_HIP_RGB8, w * 3));
        CHECK(jpeg_hip_filter_pixels(ctx, JPEG_HIP_FILTER_SEPIA, 0, 0.0f));
        CHECK(jpeg_hip_channel_gain(ctx, 1, 0.5f, 0));
        CHECK(jpeg_hip_channel_gain(ctx, 0, 1.25f, 1));
        CHECK(jpeg_hip_transform(ctx, JPEG_HIP_STAGE_ALL));
        uint64_t hist[4][256];
        CHECK(jpeg_hip_histograms_host(ctx, &hist[0][0]));
        CHECK(jpeg_hip_build_huffman(ctx, hist));
        int64_t len = 0;
        CHECK(jpeg_hip_emit_scan_host(ctx, nullptr, 0, &len));
        std::vector<uint8_t> scan((size_t)len);
        CHECK(jpeg_hip_emit_scan_host(ctx, scan.data(), len, &len));

        uint8_t *dRgb = nullptr, *dOut = nullptr;
        if (hipMalloc(&dRgb, img.size()) != hipSuccess || hipMalloc(&dOut, 1 << 22) != hipSuccess) return 3;
        if (hipMemcpy(dRgb, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) return 3;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(jpeg_hip_analyze_band(ctx, dRgb, JPEG_HIP_RGB8, w * 3, &p, nullptr));
            CHECK(jpeg_hip_finish_band(ctx, nullptr, dOut, 1 << 22, &len));
            CHECK(jpeg_hip_encode(ctx, dRgb, JPEG_HIP_RGB8, w * 3, &p, dOut, 1 << 22, &len));
        }
        // the two halves of finish_band, two contexts taking turns; a refused call in between
        {
            jpeg_hip_ctx* other = nullptr;
            CHECK(jpeg_hip_create(0, &other));
            uint8_t* dOut2 = nullptr;
            if (hipMalloc(&dOut2, 1 << 22) != hipSuccess) return 3;
            int64_t la = 0, lb = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(jpeg_hip_analyze_band(ctx, dRgb, JPEG_HIP_RGB8, w * 3, &p, nullptr));
                CHECK(jpeg_hip_analyze_band(other, dRgb, JPEG_HIP_RGB8, w * 3, &p, nullptr));
                CHECK(jpeg_hip_finish_band_start(ctx, nullptr, dOut, 1 << 22));
                CHECK(jpeg_hip_finish_band_start(other, nullptr, dOut2, 1 << 22));
                if (jpeg_hip_begin(ctx, &p) != JPEG_HIP_ERR_BAD_STATE) return 4;
                CHECK(jpeg_hip_finish_band_wait(ctx, &la));
                CHECK(jpeg_hip_finish_band_wait(other, &lb));
                if (la != lb || la != len) return 4;
            }
            (void)hipFree(dOut2);
            jpeg_hip_destroy(other);
        }
        // three float planes with padded rows (the reference Image's layout), odd width
        {
            const int pw = 333, ph = 77, sw = 336, sh = 80;
            std::vector<float> planes((size_t)3 * sw * sh);
            for (size_t i = 0; i < planes.size(); ++i) planes[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65535.0f;
            jpeg_hip_params pp = make_params(pw, ph, true);
            CHECK(jpeg_hip_begin(ctx, &pp));
            CHECK(jpeg_hip_upload_planes(ctx, planes.data(), planes.data() + (size_t)sw * sh, planes.data() + (size_t)2 * sw * sh,
                                         (int64_t)sw * sizeof(float)));
            CHECK(jpeg_hip_transform(ctx, JPEG_HIP_STAGE_ALL));
            CHECK(jpeg_hip_histograms_host(ctx, &hist[0][0]));
            CHECK(jpeg_hip_build_huffman(ctx, hist));
            int64_t n2 = 0;
            CHECK(jpeg_hip_emit_scan_host(ctx, nullptr, 0, &n2));
        }
        // device batch, small
        const int n = 40;
        uint8_t* dFrames = nullptr;
        uint8_t* dSlots = nullptr;
        if (hipMalloc(&dFrames, img.size() * n) != hipSuccess || hipMalloc(&dSlots, (size_t)n << 20) != hipSuccess) return 3;
        for (int k = 0; k < n; ++k)
            if (hipMemcpy(dFrames + img.size() * k, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) return 3;
        std::vector<int64_t> lens(n);
        for (int rep = 0; rep < 3; ++rep)
            CHECK(jpeg_hip_encode_batch(ctx, dFrames, JPEG_HIP_RGB8, w * 3, (int64_t)img.size(), n, &p, dSlots, 1 << 20, lens.data()));
        (void)hipFree(dRgb); (void)hipFree(dOut); (void)hipFree(dFrames); (void)hipFree(dSlots);
    }
    jpeg_hip_destroy(ctx);
    printf("abi exercise ok\n");
    return 0;
}
