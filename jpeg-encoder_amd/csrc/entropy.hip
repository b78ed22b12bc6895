// Entropy stage of the MI355X (gfx950) JPEG encode hot path: symbol histograms, the
// single-pass bit writer and byte stuffing.
#include <cstdlib>

#include "device_common.h"
#include "kernels.h"

#include "jpegenc_hip.h"

namespace jpeghip {

// ----------------------------------------------------------------------------------------
// Entropy stage, shared block walk.  A visitor sees the DC symbol, every AC symbol with
// its amplitude bits, zero-run-length symbols and the end-of-block symbol, in scan order
// (Encoder::calculateDCValues / calculateACValues, Encoder.cpp:260-352).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int bit_category(int v) {   // Encoder::lookupBitCategory, Encoder.cpp:354-366
    const int a = v < 0 ? -v : v;
    return 32 - __clz(a);   // __clz(0) == 32
}

// Number of leading bits equal to the sign bit (v_ffbh_i32; the compiler has no builtin for
// it and expands __builtin_clrsb into several instructions).  Not for 0 and -1.
__device__ __forceinline__ int leading_sign_bits(int x) {
    int r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ int coef_at(const u32 (&w)[32], int i) {
    return (int)(short)(w[i >> 1] >> ((i & 1) * 16));
}

// Where a block sits in its image.  The block index inside one image fits 32 bits for every
// legal JPEG (65535 x 65535 pixels are 2 * 10^8 blocks), which keeps the divisions cheap.
struct BlockRole {
    u32 mcu;       // MCU inside the image
    int k;         // block inside the MCU
    int c;         // channel
    int back;      // blocks between this one and the previous block of its channel, 1..bpm
    bool resets;   // first block of its channel in a restart interval: the DC predictor is 0
};

__device__ __forceinline__ BlockRole block_role(const KGeom& g, u32 bl) {
    BlockRole r;
    const u32 bpm = (u32)g.bpm;
    r.mcu = bl / bpm;
    r.k = (int)(bl - r.mcu * bpm);
    r.c = r.k < g.off1 ? 0 : (r.k < g.off2 ? 1 : 2);
    const int first = r.c == 0 ? 0 : (r.c == 1 ? g.off1 : g.off2);
    const int last = (r.c == 0 ? g.off1 : (r.c == 1 ? g.off2 : g.bpm)) - 1;
    r.back = r.k > first ? 1 : g.bpm - (last - first);
    // One interval per image (no restarts, or a batch): only MCU 0 starts one.
    const bool atStart = g.intervals == g.batch ? r.mcu == 0 : r.mcu % (u32)g.mcusPerInterval == 0;
    r.resets = r.k == first && atStart;
    return r;
}

// Block whose DC term the lane has to fetch from memory as predictor of block b: lanes hold
// consecutive blocks (lane l: block b0 + l), the predecessor is at most one MCU back, so all
// but the first `back` lanes of a wave get it from a neighbouring lane.  Lanes that need no
// fetch name their own block, which keeps the address valid and the line in cache.
__device__ __forceinline__ long long dc_predictor_source(const BlockRole& r, long long b) {
    const int lane = threadIdx.x & 63;
    return (lane >= r.back || r.resets) ? b : b - r.back;
}

// DC predictor (previous block of the same channel in scan order, 0 at the start of a
// restart interval and of the band) from the lane's own DC term, its neighbours' and the
// term fetched from dc_predictor_source().  Call with all lanes of the wave.
__device__ __forceinline__ int dc_predictor_lanes(const BlockRole& r, int myDc, int fetchedDc, bool active) {
    const int lane = threadIdx.x & 63;
    const int fromLane = __shfl(myDc, (lane - r.back) & 63, 64);
    if (!active || r.resets) return 0;
    return lane >= r.back ? fromLane : fetchedDc;
}

template <class Visitor>
__device__ __forceinline__ void walk_block(const u32 (&w)[32], int pred, Visitor& vis) {
    const int diff = coef_at(w, 0) - pred;
    const int dcat = bit_category(diff);
    vis.dc(dcat, (u32)(diff < 0 ? diff - 1 : diff) & ((1u << dcat) - 1u));
    int last = 0;   // index of the last non-zero coefficient seen (the DC term counts)
    // Eight coefficients (one 16-byte load) at a time: groups that are all zero -- most of
    // the high-frequency half of a typical block -- are skipped.  A zero coefficient costs
    // nothing but the test: the run in front of a non-zero one is the distance to `last`.
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const u32 any = (grp == 0 ? (w[0] & 0xffff0000u) : w[4 * grp]) | w[4 * grp + 1] | w[4 * grp + 2] | w[4 * grp + 3];
        if (any == 0) continue;
#pragma unroll
        for (int i = 8 * grp + (grp == 0 ? 1 : 0); i < 8 * grp + 8; ++i) {
            const int v = coef_at(w, i);
            if (v != 0) {
                int run = i - 1 - last;
                last = i;
                if (run > 15) {
                    vis.zrl(run >> 4);
                    run &= 15;
                }
                // The amplitude bits are those of v, or of v - 1 (the one's complement of |v|)
                // for a negative v; either way the category is the number of bits left
                // after the redundant sign bits (never 0 or -1 here: v is not 0).
                const int amp = v + (v >> 31);
                const int cat = 32 - leading_sign_bits(amp);
                vis.ac((run << 4) | cat, cat, (u32)amp & ((1u << cat) - 1u));
            }
        }
    }
    if (last != 63) vis.eob();
}

// A block's 64 coefficients: groups 0..3 and groups 4..7 from where the geometry says they lie
// (KGeom::coefPitch, KGeom::coefHigh).
__device__ __forceinline__ void load_block(const KGeom& g, const int16_t* coef, long long b, u32 (&w)[32]) {
    const uint4* low = (const uint4*)((const char*)coef + b * g.coefPitch);
    const uint4* high = (const uint4*)((const char*)low + g.coefHigh);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 t = i < 4 ? low[i] : high[i - 4];
        w[4 * i] = t.x;
        w[4 * i + 1] = t.y;
        w[4 * i + 2] = t.z;
        w[4 * i + 3] = t.w;
    }
}

// DC term of block b.
__device__ __forceinline__ int load_dc(const KGeom& g, const int16_t* coef, long long b) {
    return *(const int16_t*)((const char*)coef + b * g.coefPitch);
}

constexpr int kEntThreads = 256;
constexpr u32 kNoCodeLength = 0x4000u;   // stands in for the length of a symbol without a code (a block has at most 64 * 27 bits)

// Diagnostic switch: JPEGENC_OLD_ENTROPY=1 selects the one-lane-per-block kernels throughout.
static bool old_entropy_kernels() {
    static const bool old = getenv("JPEGENC_OLD_ENTROPY") != nullptr && getenv("JPEGENC_OLD_ENTROPY")[0] == '1';
    return old;
}

// ---- histograms -------------------------------------------------------------------------
struct HistVisitor {
    u32* dcHist;
    u32* acHist;
    __device__ __forceinline__ void dc(int cat, u32) { atomicAdd(&dcHist[cat], 1u); }
    __device__ __forceinline__ void ac(int sym, int, u32) { atomicAdd(&acHist[sym], 1u); }
    __device__ __forceinline__ void zrl(int n) { atomicAdd(&acHist[0xF0], (u32)n); }
    __device__ __forceinline__ void eob() { atomicAdd(&acHist[0], 1u); }
};

// Adds a workgroup's 1024 counters in LDS (complete: the caller has passed a barrier) to the
// image's histograms in memory, and hands the sums to the host when the export asks for it.
// tileStats (may be null): the workgroup's three maxima for HistogramExport::stats.
__device__ __forceinline__ void flush_histograms(u32* lh, u64* __restrict__ hist, const HistogramExport& ex,
                                                 const u32* tileStats = nullptr) {
    __shared__ bool lastOne;
    u64* mine = hist + (long long)blockIdx.y * 1024;
    if (ex.host == nullptr) {
        for (int i = threadIdx.x; i < 1024; i += kEntThreads)
            if (lh[i] != 0) atomicAdd(&mine[i], (u64)lh[i]);
        return;
    }
    // The workgroup that finishes last hands the complete histograms to the host.  No
    // fences (a release at device scope writes the whole L2 back, once per workgroup): the
    // additions return their old values, so they have been performed when the workgroup
    // counts itself off, and the last one reads the sums where the additions were made.
    // The counter is never reset: the launcher tells which value the last one will see.
    u64 old[1024 / kEntThreads];
#pragma unroll
    for (int k = 0; k < 1024 / kEntThreads; ++k) {   // all in flight before the first is waited for
        const u32 v = lh[k * kEntThreads + threadIdx.x];
        old[k] = v != 0 ? atomicAdd(&mine[k * kEntThreads + threadIdx.x], (u64)v) : 0;
    }
    u64 seen = 0;
    // Tile statistics: one word per workgroup, put down with an exchange that returns like the additions
    // above (three words for all to raise, one behind the other, cost a 22 us kernel of 285 workgroups 4 us).
    u64 before = 0;   // (what the word held is of no interest; asking for it makes the exchange one that is waited for below)
    if (tileStats != nullptr && ex.stats != nullptr && threadIdx.x == 0)
        before = atomicExch((unsigned long long*)&ex.stats[blockIdx.x],
                            ((u64)tileStats[0] << 40) | ((u64)tileStats[1] << 20) | (u64)tileStats[2]);
#pragma unroll
    for (int k = 0; k < 1024 / kEntThreads; ++k) seen |= old[k];
    // The returned values have to be here before the workgroup counts itself off: an explicit wait
    // for the vector memory counter (the additions are in flight on it), and `seen`, on which a
    // store depends (under a condition that never holds), so that the compiler cannot move the
    // additions behind the barrier either.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    asm volatile("" : : "v"((u32)before), "v"((u32)(before >> 32)) : "memory");
    if (seen == ~0ull) lh[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) lastOne = atomicAdd(ex.counter, 1ull) + 1 == ex.expected;
    __syncthreads();
    if (!lastOne) return;
    for (int i = threadIdx.x; i < 1024; i += kEntThreads)
        ex.host[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ex.hostStats != nullptr) {
        // [3]: whether the kernel keeps these statistics at all
        if (tileStats == nullptr || ex.stats == nullptr) {
            if (threadIdx.x < 4) ex.hostStats[threadIdx.x] = 0;
            return;
        }
        if (threadIdx.x >= 64) return;   // one wave gathers the workgroups' words
        u32 a = 0, b = 0, c = 0;
        for (u32 i = threadIdx.x; i < gridDim.x; i += 64) {
            const u64 v = __hip_atomic_load(&ex.stats[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a = max(a, (u32)(v >> 40));
            b = max(b, (u32)(v >> 20) & 0xfffffu);
            c = max(c, (u32)v & 0xfffffu);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            a = max(a, (u32)__shfl_xor((int)a, d, 64));
            b = max(b, (u32)__shfl_xor((int)b, d, 64));
            c = max(c, (u32)__shfl_xor((int)c, d, 64));
        }
        if (threadIdx.x < 4) ex.hostStats[threadIdx.x] = threadIdx.x == 0 ? a : (threadIdx.x == 1 ? b : (threadIdx.x == 2 ? c : 1ull));
    }
}

// blockIdx.y is the image of the batch; a workgroup never mixes images, so its LDS counts
// belong to one histogram set.
__global__ __launch_bounds__(kEntThreads) void k_histogram(KGeom g, const int16_t* __restrict__ coef,
                                                           u64* __restrict__ hist, HistogramExport ex) {
    __shared__ u32 lh[4 * 256];   // DC-Y, DC-C, AC-Y, AC-C (Encoder.cpp:368-416)
    for (int i = threadIdx.x; i < 1024; i += kEntThreads) lh[i] = 0;
    __syncthreads();
    const long long image = (long long)blockIdx.y * g.blocksPerImage;   // first block of this image
    const u32 blocks = (u32)g.blocksPerImage;
    const u32 stride = gridDim.x * kEntThreads;
    for (u32 bl = blockIdx.x * kEntThreads + threadIdx.x; __any(bl < blocks); bl += stride) {
        const bool active = bl < blocks;
        const u32 cur = active ? bl : blocks - 1;   // lanes past the end stay in step for the shuffle
        const BlockRole role = block_role(g, cur);
        u32 w[32];
        load_block(g, coef, image + cur, w);
        const int fetched = load_dc(g, coef, image + dc_predictor_source(role, cur));
        const int pred = dc_predictor_lanes(role, coef_at(w, 0), fetched, active);
        if (active) {
            const int t = role.c == 0 ? 0 : 1;
            HistVisitor vis{lh + 256 * t, lh + 512 + 256 * t};
            walk_block(w, pred, vis);
        }
    }
    __syncthreads();
    flush_histograms(lh, hist, ex);
}

// Zeroes n 8-byte words.  A kernel rather than hipMemsetAsync: a launch costs the host a
// few microseconds, a memset call several times that.
__global__ __launch_bounds__(256) void k_zero_words(u64* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0;
}

hipError_t launch_clear_histograms(unsigned long long* d_hist, int images, hipStream_t s) {
    const long long words = 1024LL * images;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((words + 255) / 256 > 1024 ? 1024 : (words + 255) / 256)), dim3(256), 0,
                       s, d_hist, words);
    return hipGetLastError();
}

static long long histogram_tiles_min_blocks();
static hipError_t launch_histogram_tiles(const KGeom& g, const int16_t* d_coef, u64* d_hist, hipStream_t s, HistogramExport* ex);

hipError_t launch_histogram(const KGeom& g, const int16_t* d_coef, u64* d_hist, hipStream_t s, bool clear,
                            HistogramExport* ex) {
    if (clear) {
        const hipError_t e = launch_clear_histograms(d_hist, g.batch, s);
        if (e != hipSuccess) return e;
    }
    // Every workgroup takes the same number of 256-block passes.
    const long long passes = (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    // A workgroup ends with a flush of up to 1024 counters: a mid-sized image is better off
    // with two passes per workgroup than with twice the workgroups (4K 4:4:4: 40 -> 26 us),
    // a small one needs every workgroup it can get.
    long long cap = g.batch > 1 ? (4096 / g.batch > 8 ? 4096 / g.batch : 8) : 2048;
    if (g.batch == 1 && passes / 2 < cap) cap = passes / 2 > 256 ? passes / 2 : 256;
    const long long rounds = (passes + cap - 1) / cap;
    long long grid = (passes + rounds - 1) / rounds;
    if (grid < 1) grid = 1;
    HistogramExport none = {nullptr, 0, nullptr, nullptr, nullptr};
    // Large images: one lane per group of eight coefficients that is not empty (k_histogram_tiles).
    // (Diagnostic switch JPEGENC_HIST_FORM=1: always one lane per block.)
    static const bool perBlock = getenv("JPEGENC_HIST_FORM") != nullptr && getenv("JPEGENC_HIST_FORM")[0] == '1';
    // A batch counts by its whole size, as long as an image has tiles for every wave of a few workgroups
    // (64 x 1080p 4:2:0, 3.1 million blocks: 111 against 120 us).
    const long long counted = g.batch > 1 && g.blocksPerImage >= 16384 ? g.blocksPerImage * g.batch : g.blocksPerImage;
    if (!perBlock && g.groupMask != nullptr && counted >= histogram_tiles_min_blocks()) {
        const hipError_t e = launch_histogram_tiles(g, d_coef, d_hist, s, ex != nullptr && g.batch == 1 ? ex : nullptr);
        return e;
    }
    if (ex != nullptr && g.batch == 1) ex->expected += (unsigned long long)grid;   // what the counter reads when all are done
    hipLaunchKernelGGL(k_histogram, dim3((unsigned)grid, (unsigned)g.batch), dim3(kEntThreads), 0, s, g, d_coef, d_hist,
                       ex != nullptr && g.batch == 1 ? *ex : none);
    return hipGetLastError();
}

// ---- bit lengths -------------------------------------------------------------------------
// Bits a block occupies in the stream under the given tables.
struct LengthVisitor {
    const u32* dcTab;
    const u32* acTab;
    u32 bits;
    __device__ __forceinline__ void dc(int cat, u32) { bits += (dcTab[cat] >> 16) + cat; }
    __device__ __forceinline__ void ac(int sym, int cat, u32) { bits += (acTab[sym] >> 16) + cat; }
    __device__ __forceinline__ void zrl(int n) { bits += n * (acTab[0xF0] >> 16); }
    __device__ __forceinline__ void eob() { bits += acTab[0] >> 16; }
};

// ---- workgroup scan ----------------------------------------------------------------------
constexpr int kScanThreads = 256;

__device__ __forceinline__ u64 wave_inclusive_scan(u64 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread workgroup; returns the
// exclusive prefix and the workgroup total.  lds: 4 values.
__device__ __forceinline__ u64 block_exclusive_scan(u64 v, u64* lds, u64& total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 inc = wave_inclusive_scan(v);
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    u64 base = 0;
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < kScanThreads / 64; ++i) {
        if (i < wave) base += lds[i];
        sum += lds[i];
    }
    total = sum;
    return base + inc - v;
}

// A barrier for a caller with loads in flight that it must not wait for: it orders accesses to LDS only.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the LDS accesses, not the loads from memory
    __builtin_amdgcn_s_barrier();
}

// Per 4 KiB chunk of the unstuffed stream, ffCount holds the number of stuffing bytes the chunk
// needs (its 0xFF bytes that are data) in the low half and the number of markers that begin in it
// in the high half.  Either the counting pass writes the word, or the kernels that place the bits
// add to it (a marker's 0xFF is counted like any other byte there and taken off again where the
// marker is known: kChunkMarkerDelta does both; the sums are exact once all have been added).
constexpr u32 kChunkCountMask = 0xffffu;
constexpr u32 kChunkMarkerOne = 0x10000u;
constexpr u32 kChunkMarkerDelta = kChunkMarkerOne - 1u;
constexpr int kChunkWordShift = 10;   // stream words per chunk: 1024
static_assert(kStuffChunk == 4 << kChunkWordShift, "a chunk is 1024 words");

__device__ __forceinline__ u32 ff_bits(u32 w) {   // bit k: byte k of w is 0xFF
    u32 t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    // The flags sit at bits 0, 8, 16, 24; the product brings them to bits 24..27 (no two of
    // the sixteen partial products meet, so nothing carries).
    return ((t & 0x01010101u) * 0x01020408u) >> 24;
}

// ORs v into a word of the (zero-initialised) stream that other writers may OR into as well, and
// keeps the chunk's count of 0xFF bytes right: every writer adds what its own OR changed (the
// bytes that are 0xFF behind it minus those that were before), which adds up to the count of the
// finished word whatever the order.
__device__ __forceinline__ void or_counted(u32* __restrict__ words, u64 index, u32 v, u32* __restrict__ ffCount) {
    const u32 old = atomicOr(&words[index], v);
    const u32 n = (u32)__popc(ff_bits(old | v)) - (u32)__popc(ff_bits(old));
    if (n != 0) atomicAdd(&ffCount[index >> kChunkWordShift], n);
}

// A workgroup's staged words (native order, word 0 = stream word wordBase) go to the stream byte
// swapped: every wave takes a contiguous quarter, 64 consecutive words per instruction.  The first
// and the last word may be shared with the neighbouring tiles and are merged.  ffCount (may be
// null): the 0xFF bytes of the words are added to their chunks' counts on the way (see
// kChunkCountMask), one addition per wave and chunk.
__device__ __forceinline__ void copy_out_counted(const u32* stage, u32 nWords, u64 wordBase, u32* __restrict__ words,
                                                 u32* __restrict__ ffCount) {
    const int lane = threadIdx.x & 63;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 quarter = (nWords + 3u) / 4u;
    const u32 begin = wave * quarter;
    const u32 end = begin + quarter < nWords ? begin + quarter : nWords;
    if (begin >= end) return;
    u64 curChunk = (wordBase + begin) >> kChunkWordShift;
    u32 acc = 0;
    auto flush = [&](u32 mine) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += (u32)__shfl_xor((int)mine, d, 64);
        if (lane == 0 && mine != 0) atomicAdd(&ffCount[curChunk], mine);
    };
    for (u32 base = begin; base < end; base += 64u) {
        const u32 i = base + (u32)lane;
        u32 n = 0;
        bool inCur = true;
        if (i < end) {
            const u32 v = __builtin_bswap32(stage[i]);
            const u64 at = wordBase + i;
            if (i == 0 || i == nWords - 1u) {
                if (v != 0) {
                    if (ffCount != nullptr) {
                        const u32 old = atomicOr(&words[at], v);
                        n = (u32)__popc(ff_bits(old | v)) - (u32)__popc(ff_bits(old));
                    } else {
                        atomicOr(&words[at], v);
                    }
                }
            } else {
                words[at] = v;
                n = (u32)__popc(ff_bits(v));
            }
            inCur = (at >> kChunkWordShift) == curChunk;
        }
        if (ffCount == nullptr) continue;
        const u32 last = base + 63u < end ? base + 63u : end - 1u;
        const u64 lastChunk = (wordBase + last) >> kChunkWordShift;   // the same for the whole wave
        if (lastChunk == curChunk) {
            acc += n;
        } else {   // a chunk ends inside these 64 words (at most one does)
            flush(acc + (inCur ? n : 0u));
            acc = inCur ? 0u : n;
            curChunk = lastChunk;
        }
    }
    if (ffCount != nullptr) flush(acc);
}

// ---- restart intervals -------------------------------------------------------------------
// The unstuffed stream holds, per interval, an RSTn marker (every interval but the image's
// first) and the interval's bit string padded to a whole byte; the band that ends the
// image appends EOI.  Markers travel inside the stream so that the stuffing pass needs no
// interval bookkeeping; it recognises them by position (is_marker_ff).
__device__ __forceinline__ int interval_marker_bytes(const KGeom& g, long long k) {
    return g.batch == 1 && g.firstInterval + k > 0 ? 2 : 0;
}

// Does EOI follow interval k?  The last interval of a band that ends its image; in a batch
// every interval is a whole image.
__device__ __forceinline__ bool interval_has_eoi(const KGeom& g, long long k) {
    return g.writeEoi && (g.batch > 1 || k == g.intervals - 1);
}

// ---- bit writer ------------------------------------------------------------------------
// One thread per block appends its codes MSB first.  A workgroup's 256 blocks cover one
// contiguous bit range; it is assembled in LDS (native word order, ds_or for the words two
// blocks share) and copied out byte swapped with coalesced stores, so that memory holds the
// big-endian bit string BitBuffer::pushBits builds (BitBuffer.cpp:128-204).  Only the
// range's first and last word are merged into the zeroed buffer with atomicOr.  Ranges too
// long for the LDS stage are written straight to memory.
constexpr int kEmitStageShort = 2048;   // stage words of a tile of short blocks (8 bytes per block on average)

// Where a lane's bits go: the LDS stage (native order) or the global stream (swapped).
struct BitTarget {
    u32* words;
    long long bias;   // index of the stream word that words[0] stands for
    bool swap;
    u32* ff;          // per-chunk 0xFF counts to keep up to date when the target is the stream itself, or null
    // Stream words that do not go to memory but to two words in LDS (a tile's first and last word,
    // which k_merge_edges writes): edge is null when there are none.
    u32* edge;
    long long edgeLo, edgeHi;
    __device__ __forceinline__ bool diverted(long long index, u32 v) const {
        if (edge == nullptr || (index != edgeLo && index != edgeHi)) return false;
        atomicOr(&edge[index == edgeLo ? 0 : 1], __builtin_bswap32(v));
        return true;
    }
    __device__ __forceinline__ void or_word(long long index, u32 v) const {
        if (v == 0) return;
        if (diverted(index, v)) return;
        if (swap && ff != nullptr) or_counted(words, (u64)index, __builtin_bswap32(v), ff);
        else atomicOr(&words[index - bias], swap ? __builtin_bswap32(v) : v);
    }
    // A whole word of the stream that belongs to the caller alone.
    __device__ __forceinline__ void store_word(long long index, u32 v) const {
        if (diverted(index, v)) return;
        words[index - bias] = swap ? __builtin_bswap32(v) : v;
        if (swap && ff != nullptr) {
            const u32 n = (u32)__popc(ff_bits(v));
            if (n != 0) atomicAdd(&ff[(u64)index >> kChunkWordShift], n);
        }
    }
    // ORs up to 32 bits, given left aligned in v, into the stream at bit position pos.
    __device__ __forceinline__ void or32(u64 pos, u32 v) const {
        const long long d = (long long)(pos >> 5);
        const int sh = (int)(pos & 31);
        or_word(d, v >> sh);
        if (sh != 0) or_word(d + 1, v << (32 - sh));
    }
};

// First walk: the block's bit string, starting at bit 0, into a few words of LDS private to
// the lane, and its length.  A block longer than the private words is only measured.
constexpr int kPrivWords = 8;
constexpr int kPrivPitch = 9;   // odd, so that the lanes' words spread over the banks

struct PrivateVisitor {
    const u32* dcTab;
    const u32* acTab;
    u32* words;
    u64 acc;    // pending bits, right aligned
    int n;      // number of pending bits (< 32 between calls)
    int word;   // words completed
    __device__ __forceinline__ void put(u32 bits, int len) {
        acc = (acc << len) | bits;
        n += len;
        if (n >= 32) {
            if (word < kPrivWords) words[word] = (u32)(acc >> (n - 32));
            ++word;
            n -= 32;
        }
    }
    __device__ __forceinline__ void dc(int cat, u32 amp) {
        const u32 e = dcTab[cat];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void ac(int sym, int cat, u32 amp) {
        const u32 e = acTab[sym];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void zrl(int cnt) {
        const u32 e = acTab[0xF0];
        for (int i = 0; i < cnt; ++i) put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void eob() {
        const u32 e = acTab[0];
        put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ u32 finish() {   // returns the length in bits
        if (n > 0 && word < kPrivWords) words[word] = (u32)(acc << (32 - n));
        return (u32)(word * 32 + n);
    }
};

// Second walk, for the blocks that did not fit their private words: straight to the target.
struct EmitVisitor {
    const u32* dcTab;
    const u32* acTab;
    BitTarget out;
    long long word;   // index of the word being filled
    u64 acc;          // pending bits, right aligned
    int n;            // number of pending bits (< 32 between calls)
    bool first;
    __device__ __forceinline__ void flush_word() {
        const u32 w = (u32)(acc >> (n - 32));
        if (first) {
            out.or_word(word, w);
            first = false;
        } else {
            out.store_word(word, w);
        }
        ++word;
        n -= 32;
    }
    __device__ __forceinline__ void put(u32 bits, int len) {
        acc = (acc << len) | bits;
        n += len;
        if (n >= 32) flush_word();
    }
    // Code and amplitude go in together: at most 16 + 11 bits on top of fewer than 32 pending.
    __device__ __forceinline__ void dc(int cat, u32 amp) {
        const u32 e = dcTab[cat];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void ac(int sym, int cat, u32 amp) {
        const u32 e = acTab[sym];
        put(((e & 0xffffu) << cat) | amp, (int)(e >> 16) + cat);
    }
    __device__ __forceinline__ void zrl(int cnt) {
        const u32 e = acTab[0xF0];
        for (int i = 0; i < cnt; ++i) put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void eob() {
        const u32 e = acTab[0];
        put(e & 0xffffu, (int)(e >> 16));
    }
    __device__ __forceinline__ void finish() {
        if (n > 0) out.or_word(word, (u32)(acc << (32 - n)));
    }
};

// The same into a stage in LDS, without an accumulator: every token is ORed into the (zeroed)
// stage where it belongs, two words at a time.  No branch per token (when a lane's pending word
// fills up differs from lane to lane, so the flush above is paid at nearly every token of a dense
// block), half the instructions per token.  `pos` counts bits from bit 31 of stage word 0.
struct StageVisitor {
    const u32* dcTab;
    const u32* acTab;
    u32* stage;
    u32 pos;
    __device__ __forceinline__ void put(u32 bits, u32 len) {   // len <= 27, bits == 0 when len == 0
        const u32 sh = pos & 31u;
        const u64 v = (u64)bits << ((64u - sh - len) & 63u);
        u32* w = stage + (pos >> 5);
        atomicOr(w, (u32)(v >> 32));
        atomicOr(w + 1, (u32)v);
        pos += len;
    }
    __device__ __forceinline__ void dc(int cat, u32 amp) {
        const u32 e = dcTab[cat];
        put(((e & 0xffffu) << cat) | amp, (e >> 16) + (u32)cat);
    }
    __device__ __forceinline__ void ac(int sym, int cat, u32 amp) {
        const u32 e = acTab[sym];
        put(((e & 0xffffu) << cat) | amp, (e >> 16) + (u32)cat);
    }
    __device__ __forceinline__ void zrl(int cnt) {
        const u32 e = acTab[0xF0];
        for (int i = 0; i < cnt; ++i) put(e & 0xffffu, e >> 16);
    }
    __device__ __forceinline__ void eob() {
        const u32 e = acTab[0];
        put(e & 0xffffu, e >> 16);
    }
};

// ---- single-pass bit writer --------------------------------------------------------------
// Bit lengths, stream positions and the bit writer in one pass over the coefficients.  A
// tile is 256 consecutive blocks of one image.  The stream position after a block is a
// function of the position before it: P + bits, or, behind the last block of a restart
// interval, roundup8(P + bits) + the marker that follows (the next interval's RSTn, or EOI).
// A run of blocks composes to "P + h" (no interval ends inside) or "roundup8(P + h) + m",
// an associative summary, so tiles chain with a decoupled look-back: every tile publishes
// its summary as soon as its lengths are known, then its first wave reads the summaries of
// its predecessors back to the nearest tile whose absolute position is already known.
// Tiles take their numbers from a counter, so every predecessor of a running tile is
// running or finished and the wait always ends.
struct PosSummary {
    u32 x;   // h, bit 31: an interval ends inside
    u32 y;   // m
};

__device__ __forceinline__ PosSummary pos_combine(PosSummary a, PosSummary b) {   // a precedes b
    PosSummary r;
    if ((a.x >> 31) == 0) {
        r.x = a.x + b.x;
        r.y = b.y;
    } else {
        const u32 bh = b.x & 0x7fffffffu;
        r.x = a.x;
        r.y = (b.x >> 31) ? ((a.y + bh + 7u) & ~7u) + b.y : a.y + bh;
    }
    return r;
}

__device__ __forceinline__ u64 pos_apply(PosSummary s, u64 p) {
    const u32 h = s.x & 0x7fffffffu;
    return (s.x >> 31) ? ((p + h + 7) & ~7ull) + s.y : p + h;
}

constexpr u64 kTileAggregate = 1ull << 62, kTilePosition = 2ull << 62, kTilePayload = (1ull << 62) - 1;
constexpr int kLookBackSpins = 1 << 18;

__device__ __forceinline__ PosSummary shfl_down_summary(PosSummary v, int d, int lane) {
    PosSummary o;
    o.x = __shfl_down(v.x, d, 64);
    o.y = __shfl_down(v.y, d, 64);
    if (lane + d >= 64) o.x = o.y = 0;
    return o;
}

// Absolute bit position at which tile `tile` starts; p0 is the position of the stream's
// first block.  Called by all 64 lanes of one wave.  Returns false if a predecessor never
// reported (cannot happen with ordered tile numbers; bounded so a fault cannot hang the GPU).
__device__ __forceinline__ bool look_back(const u64* state, long long tile, u64 p0, u64& position) {
    const int lane = threadIdx.x & 63;
    PosSummary acc = {0u, 0u};
    int spins = 0;
    for (long long base = tile - 1;; base -= 64) {
        const long long idx = base - lane;
        u64 st = idx < 0 ? (kTilePosition | p0) : __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__ballot((st >> 62) == 0) != 0) {
            if (++spins > kLookBackSpins) return false;
            __builtin_amdgcn_s_sleep(2);
            if ((st >> 62) == 0) st = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u64 known = __ballot((st >> 62) == 2);
        const int nearest = known ? __builtin_ctzll(known) : 64;   // lanes below it hold summaries only
        PosSummary v = {0u, 0u};
        if (lane < nearest) {
            v.x = (u32)((st & kTilePayload) >> 30);
            v.y = (u32)(st & 0x3fffffffu);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v = pos_combine(shfl_down_summary(v, d, lane), v);   // older tiles first
        PosSummary window;
        window.x = __shfl(v.x, 0, 64);
        window.y = __shfl(v.y, 0, 64);
        acc = pos_combine(window, acc);
        if (known) {
            const u64 p = __shfl(st & kTilePayload, nearest, 64);
            position = pos_apply(acc, p);
            return true;
        }
    }
}

#include "entropy_groups.h"
#include "entropy_dense.h"

// It pays from about a quarter of a million blocks on (tiles of 256 blocks, natural content, against
// k_histogram: 49 k blocks 19 against 13 us, 98 k 20 against 17, 197 k 21 against 20, 393 k 23 against 29,
// 0.79 million 36 against 41, 6.3 million 181 against 208; 8192 x 8192 4:2:0 noise, 1.6 million: 99 against 117).
// (Diagnostic switch JPEGENC_HIST_TILES_MIN_BLOCKS: another threshold, read at every call so that a
// test can set it.)
static long long histogram_tiles_min_blocks() {
    const char* env = getenv("JPEGENC_HIST_TILES_MIN_BLOCKS");
    // (From where the bit writer on groups takes over, kGrpMinBlocks, although the pass itself only wins from a quarter
    // of a million blocks on -- 197 k blocks: 21 against 20 us: the bit writer chooses its tile size by the
    // statistics this pass keeps of the densest tiles.)
    return env ? atoll(env) : 150000;
}

static hipError_t launch_histogram_tiles(const KGeom& g, const int16_t* d_coef, u64* d_hist, hipStream_t s, HistogramExport* ex) {
    const long long tilesPerImage = (g.blocksPerImage + kHistTileBlocks - 1) / kHistTileBlocks;
    long long grid = (tilesPerImage + 3) / 4;
    // Five workgroups per CU: measured best for the 16384 x 16384 image (tiles of 128 blocks: 1024 workgroups
    // 199 us, 1280: 191, 2048: 195; tiles of 256: 768: 190, 1024: 186, 1280: 182, 1536: 186, 2048: 207).
    // (Diagnostic switch JPEGENC_HIST_GRID: another number.)
    static const long long gridCap = getenv("JPEGENC_HIST_GRID") ? atoll(getenv("JPEGENC_HIST_GRID")) : 1280;
    const long long cap = g.batch > 1 ? (4096 / g.batch > 8 ? 4096 / g.batch : 8) : gridCap;
    // (Giving every workgroup the same number of turns was measured and is worse than the plain cap -- 16384 x 16384:
    // 1229 workgroups 153 us, 1280: 141 -- and so is, since full tiles are counted in registers, taking up to six
    // workgroups per CU as they are: 8192 x 8192 4:2:0 noise, 1536 workgroups 83 us, 1280: 77.)
    if (grid > cap) grid = cap;
    HistogramExport none = {nullptr, 0, nullptr, nullptr, nullptr};
    if (ex != nullptr) ex->expected += (unsigned long long)grid;
    HistogramExport given = ex != nullptr ? *ex : none;
    if (grid > kHistogramMaxGrid) given.stats = nullptr;   // (no room for a word per workgroup: no tile statistics)
    hipLaunchKernelGGL(k_histogram_tiles<4>, dim3((unsigned)grid, (unsigned)g.batch), dim3(256), 0, s, g, d_coef, d_hist, given);
    return hipGetLastError();
}

static long long group_tiles_per_image(long long blocksPerImage, int tileBlocks) {
    return ((blocksPerImage + tileBlocks - 1) / tileBlocks + 3) / 4 * 4;
}

static long long group_min_blocks() {
    // (Diagnostic switch JPEGENC_GROUP_MIN_BLOCKS: another threshold, e.g. 0 to test it on small images.)
    const char* minEnv = getenv("JPEGENC_GROUP_MIN_BLOCKS");
    return minEnv ? atoll(minEnv) : kGrpMinBlocks;
}

long long group_slot_bytes(const KGeom& g) {
    return group_tiles_per_image(g.blocksPerImage, kGrpTileBlocks) * g.batch * (long long)kGrpSlotWords * 4;
}

// Blocks that are short on average (photographic content; k_encode_dense has the long ones): the
// first walk keeps each block's bits in a few private LDS words and the second is a copy.
__global__ __launch_bounds__(kEntThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_encode_bits(KGeom g, const int16_t* __restrict__ coef,
                                                             const u32* __restrict__ tables, u64* __restrict__ state,
                                                             u64* __restrict__ ivOff, u64* __restrict__ result,
                                                             u32* __restrict__ words, u64* __restrict__ hostResult,
                                                             long long zeroedBytes, u32* __restrict__ ffCount) {
    // ffCount (may be null): per-chunk counts of the stream's 0xFF bytes (zero at the start), kept up
    // to date as the words are written, so that no counting pass has to read the stream again
    // (1080p: entropy stage 39 -> 34 us; 8192 x 8192 noise 4:4:4: 0.508 -> 0.491 ms).
    __shared__ u32 tab[4 * 256];
    constexpr int kStageWords = kEmitStageShort;
    __shared__ u32 stage[kStageWords];
    __shared__ u64 span[2];
    __shared__ PosSummary waveSum[kEntThreads / 64];
    __shared__ u32 priv[kEntThreads * kPrivPitch];
    __shared__ u64 tileStart;
    __shared__ long long tileNo;
    const long long tilesPerImage = (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    const long long tiles = tilesPerImage * g.batch;
    u64* counter = state + tiles;
    if (threadIdx.x == 0) tileNo = (long long)atomicAdd((unsigned long long*)counter, 1ull);
    // One picture: the tables do not depend on the tile, and come in while the ticket is on its way
    // (a round trip to memory that every tile of the launch is queueing for).
    // A symbol the tables have no code for (length 0) gets a length no block can reach, so
    // that a block which uses one shows in its bit count.
    auto stage_tables = [&](const u32* image) {
        for (int i = threadIdx.x; i < 1024; i += kEntThreads) {
            const u32 e = image[i];
            tab[i] = (e >> 16) == 0 ? kNoCodeLength << 16 : e;
        }
    };
    if (g.batch == 1) stage_tables(tables);
    __syncthreads();
    // Tile and block numbers fit 32 bits (see BlockRole); the tile number is the same for
    // the whole workgroup.
    const u32 tile = __builtin_amdgcn_readfirstlane((u32)tileNo);
    const u32 image = g.batch > 1 ? tile / (u32)tilesPerImage : 0u;
    const u32 bl = (tile - image * (u32)tilesPerImage) * kEntThreads + threadIdx.x;
    const long long b = (long long)image * g.blocksPerImage + bl;
    const bool active = bl < (u32)g.blocksPerImage;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long blocksPerInterval = g.mcusPerInterval * g.bpm;
    const BlockRole role = block_role(g, active ? bl : (u32)g.blocksPerImage - 1);

    u32 w[32];
    int fetchedDc = 0;
    if (active) {   // in flight while the tables are staged
        load_block(g, coef, b, w);
        fetchedDc = load_dc(g, coef, dc_predictor_source(role, b));
    }
    if (g.batch > 1) {
        stage_tables(tables + image * g.tableStride);
        __syncthreads();
    }

    long long iv = 0;
    bool firstOfInterval = false, lastOfInterval = false;
    PosSummary mine = {0u, 0u};
    u32 bits = 0;
    const int c = role.c;
    const int pred = dc_predictor_lanes(role, active ? coef_at(w, 0) : 0, fetchedDc, active);
    if (active) {
        iv = g.intervals == g.batch ? (long long)image : (long long)(role.mcu / (u32)g.mcusPerInterval);
        const long long ivFirstBlock = iv * blocksPerInterval;
        long long ivEnd = ivFirstBlock + blocksPerInterval;
        if (ivEnd > g.totalBlocks) ivEnd = g.totalBlocks;
        firstOfInterval = b == ivFirstBlock;
        lastOfInterval = b == ivEnd - 1;
        const int t = c == 0 ? 0 : 1;
        PrivateVisitor pv{tab + 256 * t, tab + 512 + 256 * t, priv + threadIdx.x * kPrivPitch, 0, 0, 0};
        walk_block(w, pred, pv);
        bits = pv.finish();
        if (bits >= kNoCodeLength) {   // report it and write nothing for the block
            atomicOr((unsigned long long*)&result[1], 16ull);
            bits = 0;
        }
        mine.x = bits;
        if (lastOfInterval) {
            mine.x |= 0x80000000u;
            mine.y = iv + 1 < g.intervals ? 8u * (u32)interval_marker_bytes(g, iv + 1) : 0u;
            if (interval_has_eoi(g, iv)) mine.y += 16u;
        }
    }
    // Inclusive scan of the summaries over the workgroup.
    PosSummary inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        PosSummary o;
        o.x = __shfl_up(inc.x, d, 64);
        o.y = __shfl_up(inc.y, d, 64);
        if (lane >= d) inc = pos_combine(o, inc);
    }
    if (lane == 63) waveSum[wave] = inc;
    __syncthreads();
    PosSummary before = {0u, 0u}, whole = {0u, 0u};
#pragma unroll
    for (int i = 0; i < kEntThreads / 64; ++i) {
        if (i < wave) before = pos_combine(before, waveSum[i]);
        whole = pos_combine(whole, waveSum[i]);
    }
    PosSummary excl;
    excl.x = __shfl_up(inc.x, 1, 64);
    excl.y = __shfl_up(inc.y, 1, 64);
    if (lane == 0) excl.x = excl.y = 0;
    excl = pos_combine(before, excl);

    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    if (wave == 0) {
        if (tile > 0 && lane == 0)
            __hip_atomic_store(state + tile, kTileAggregate | ((u64)whole.x << 30) | whole.y, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        u64 start = p0;
        bool ok = true;
        if (tile > 0) ok = look_back(state, tile, p0, start);
        if (lane == 0) {
            if (!ok) {
                start = p0;
                atomicOr((unsigned long long*)&result[1], 2ull);
                if (hostResult) hostResult[2] = 2;
            }
            __hip_atomic_store(state + tile, kTilePosition | pos_apply(whole, start), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            tileStart = start;
        }
    }
    __syncthreads();

    u64 pos = 0;
    int padBits = 0;
    u32 leadMarker = 0;
    bool tailEoi = false;
    if (active) {
        pos = pos_apply(excl, tileStart);
        u64 begin = pos;
        u64 end = pos + bits;
        if (firstOfInterval) {
            const int mk = interval_marker_bytes(g, iv);
            if (mk) {
                leadMarker = 0xffd0u + (u32)((g.firstInterval + iv - 1) & 7);
                begin -= 16;
            }
            ivOff[iv] = begin >> 3;
        }
        if (lastOfInterval) {   // pad the interval to a byte boundary (BitBuffer.cpp:12-16)
            padBits = (int)((8 - (end & 7)) & 7);
            end += padBits;
            if (interval_has_eoi(g, iv)) {
                tailEoi = true;
                end += 16;
            }
            if (b == g.totalBlocks - 1) {
                const u64 bytes = end >> 3;
                ivOff[g.intervals] = bytes;
                result[2] = bytes;
                result[3] = (bytes + kStuffChunk - 1) / kStuffChunk;
                // The stream was ORed into a buffer cleared up to zeroedBytes: beyond, it is garbage.
                if ((long long)bytes > zeroedBytes) atomicOr((unsigned long long*)&result[1], 4ull);
            }
        }
        if (threadIdx.x == 0) span[0] = begin;
        if (threadIdx.x == kEntThreads - 1 || bl == g.blocksPerImage - 1) span[1] = end;
        pos = begin;
    }
    __syncthreads();
    const u64 wordBase = span[0] >> 5;
    const u64 nWords = ((span[1] + 31) >> 5) - wordBase;
    const bool staged = nWords < (u64)kStageWords;   // (a token's second word may be the one behind the last)
    if (staged) {
        for (u32 i = threadIdx.x; i < (u32)nWords; i += kEntThreads) stage[i] = 0;
        __syncthreads();
    }
    if (active) {
        const BitTarget out{staged ? stage : words, staged ? (long long)wordBase : 0, !staged, ffCount, nullptr, 0, 0};
        const u64 start = pos + (leadMarker ? 16 : 0);   // first bit of the block itself
        if (leadMarker) {
            out.or32(pos, leadMarker << 16);
            if (ffCount != nullptr) atomicAdd(&ffCount[pos >> 15], kChunkMarkerDelta);   // the marker's 0xFF is no data (bit position -> 4 KiB chunk)
        }
        if (bits == 0) {
            // a block with a symbol that has no code: flagged above
        } else if (bits <= (u32)(kPrivWords * 32)) {
            const u32* mine32 = priv + threadIdx.x * kPrivPitch;
#pragma unroll
            for (int k = 0; k < kPrivWords; ++k)
                if ((u32)(32 * k) < bits) out.or32(start + 32 * k, mine32[k]);
        } else {
            const int t = c == 0 ? 0 : 1;   // a block too long for its private words: walked once more
            if (staged) {
                StageVisitor vis{tab + 256 * t, tab + 512 + 256 * t, stage, (u32)(start - (wordBase << 5))};
                walk_block(w, pred, vis);
            } else {
                EmitVisitor vis{tab + 256 * t, tab + 512 + 256 * t, out, (long long)(start >> 5), 0, (int)(start & 31), true};
                walk_block(w, pred, vis);
                vis.finish();
            }
        }
        if (padBits != 0 && g.padOnes) out.or32(start + bits, ((1u << padBits) - 1u) << (32 - padBits));
        if (tailEoi) {
            out.or32(start + bits + padBits, 0xffd9u << 16);
            if (ffCount != nullptr) atomicAdd(&ffCount[(start + bits + padBits) >> 15], kChunkMarkerDelta);
        }
    }
    if (staged) {
        __syncthreads();
        if (ffCount != nullptr) {
            copy_out_counted(stage, (u32)nWords, wordBase, words, ffCount);
        } else {
            for (u32 i = threadIdx.x; i < (u32)nWords; i += kEntThreads) {
                const u32 v = __builtin_bswap32(stage[i]);
                if (i == 0 || i == (u32)nWords - 1) {
                    if (v != 0) atomicOr(&words[wordBase + i], v);
                } else {
                    words[wordBase + i] = v;
                }
            }
        }
    }
}

// ---- byte stuffing -----------------------------------------------------------------------
__device__ __forceinline__ int count_ff(u32 w) {
    // A byte is 0xFF iff all its bits are set: fold the 8 bits of every byte into bit 0.
    u32 t = w & (w >> 4);
    t &= t >> 2;
    t &= t >> 1;
    return __popc(t & 0x01010101u);
}

// Is the 0xFF at byte position i (followed by `next`) the first byte of a marker that the
// bit writer placed in the stream?  Only reached for FF D0..D7 / FF D9 pairs, i.e. rarely.
__device__ __forceinline__ bool is_marker_ff(const KGeom& g, const u64* __restrict__ ivOff, u64 bytes, u64 i,
                                             uint8_t next) {
    (void)bytes;
    if (next != 0xd9 && (next < 0xd0 || next > 0xd7)) return false;
    long long lo = 0, hi = g.intervals - 1;   // last interval starting at or before i
    while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ivOff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    if (next == 0xd9) return interval_has_eoi(g, lo) && i + 2 == ivOff[lo + 1];
    return ivOff[lo] == i && interval_marker_bytes(g, lo) != 0;
}

// Loads 16 stream bytes (plus one of lookahead) and returns a 16-bit mask of the data bytes
// equal to 0xFF, i.e. those that need a 0x00 behind them (BitBuffer.cpp:156-161, 171-176,
// 190-195).  Only bytes of the stream count.
// noMarkers: the caller knows that none of these bytes begins a marker (the counting pass found
// none in the chunk), which saves the look-up of the interval.  *foundMarker is set when one does.
__device__ __forceinline__ u32 stuff_mask(const KGeom& g, const uint8_t* __restrict__ buf,
                                          const u64* __restrict__ ivOff, u64 bytes, long long i0, const uint4& q,
                                          bool noMarkers = false, bool* foundMarker = nullptr) {
    u32 ff = ff_bits(q.x) | (ff_bits(q.y) << 4) | (ff_bits(q.z) << 8) | (ff_bits(q.w) << 12);
    const long long left = (long long)bytes - i0;
    if (left < 16) ff &= left > 0 ? (1u << (int)left) - 1u : 0u;
    if (ff == 0 || noMarkers) return ff;
    // One round per 0xFF byte of the piece (one or two as a rule) instead of one per byte.
    const u64 lo = (u64)q.x | ((u64)q.y << 32), hi = (u64)q.z | ((u64)q.w << 32);
    u32 mask = ff;
    for (u32 todo = ff; todo != 0; todo &= todo - 1) {
        const int j = __builtin_ctz(todo), n = j + 1;
        const uint8_t nxt = (uint8_t)(n < 8 ? lo >> (8 * n) : (n < 16 ? hi >> (8 * (n - 8)) : (u64)buf[i0 + 16]));
        const bool maybe = (nxt >= 0xd0 && nxt <= 0xd7) || nxt == 0xd9;
        if (maybe && is_marker_ff(g, ivOff, bytes, (u64)(i0 + j), nxt)) {
            mask &= ~(1u << j);
            if (foundMarker) *foundMarker = true;
        }
    }
    return mask;
}

constexpr int kChunkGroup = 64;   // chunks whose counts are also kept as one sum

// For long streams: the counts of kChunkGroup consecutive chunks added up, one wave per sum.
__global__ __launch_bounds__(256) void k_group_sums(const u32* __restrict__ ffCount, const u64* __restrict__ result,
                                                    u64* __restrict__ groupSum) {
    static_assert(kChunkGroup == 64, "one lane per chunk of a group");
    const long long chunks = (long long)((result[2] + kStuffChunk - 1) / kStuffChunk);
    const int lane = threadIdx.x & 63;
    const long long grp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp * kChunkGroup >= chunks) return;
    const long long c = grp * kChunkGroup + lane;
    u64 sum = c < chunks ? ffCount[c] & kChunkCountMask : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) groupSum[grp] = sum;
}

// Stuffing bytes in front of chunk ch, added up by the 256 threads of a workgroup from the
// per-chunk counts; cheap enough once per workgroup that the stream needs no device-wide
// scan kernel.  Long streams (groupSum set) have the counts of whole groups added up already.
__device__ __forceinline__ u64 chunk_prefix(const u64* __restrict__ groupSum, const u32* __restrict__ ffCount,
                                            long long ch, u64* lds) {
    u64 v = 0;
    if (groupSum != nullptr) {
        const long long grp = ch / kChunkGroup;
        for (long long i = threadIdx.x; i < grp; i += 256) v += groupSum[i];
        const long long c = grp * kChunkGroup + threadIdx.x;
        if (threadIdx.x < kChunkGroup && c < ch) v += ffCount[c] & kChunkCountMask;
    } else {
        // Eight loads in flight per thread: one after the other they would cost a memory
        // latency each.
        for (long long i0 = threadIdx.x; i0 < ch; i0 += 8 * 256) {
            u32 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = i0 + k * 256 < ch ? ffCount[i0 + k * 256] & kChunkCountMask : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
    }
    u64 total;
    block_exclusive_scan(v, lds, total);
    __syncthreads();   // lds is free again
    return total;
}

// Copies the stream to the output with the stuffing bytes inserted.  A workgroup's 4 KiB
// chunk becomes one contiguous output range; every lane builds the output of its sixteen
// bytes in registers and stores it as dwords at the output's own 4-byte phase (a stage in LDS
// that the lanes fill byte by byte costs the kernel half its time).  In a batch the stream of
// image k goes behind header k in slot k; a chunk that holds the end of one image and the
// start of the next is written byte by byte instead.

// Interval (= image of a batch) that contains stream byte i.
__device__ __forceinline__ long long interval_of(const KGeom& g, const u64* __restrict__ ivOff, u64 i) {
    long long lo = 0, hi = g.intervals - 1;
    while (lo < hi) {
        const long long mid = (lo + hi + 1) >> 1;
        if (ivOff[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Output position of stream byte i of image k, given the stuffing bytes in front of it.
__device__ __forceinline__ long long batch_position(const BatchOutput& bo, const u64* __restrict__ ivOff, long long k,
                                                    u64 i, u64 ffBefore) {
    return k * bo.slotBytes + (long long)bo.hdrLen[k] + (long long)(i - ivOff[k]) + (long long)(ffBefore - bo.ffStart[k]);
}

// kOwnBytes: a lane stores exactly its own output bytes, at whatever address they begin (the target
// takes unaligned stores), instead of the aligned dwords that begin among them -- fewer instructions,
// which is what a long stream is bound by (16384 x 16384: 42.5 -> 38.8 us, 8192 x 8192 noise 4:4:4:
// 133 -> 108 us), but a longer way through a single chunk (1080p: 6.9 -> 11.4 us), so short streams
// keep the other form.
// Exclusive scan of one value per thread over a workgroup of 256, the wave's part with row shifts and row
// broadcasts (wave_scan_add) instead of six shuffles through the LDS crossbar, each of which is waited for:
// the stuffing pass does one scan per 4 KiB chunk, one chunk behind the other.  lds: 4 values; the barriers
// order the accesses to it only (the caller has loads in flight).
__device__ __forceinline__ u32 block_exclusive_scan_dpp(u32 v, u32* lds, u32& total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 inc = wave_scan_add(v);
    lds_barrier();
    if (lane == 63) lds[wave] = inc;
    lds_barrier();
    u32 base = 0, sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 t = lds[i];
        if (i < wave) base += t;
        sum += t;
    }
    total = sum;
    return base + inc - v;
}

// A lane's sixteen stream bytes (the first `valid` of them count) with a zero byte behind every 0xFF flagged in
// `mask`, stored at dst, whatever its alignment: the lane's output as a string of bytes in registers (stores to memory
// need no alignment on this target, so no lane has to know its neighbours' bytes).
__device__ __forceinline__ void stuff_store_own(const uint4& q, u32 mask, int valid, uint8_t* __restrict__ dst) {
    u32 str[8] = {q.x, q.y, q.z, q.w, 0u, 0u, 0u, 0u};
    u32 inserted = 0;
    for (u32 todo = mask; todo != 0; todo &= todo - 1, ++inserted) {
        const u32 z = (u32)__builtin_ctz(todo) + inserted + 1;   // index the zero byte gets
        const u32 w = z >> 2, r = z & 3u;
        const u32 keep = (1u << (8 * r)) - 1u, above = 0xffffff00u << (8 * r);
#pragma unroll
        for (int i = 7; i >= 0; --i) {   // bytes from z on move up by one
            const u32 up = i > 0 ? __builtin_amdgcn_alignbyte(str[i], str[i > 0 ? i - 1 : 0], 3) : str[0] << 8;
            if ((u32)i > w) str[i] = up;
            else if ((u32)i == w) str[i] = (str[i] & keep) | (up & above);
        }
    }
    // byte b of the string, for the few bytes that are not stored as whole words
    auto byte_at = [&](u32 b) {
        const u32 i = b >> 2;
        const u32 lo4 = i & 2u ? (i & 1u ? str[3] : str[2]) : (i & 1u ? str[1] : str[0]);
        const u32 hi4 = i & 2u ? (i & 1u ? str[7] : str[6]) : (i & 1u ? str[5] : str[4]);
        return (uint8_t)((i & 4u ? hi4 : lo4) >> (8 * (b & 3u)));
    };
    if (valid == 16) {
        const uint4 first = make_uint4(str[0], str[1], str[2], str[3]);
        __builtin_memcpy(dst, &first, 16);
        // what the insertions pushed out of the first sixteen bytes: one byte as a rule
        if (inserted != 0) {
            dst[16] = (uint8_t)str[4];
            for (u32 b = 17; b < 16u + inserted; ++b) dst[b] = byte_at(b);
        }
    } else {   // the stream's last bytes
        for (u32 b = 0; b < (u32)valid + inserted; ++b) dst[b] = byte_at(b);
    }
}

template <bool kBatch, bool kOwnBytes>
__global__ __launch_bounds__(256) void k_stuff_bytes(KGeom g, const uint8_t* __restrict__ buf,
                                                     const u64* __restrict__ ivOff, const u32* __restrict__ ffCount,
                                                     const u64* __restrict__ groupSum, u64* __restrict__ result,
                                                     uint8_t* __restrict__ out, long long prefixLen,
                                                     long long outCapacity, BatchOutput bo, u64* __restrict__ hostResult) {
    __shared__ u64 lds[4];
    const u64 bytes = result[2];
    const long long chunks = (long long)((bytes + kStuffChunk - 1) / kStuffChunk);
    // A workgroup takes a run of consecutive chunks: the stuffing bytes in front of the
    // first come from the counting pass, those in front of the others from its own work.
    // The runs differ by one chunk at most, so that every workgroup of the grid has work.
    const long long chFirst = chunks * blockIdx.x / gridDim.x;
    const long long chEnd = chunks * (blockIdx.x + 1) / gridDim.x;
    if (chunks == 0 && !kBatch && blockIdx.x == 0 && threadIdx.x == 0 && hostResult) {
        // No stream at all (a bit writer that gave up never reports the stream's end): the
        // host still has to see why.
        hostResult[0] = (u64)prefixLen;
        hostResult[3] = result[1] & ~1ull;
    }
    if (chFirst >= chEnd) return;
    // A chunk's bytes (sixteen per lane and the four behind them) are asked for while the chunk in
    // front of it is worked on.
    __shared__ u32 scanLds[4];
    uint4 qNext = make_uint4(0, 0, 0, 0);
    u32 eNext = 0, flagsNext = 0;
    auto fetch = [&](long long ch) {
        const long long i0 = ch * kStuffChunk + (long long)threadIdx.x * 16;
        qNext = make_uint4(0, 0, 0, 0);
        eNext = 0;
        if ((u64)i0 < bytes) qNext = *(const uint4*)(buf + i0);
        if (!kOwnBytes && (u64)i0 + 16 < bytes) eNext = *(const u32*)(buf + i0 + 16);
        // does a marker begin in this chunk, or (bit 1) in the next one, where the last lane's four bytes lie
        flagsNext = (ffCount[ch] > kChunkCountMask ? 1u : 0u) | (ch + 1 < chunks && ffCount[ch + 1] > kChunkCountMask ? 2u : 0u);
    };
    fetch(chFirst);
    u64 ffBefore = chunk_prefix(groupSum, ffCount, chFirst, lds);
    for (long long ch = chFirst; ch < chEnd; ++ch) {
        const long long c0 = ch * kStuffChunk;
        const long long i0 = c0 + (long long)threadIdx.x * 16;
        const uint4 q = qNext;
        const u32 e = eNext, flags = flagsNext;
        if (ch + 1 < chEnd) fetch(ch + 1);
        int valid = 0;
        u32 mask = 0;
        if ((u64)i0 < bytes) {
            valid = (int)(bytes - (u64)i0 < 16 ? bytes - (u64)i0 : 16);
            mask = stuff_mask(g, buf, ivOff, bytes, i0, q, (flags & 1u) == 0);
        }
        u32 total;
        const u32 local = block_exclusive_scan_dpp((u32)(valid + __popc(mask)), scanLds, total);
        // Output range of the chunk: one piece, except for a batch chunk that spans images.
        long long base, limit;
        bool onePiece = true;
        if (kBatch) {
            const long long k = interval_of(g, ivOff, (u64)c0);
            const u64 cEnd = (u64)c0 + kStuffChunk < bytes ? (u64)c0 + kStuffChunk : bytes;
            onePiece = cEnd <= ivOff[k + 1];
            base = batch_position(bo, ivOff, k, (u64)c0, ffBefore);
            limit = (k + 1) * bo.slotBytes;
        } else {
            base = prefixLen + c0 + (long long)ffBefore;
            limit = outCapacity;
        }
        if (onePiece) {
            if (base + (long long)total > limit) {
                if (threadIdx.x == 0) {
                    atomicOr((unsigned long long*)&result[1], 1ull);
                    if (hostResult) hostResult[1] = 1;
                }
            } else if (valid > 0) {
                if (kOwnBytes) {
                    stuff_store_own(q, mask, valid, out + base + (long long)local);
                } else {
                    // The lane's output as a string of bytes in registers: its sixteen bytes, then the
                    // four that follow them in the stream (the lane owns every output dword that
                    // begins among its own bytes, and such a dword may end among its successor's),
                    // with a zero byte inserted behind every 0xFF that is data.
                    u32 str[9] = {q.x, q.y, q.z, q.w, 0u, 0u, 0u, 0u, 0u};
                    u32 todo = mask;
                    if ((u64)i0 + 16 < bytes) {
                        str[4] = e;
                        u32 ef = ff_bits(e) & 7u;
                        const bool mayHold = (flags >> (threadIdx.x == 255 ? 1 : 0)) & 1u;   // a marker may begin among these bytes
                        for (u32 t = ef; t != 0; t &= t - 1) {
                            const int j = __builtin_ctz(t);
                            const uint8_t nxt = (uint8_t)(e >> (8 * (j + 1)));
                            if ((u64)i0 + 16 + j >= bytes || (mayHold && is_marker_ff(g, ivOff, bytes, (u64)(i0 + 16 + j), nxt))) ef &= ~(1u << j);
                        }
                        todo |= ef << 16;
                    }
                    for (u32 inserted = 0; todo != 0; todo &= todo - 1, ++inserted) {
                        const u32 z = (u32)__builtin_ctz(todo) + inserted + 1;   // index the zero byte gets
                        const u32 w = z >> 2, r = z & 3u;
                        const u32 keep = (1u << (8 * r)) - 1u, above = 0xffffff00u << (8 * r);
#pragma unroll
                        for (int i = 8; i >= 0; --i) {   // bytes from z on move up by one
                            const u32 up = i > 0 ? __builtin_amdgcn_alignbyte(str[i], str[i > 0 ? i - 1 : 0], 3) : str[0] << 8;
                            if ((u32)i > w) str[i] = up;
                            else if ((u32)i == w) str[i] = (str[i] & keep) | (up & above);
                        }
                    }
                    const u32 nOut = (u32)valid + (u32)__popc(mask);
                    const long long mine = base + (long long)local, chunkEnd = base + (long long)total;
                    uint8_t* dst = out + mine;
                    const u32 a = (u32)(-(long long)(uintptr_t)dst) & 3u;   // bytes in front of the first dword that begins here
                    // The chunk's first bytes up to a dword boundary (the dword belongs to another
                    // chunk, or to the header): byte by byte.
                    if (threadIdx.x == 0)
                        for (u32 b = 0; b < a && b < nOut; ++b) dst[b] = (uint8_t)(str[0] >> (8 * b));
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const u32 off = a + 4u * i;
                        if (off < nOut) {
                            const u32 v = __builtin_amdgcn_alignbyte(str[i + 1], str[i], a);
                            if (mine + off + 4 <= chunkEnd) {
                                *(u32*)(dst + off) = v;
                            } else {   // the chunk ends inside this dword
                                for (u32 b = 0; mine + off + b < chunkEnd; ++b) dst[off + b] = (uint8_t)(v >> (8 * b));
                            }
                        }
                    }
                }
            }
        } else if (valid > 0) {
            // Batch chunk across an image boundary: every byte finds its own slot.
            const u32 ws[4] = {q.x, q.y, q.z, q.w};
            long long k = interval_of(g, ivOff, (u64)i0);
            u64 ff = ffBefore + (local - (u64)((long long)i0 - c0));   // stuffing bytes in front of byte i0
            bool overflow = false;
            for (int j = 0; j < valid; ++j) {
                const u64 i = (u64)(i0 + j);
                while (i >= ivOff[k + 1]) ++k;
                const long long pos = batch_position(bo, ivOff, k, i, ff);
                const int extra = (mask >> j) & 1;
                if (pos + 1 + extra <= (k + 1) * bo.slotBytes) {
                    out[pos] = (uint8_t)(ws[j >> 2] >> ((j & 3) * 8));
                    if (extra) out[pos + 1] = 0x00;
                } else {
                    overflow = true;
                }
                ff += extra;
            }
            if (overflow) atomicOr((unsigned long long*)&result[1], 1ull);
        }
        if (!kBatch && ch == chunks - 1 && threadIdx.x == 0) {
            result[0] = (u64)(base + (long long)total);
            if (hostResult) {
                hostResult[0] = (u64)(base + (long long)total);
                hostResult[3] = result[1] & ~1ull;   // what the bit writer flagged (it has finished)
            }
        }
        const u64 chunkBytes = bytes - (u64)c0 < (u64)kStuffChunk ? bytes - (u64)c0 : (u64)kStuffChunk;
        ffBefore += total - chunkBytes;   // what the chunk wrote beyond its own bytes
    }
}

// The same for long streams of one image, a wave per chunk and no workgroup: the wave's lanes take a chunk as two
// halves of two rows of 1 KiB each, sixteen bytes per lane and row, the output offsets come from a scan inside the
// wave (both rows in one, sixteen bits each), and nothing is waited for but memory -- the scan over a workgroup
// cost two barriers per chunk, with every wave's loads and stores behind them.  The stuffing bytes in front of the
// wave's first chunk come from the group sums (which the host therefore always asks for).
__global__ __launch_bounds__(64) void k_stuff_waves(KGeom g, const uint8_t* __restrict__ buf, const u64* __restrict__ ivOff,
                                                    const u32* __restrict__ ffCount, const u64* __restrict__ groupSum,
                                                    u64* __restrict__ result, uint8_t* __restrict__ out, long long prefixLen,
                                                    long long outCapacity, u64* __restrict__ hostResult) {
    const int lane = threadIdx.x;
    const u64 bytes = result[2];
    const long long chunks = (long long)((bytes + kStuffChunk - 1) / kStuffChunk);
    const long long chFirst = chunks * blockIdx.x / gridDim.x;
    const long long chEnd = chunks * (blockIdx.x + 1) / gridDim.x;
    if (chunks == 0 && blockIdx.x == 0 && lane == 0 && hostResult) {   // (see k_stuff_bytes)
        hostResult[0] = (u64)prefixLen;
        hostResult[3] = result[1] & ~1ull;
    }
    if (chFirst >= chEnd) return;
    constexpr long long kHalf = kStuffChunk / 2;
    uint4 qNext[2];
    u32 countNext = 0;
    auto fetch = [&](long long half) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long i0 = half * kHalf + r * 1024 + lane * 16;
            qNext[r] = make_uint4(0, 0, 0, 0);
            if ((u64)i0 < bytes) qNext[r] = *(const uint4*)(buf + i0);
        }
        countNext = ffCount[half >> 1];
    };
    fetch(2 * chFirst);
    u64 ffBefore;
    {
        u64 v = 0;
        const long long grp = chFirst / kChunkGroup;
        for (long long i = lane; i < grp; i += 64) v += groupSum[i];
        const long long c = grp * kChunkGroup + lane;
        if (c < chFirst) v += ffCount[c] & kChunkCountMask;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (u64)__shfl_xor((long long)v, d, 64);
        ffBefore = v;
    }
    bool full = false;
    for (long long half = 2 * chFirst; half < 2 * chEnd; ++half) {
        const long long h0 = half * kHalf;
        const uint4 q[2] = {qNext[0], qNext[1]};
        const bool noMarkers = countNext <= kChunkCountMask;   // no marker begins in this chunk
        if (half + 1 < 2 * chEnd) fetch(half + 1);
        int valid[2];
        u32 mask[2], n[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long i0 = h0 + r * 1024 + lane * 16;
            valid[r] = 0;
            mask[r] = 0;
            if ((u64)i0 < bytes) {
                valid[r] = (int)(bytes - (u64)i0 < 16 ? bytes - (u64)i0 : 16);
                mask[r] = stuff_mask(g, buf, ivOff, bytes, i0, q[r], noMarkers);
            }
            n[r] = (u32)valid[r] + (u32)__popc(mask[r]);
        }
        // (a row puts out 2 KiB at most: sixteen bits hold every sum)
        const u32 incl = wave_scan_add(n[0] | (n[1] << 16));
        const u32 tot = (u32)__builtin_amdgcn_readlane(incl, 63);
        const u32 total = (tot & 0xffffu) + (tot >> 16);
        const u32 local[2] = {(incl & 0xffffu) - n[0], (tot & 0xffffu) + (incl >> 16) - n[1]};
        const long long base = prefixLen + h0 + (long long)ffBefore;
        if (base + (long long)total > outCapacity) {
            full = true;
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (valid[r] > 0) stuff_store_own(q[r], mask[r], valid[r], out + base + (long long)local[r]);
        }
        const u64 halfBytes = (u64)h0 >= bytes ? 0ull : (bytes - (u64)h0 < (u64)kHalf ? bytes - (u64)h0 : (u64)kHalf);
        ffBefore += total - halfBytes;   // what the half wrote beyond its own bytes
    }
    if (lane == 0) {
        if (full) {
            atomicOr((unsigned long long*)&result[1], 1ull);
            if (hostResult) hostResult[1] = 1;
        }
        if (chEnd == chunks) {   // the stream's last chunk was this wave's
            const u64 end = (u64)prefixLen + bytes + ffBefore;
            result[0] = end;
            if (hostResult) {
                hostResult[0] = end;
                hostResult[3] = result[1] & ~1ull;   // what the bit writer flagged (it has finished)
            }
        }
    }
}

// Batch: stuffing bytes in front of each image's stream (and, for k = batch, in total).
__global__ __launch_bounds__(256) void k_image_ff_start(KGeom g, const uint8_t* __restrict__ buf,
                                                        const u64* __restrict__ ivOff, const u32* __restrict__ ffCount,
                                                        const u64* __restrict__ groupSum,
                                                        const u64* __restrict__ result, u64* __restrict__ ffStart) {
    __shared__ u64 lds[4];
    const u64 bytes = result[2];
    const long long k = blockIdx.x;
    const u64 pos = ivOff[k];
    const long long ch = (long long)(pos / kStuffChunk);
    const long long i0 = ch * kStuffChunk + (long long)threadIdx.x * 16;
    int cnt = 0;
    if ((u64)i0 < pos) {
        const uint4 q = *(const uint4*)(buf + i0);
        u32 mask = stuff_mask(g, buf, ivOff, bytes, i0, q);
        const u64 span = pos - (u64)i0;
        if (span < 16) mask &= (1u << span) - 1u;
        cnt = __popc(mask);
    }
    u64 total;
    block_exclusive_scan((u64)cnt, lds, total);
    __syncthreads();
    const u64 before = chunk_prefix(groupSum, ffCount, ch, lds);
    if (threadIdx.x == 0) ffStart[k] = before + total;
}

// Batch: header k to the front of slot k.
__global__ __launch_bounds__(256) void k_copy_headers(BatchOutput bo) {
    const long long k = blockIdx.x;
    const uint8_t* src = bo.hdr + k * bo.hdrPitch;
    uint8_t* dst = bo.out + k * bo.slotBytes;
    const u32 n = bo.hdrLen[k];
    for (u32 i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// Everything the single-pass bit writer and the stuffing pass expect to find cleared, in one
// launch: the reachable part of the unstuffed stream, the tile states with their counter,
// the result record.  Also moves the file header from its staging place to the output.
__global__ __launch_bounds__(256) void k_prepare_entropy(uint4* __restrict__ stream, long long stream16,
                                                         u64* __restrict__ tileState, long long tileWords,
                                                         u64* __restrict__ result, const uint8_t* __restrict__ hdr,
                                                         int hdrLen, uint8_t* __restrict__ out,
                                                         const u32* __restrict__ pending, u32* __restrict__ tables,
                                                         u32* __restrict__ ffCount, long long ffWords) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    for (long long i = tid; i < ffWords; i += step) ffCount[i] = 0;   // (when the placing kernels count the 0xFF bytes)
    if (pending != nullptr) {
        // Tables and header still sit in pinned host memory: bring them over (the device
        // copy serves later launches) and take the header for the output from there.
        const int words = kTabImageWords + (hdrLen + 3) / 4;
        for (long long i = tid; i < words; i += step) tables[i] = pending[i];
        hdr = (const uint8_t*)(pending + kTabImageWords);
    }
    for (long long i = tid; i < stream16; i += step) stream[i] = make_uint4(0, 0, 0, 0);
    for (long long i = tid; i < tileWords; i += step) tileState[i] = 0;
    if (tid < 4) result[tid] = 0;
    for (long long i = tid; i < hdrLen; i += step) out[i] = hdr[i];
}

static hipError_t run_entropy_stage(const KGeom& g, const int16_t* d_coef, const u32* d_tables,
                                    const EntropyWorkspace& ws, uint8_t* d_out, long long prefixLen,
                                    long long outCapacity, const BatchOutput* bo, hipStream_t s) {
    // Bit writer on groups (tiles of ws.groupTile blocks, one per wave) unless the restart
    // intervals are shorter than it takes, or the diagnostic switch asks for the other one.
    // It pays from a few hundred thousand blocks on; below, a wave's walk through a whole tile
    // is a longer chain than the image has work to hide (1080p: 37 -> 44 us).
    // (Diagnostic switch JPEGENC_GROUP_MIN_BLOCKS: another threshold, e.g. 0 to test it on small images.)
    const long long minBlocks = group_min_blocks();
    // (Diagnostic switch JPEGENC_WRITER=dense: the writer for long blocks whatever the content; read at every call so
    // that a test can set it.)
    const char* writer = getenv("JPEGENC_WRITER");
    const bool forceDense = writer != nullptr && writer[0] == 'd';
    const bool groups = ws.groupTile > 0 && !old_entropy_kernels() && !forceDense && g.totalBlocks >= minBlocks &&
                        (g.intervals == g.batch || g.mcusPerInterval * g.bpm >= kGrpMinIntervalBlocks);
    const long long tilesPerImage = groups ? group_tiles_per_image(g.blocksPerImage, ws.groupTile)
                                           : (g.blocksPerImage + kEntThreads - 1) / kEntThreads;
    const long long tileCount = tilesPerImage * g.batch;
    // The bit writer ORs into the stream, so the part it can reach starts out as zeros.
    long long zero = ws.streamBound < 0 ? ws.unstuffedCap : (ws.streamBound + kStuffChunk + 63) / 16 * 16;
    if (zero > ws.unstuffedCap) zero = ws.unstuffedCap;
    // (Not when the tiles' bits are placed by k_place_tiles and k_merge_edges: those write every
    // word of the stream exactly once, with plain stores.)
    const bool deferred = groups && ws.tileSlots != nullptr && ws.groupTile == kGrpTileBlocks;
    // The kernels that write the stream keep the per-chunk counts of its 0xFF bytes.
    const bool shortBlocks = ws.avgBlockBits <= 128 && !forceDense;   // long blocks (noise, q=100) do not fit the private words; unknown lengths count as short
    const bool dense = !groups && !shortBlocks;   // (k_encode_dense writes every word once as well)
    const long long clear16 = deferred || dense ? 0 : zero / 16;
    // Chunks the stream can have at most: from the bound that was cleared.
    long long chunkBound = zero / kStuffChunk + 1;
    if (chunkBound > ws.maxChunks) chunkBound = ws.maxChunks;
    // (every thread four 16-byte stores, or a handful of the small words: the per-chunk counts, the tile states)
    long long prepWork = clear16 > tileCount / 2 ? clear16 : tileCount / 2;
    if (chunkBound / 4 > prepWork) prepWork = chunkBound / 4;
    long long prepGrid = prepWork / (256 * 4);
    // (At least eight workgroups: the table image comes from pinned host memory, a word per thread, and a workgroup
    // that takes several turns waits for the bus every time -- 4096 x 4096 through the dense writer: 13.5 -> 5 us.)
    prepGrid = prepGrid < 8 ? 8 : (prepGrid > 2048 ? 2048 : prepGrid);
    // Tile states: the chaining tiles start from zeros (and a zero ticket counter behind them); deferred
    // tiles write theirs before anybody reads them, and only the word k_tile_positions counts its waves
    // off with has to be zero.
    const long long deferredRuns = (tileCount + 63) / 64;
    u64* clearStates = deferred ? ws.tileState + tileCount + deferredRuns : ws.tileState;
    // (k_encode_dense: the tiles' states and the words behind them that start at zero)
    const long long clearStateWords = deferred ? 1 : (dense ? dense_zeroed_words(tileCount) : tileCount + 2);
    hipLaunchKernelGGL(k_prepare_entropy, dim3((unsigned)prepGrid), dim3(256), 0, s, (uint4*)ws.unstuffed, clear16,
                       clearStates, clearStateWords, ws.result, ws.header, bo ? 0 : ws.headerLen, d_out,
                       bo ? nullptr : ws.pendingTables, const_cast<u32*>(d_tables), ws.ffCount, chunkBound);
    if (deferred) {
        // Nobody waits: the tiles leave their bits in slots, two small kernels find where every
        // tile starts, a third puts the bits there (a tile that chains with a look-back spends a
        // quarter of its life waiting for the slowest of the thousands of tiles in front of it).
        hipLaunchKernelGGL(k_encode_groups<true>, dim3((unsigned)(tileCount / 4)), dim3(256), 0, s, g, d_coef, d_tables,
                           ws.tileState, ws.ivOff, ws.result, (u32*)ws.unstuffed, ws.groupTile, zero, ws.tileSlots, nullptr);
        const u32 runs = (u32)((tileCount + 63) / 64);
        u64* runStart = ws.tileState + tileCount;   // (the buffer is sized for tiles of 16 blocks)
        hipLaunchKernelGGL(k_tile_positions, dim3((runs + kPositionRunsPerWave - 1) / kPositionRunsPerWave), dim3(64), 0, s, g,
                           ws.tileState, runStart, (u32)tileCount, runStart + runs);   // (a word the clearing kernel has zeroed)
        hipLaunchKernelGGL(k_place_tiles, dim3((unsigned)((tileCount + 3) / 4)), dim3(256), 0, s, g, ws.tileState, runStart,
                           ws.tileSlots, ws.ivOff, ws.result, (u32*)ws.unstuffed, (u32)tileCount, zero, ws.ffCount);
        hipLaunchKernelGGL(k_merge_edges, dim3((unsigned)((4 * tileCount + 255) / 256)), dim3(256), 0, s, g,
                           ws.tileSlots + kGrpSlotEdges, (u32)kGrpSlotWords, 2u, (u32*)ws.unstuffed, (u32)tileCount, ws.ffCount);
    } else if (groups)
        hipLaunchKernelGGL(k_encode_groups<false>, dim3((unsigned)(tileCount / 4)), dim3(256), 0, s, g, d_coef, d_tables, ws.tileState,
                           ws.ivOff, ws.result, (u32*)ws.unstuffed, ws.groupTile, zero, nullptr, ws.ffCount);
    else if (shortBlocks)
        hipLaunchKernelGGL(k_encode_bits, dim3((unsigned)tileCount), dim3(kEntThreads), 0, s, g, d_coef, d_tables,
                           ws.tileState, ws.ivOff, ws.result, (u32*)ws.unstuffed, bo ? nullptr : ws.hostResult, zero,
                           ws.ffCount);
    else {
        // As many workgroups as the device holds at once (four per CU, by registers and LDS); each takes tile
        // after tile.  (Diagnostic switch JPEGENC_DENSE_GRID: another number.)
        static const long long denseGrid = getenv("JPEGENC_DENSE_GRID") ? atoll(getenv("JPEGENC_DENSE_GRID")) : 1024;
        unsigned grid = (unsigned)(tileCount < denseGrid ? tileCount : denseGrid);
        grid = grid < 1u ? 1u : (grid > kDenseMaxGrid ? kDenseMaxGrid : grid);
        // Tile numbers by XCD from three tiles per workgroup on (4096 x 4096 4:4:4, 3072 tiles: 72 against 84 us).
        // (Diagnostic switch JPEGENC_DENSE_PER_XCD_FROM, read at every call so that a test can set it.)
        const char* fromText = getenv("JPEGENC_DENSE_PER_XCD_FROM");
        const long long perXcdFrom = fromText ? atoll(fromText) : 3072;
        hipLaunchKernelGGL(k_encode_dense, dim3(grid), dim3(kEntThreads), 0, s, g, d_coef, d_tables, ws.tileState, ws.ivOff,
                           ws.result, (u32*)ws.unstuffed, bo ? nullptr : ws.hostResult, zero, ws.ffCount, tileCount >= perXcdFrom ? 1u : 0u);
        hipLaunchKernelGGL(k_encode_dense_rest, dim3(grid < 256u ? grid : 256u), dim3(kEntThreads), 0, s, g, d_coef, d_tables, ws.tileState, ws.ivOff,
                           ws.result, (u32*)ws.unstuffed, zero, ws.ffCount);
        hipLaunchKernelGGL(k_merge_edges, dim3((unsigned)((2 * tileCount + 255) / 256)), dim3(256), 0, s, g,
                           (const u32*)(ws.tileState + dense_zeroed_words(tileCount)), 4u, 1u, (u32*)ws.unstuffed, (u32)tileCount, ws.ffCount);
    }
    // As many workgroups as the device holds at once (five per CU): a workgroup's run of chunks is
    // one chain (a chunk's bytes are asked for while the chunk in front is worked on), a second
    // round of workgroups would start that chain all over.  (Diagnostic switch JPEGENC_STUFF_GRID.)
    static const long long gridCap = getenv("JPEGENC_STUFF_GRID") ? atoll(getenv("JPEGENC_STUFF_GRID")) : 1280;

    long long chunkGrid = chunkBound < gridCap ? chunkBound : gridCap;
    if (chunkGrid < 1) chunkGrid = 1;
    // Up to this many chunks a workgroup adds the counts in front of its first chunk up
    // directly (64 loads per thread at most); beyond, group sums come first.
    const u64* groupSum = nullptr;
    if (chunkBound > (ws.groupSumsFrom > 0 ? ws.groupSumsFrom : 16384)) {
        const long long groups = chunkBound / kChunkGroup + 1;
        hipLaunchKernelGGL(k_group_sums, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, s, ws.ffCount, ws.result,
                           ws.groupSum);
        groupSum = ws.groupSum;
    }
    // Streams of several chunks per workgroup store every lane's own bytes (see k_stuff_bytes).
    // (Diagnostic switch JPEGENC_STUFF_FORM=0 / 1: always the aligned form / always the other; read at every call so
    // that a test can set it.)
    const char* stuffForm = getenv("JPEGENC_STUFF_FORM");
    const bool ownBytes = stuffForm ? stuffForm[0] == '1' : chunkBound >= 4 * gridCap;
    if (bo) {
        hipLaunchKernelGGL(k_image_ff_start, dim3((unsigned)(g.batch + 1)), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                           ws.ffCount, groupSum, ws.result, bo->ffStart);
        hipLaunchKernelGGL(k_copy_headers, dim3((unsigned)g.batch), dim3(256), 0, s, *bo);
        if (ownBytes)
            hipLaunchKernelGGL((k_stuff_bytes<true, true>), dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                               ws.ffCount, groupSum, ws.result, bo->out, 0, 0, *bo, nullptr);
        else
            hipLaunchKernelGGL((k_stuff_bytes<true, false>), dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                               ws.ffCount, groupSum, ws.result, bo->out, 0, 0, *bo, nullptr);
    } else {
        BatchOutput none = {};
        // Streams long enough to have group sums take the wave-per-chunk form, six waves per SIMD (8192 x 8192 noise
        // 4:4:4, 196 MB: 104 -> 87 us with 6144 waves, 99 with 4096, 92 with 5120, 94 with 8192; 4:2:0: 59 -> 53 us;
        // the 57 MB of the 16384 x 16384 picture would gain the 4 us that the group sums cost it).
        // (Diagnostic switch JPEGENC_STUFF_WAVES: another number of waves, 0 for the other forms.)
        static const long long waveGrid = getenv("JPEGENC_STUFF_WAVES") ? atoll(getenv("JPEGENC_STUFF_WAVES")) : 6144;
        if (ownBytes && groupSum != nullptr && waveGrid > 0)
            hipLaunchKernelGGL(k_stuff_waves, dim3((unsigned)(chunkBound < waveGrid ? chunkBound : waveGrid)), dim3(64), 0, s, g, ws.unstuffed,
                               ws.ivOff, ws.ffCount, groupSum, ws.result, d_out, prefixLen, outCapacity, ws.hostResult);
        else if (ownBytes)
            hipLaunchKernelGGL((k_stuff_bytes<false, true>), dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                               ws.ffCount, groupSum, ws.result, d_out, prefixLen, outCapacity, none, ws.hostResult);
        else
            hipLaunchKernelGGL((k_stuff_bytes<false, false>), dim3((unsigned)chunkGrid), dim3(256), 0, s, g, ws.unstuffed, ws.ivOff,
                               ws.ffCount, groupSum, ws.result, d_out, prefixLen, outCapacity, none, ws.hostResult);
    }
    return hipGetLastError();
}

hipError_t launch_entropy(const KGeom& g, const int16_t* d_coef, const u32* d_tables, const EntropyWorkspace& ws,
                          uint8_t* d_out, long long prefixLen, long long outCapacity, int writeEoi, hipStream_t s) {
    (void)writeEoi;   // carried in g.writeEoi
    return run_entropy_stage(g, d_coef, d_tables, ws, d_out, prefixLen, outCapacity, nullptr, s);
}

hipError_t launch_entropy_batch(const KGeom& g, const int16_t* d_coef, const u32* d_tables,
                                const EntropyWorkspace& ws, const BatchOutput& bo, hipStream_t s) {
    return run_entropy_stage(g, d_coef, d_tables, ws, nullptr, 0, 0, &bo, s);
}

}  // namespace jpeghip
