// Entropy stage in two passes over different data -- included by entropy.hip inside namespace jpeghip.
//
// The histograms and the bit stream are made of the same symbols (Encoder::calculateDCValues /
// calculateACValues, Encoder.cpp:260-352): run lengths and categories depend on the
// coefficients only, not on the Huffman tables.  Finding them is the expensive part, so it
// is done once.  k_symbol_records walks the non-empty groups of eight coefficients like the
// bit writer on groups does (entropy_groups.h), counts every symbol into the histograms and
// leaves it behind as a 32-bit record, in scan order, in a slot of its tile.  Once the tables
// exist the records become bits: k_record_lengths adds up how long every tile's part of the
// stream is, k_tile_positions turns the lengths into start positions, k_emit_records writes the
// bits where they belong -- four records per lane and step, a table lookup each, one prefix sum,
// no coefficient is looked at again and no tile waits for another.
//
// A record: bits 7..0 the symbol (DC: the category; AC: run << 4 | category, 0xF0 and 0x00 for
// the zero-run-length and end-of-block symbols), bits 9..8 the table (DC-Y, DC-C, AC-Y, AC-C),
// bits 31..16 the amplitude bits; or bit 10 set: a restart interval ends here, bits 31..11 hold
// its number inside the band.

constexpr int kRecPerTile = 6144;        // records a tile's slot holds (24 per block; photographic content has 8 to 12)
constexpr int kRecMaxItems = kGrpTileBlocks * 9;
constexpr u32 kRecEnd = 0x400u;
constexpr int kRecMaxIntervals = 1 << 21;
constexpr int kRecStageWords = 512;      // records of one step gathered in LDS so that they leave in whole lines (a step has 64 to 832)
constexpr int kEmitStageWords = 240;     // 256 records of 27 bits, the padding and markers of eight interval ends, slack

// ---- pass 1: symbols --------------------------------------------------------------------------
// Tiles of 256 blocks, one per wave, taken in rounds by a grid that may be smaller than the
// number of tiles (a workgroup's histogram flush costs as much as a tile).  recCount[tile]
// receives the number of records; *recFlag receives `generation` (a number the host raises from
// launch to launch, so that the word never needs clearing) when a slot was too small -- the
// histograms are complete all the same.
__global__ __launch_bounds__(256) void k_symbol_records(KGeom g, const int16_t* __restrict__ coef, u64* __restrict__ hist,
                                                        HistogramExport ex, u32* __restrict__ records,
                                                        u32* __restrict__ recCount, u32* __restrict__ recFlag, u32 generation) {
    __shared__ u32 lh[4 * 256];   // DC-Y, DC-C, AC-Y, AC-C (Encoder.cpp:368-416)
    __shared__ unsigned short items[4][kRecMaxItems + 2];
    __shared__ unsigned char roleTab[4][kGrpTileBlocks];
    __shared__ u32 outStage[4][kRecStageWords];
    __shared__ bool lastOne;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 1024; i += 256) lh[i] = 0;
    __syncthreads();
    const u32 tilesPerImage = (((u32)g.blocksPerImage + (u32)kGrpTileBlocks - 1) / (u32)kGrpTileBlocks + 3u) & ~3u;
    const u32 quads = tilesPerImage / 4u;
    bool overflow = false;
    // No workgroup barrier inside the loop: the waves go their own ways.
    for (u32 quad = blockIdx.x; quad < quads; quad += gridDim.x) {
        const u32 tile = quad * 4u + (u32)wave;
        GroupTile t;
        t.items = items[wave];
        t.cap = kRecMaxItems;
        t.role = roleTab[wave];
        const long long firstLL = (long long)tile * kGrpTileBlocks;
        t.first = (u32)(firstLL < g.blocksPerImage ? firstLL : g.blocksPerImage);
        {
            const long long left = g.blocksPerImage - (long long)t.first;
            t.nBlocks = (int)(left < kGrpTileBlocks ? left : kGrpTileBlocks);
        }
        t.shift = g.coefPitch == 64 ? 6u : 7u;
        t.high = (u32)g.coefHigh;
        t.base = (const char*)coef + (((long long)t.first - (long long)kGrpBaseBlocks) << t.shift);
        t.mask = g.groupMask + t.first;
        t.nItems = 0;
        t.tooMany = false;
        if (t.nBlocks > 0) group_index_tile<true>(g, t);
        u32* out = records + (size_t)tile * kRecPerTile;
        u32 written = 0;
        u32 carryVal = ~0u;
        GroupLoad ld;
        if (t.nItems > 0) group_issue_loads(t, 0, ld);
        for (int first = 0; first < t.nItems; first += 64) {
            GroupFront f;
            group_front(t, first, ld, carryVal, f);
            const int n = t.nItems - first < 64 ? t.nItems - first : 64;
            carryVal = __builtin_amdgcn_readlane(f.val, n - 1);
            group_issue_loads(t, first + 64, ld);   // the next chunk's groups arrive while this one is looked at

            typedef short s2 __attribute__((ext_vector_type(2)));
            u32 amp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // v for v >= 0, v - 1 (the one's complement of |v|) for v < 0, Encoder.cpp:281-284
                const s2 v = __builtin_bit_cast(s2, f.d[i]);
                amp[i] = __builtin_bit_cast(u32, (s2)(v + (v >> 15)));
            }
            const u32 acSel = (2u + f.chroma) << 8;
            const u32 zr = (u32)f.zr;
            const u32 mine = f.boundary ? 1u : (f.real ? zr + (u32)__popc(f.mm & 0xffu) + (f.needEob ? 1u : 0u) : 0u);
            const u32 incl = wave_scan_add(mine);
            const u32 total = __builtin_amdgcn_readlane(incl, 63);
            if (written + total > (u32)kRecPerTile) overflow = true;   // the slot is full: what does not fit is dropped
            const u32 at = incl - mine;
            // The step's records are gathered in LDS and leave as consecutive words, kRecStageWords
            // at a time (one round, except for steps of very busy groups).
            for (u32 part = 0; part < total; part += (u32)kRecStageWords) {
                const u32 lim = total - part < (u32)kRecStageWords ? total - part : (u32)kRecStageWords;
                u32* dst = outStage[wave];
                const bool count = part == 0;
                if (f.boundary && at - part < lim) {
                    const u32 bl = t.first + f.lb;
                    const u32 iv = (bl / (u32)g.bpm) / (u32)g.mcusPerInterval;
                    dst[at - part] = kRecEnd | (iv << 11);
                }
                if (f.real) {
                    if (zr != 0) {
                        if (count) atomicAdd(&lh[(acSel | 0xF0u)], zr);
                        for (u32 z = 0; z < zr; ++z)
                            if (at + z - part < lim) dst[at + z - part] = acSel | 0xF0u;
                    }
                    u32 slot = at + zr - part;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int v = (int)(short)(f.d[j >> 1] >> (16 * (j & 1)));
                        const u32 cat = (u32)__builtin_amdgcn_frexp_expf((float)v);   // bits of |v|, 0 for 0
                        const u32 run = (u32)__builtin_clz(f.X << (16 - j));
                        u32 key = acSel | ((run & 15u) << 4) | cat;
                        if (j == 0 && f.isK0) key = (f.chroma << 8) | cat;
                        const u32 rec = (__builtin_amdgcn_ubfe(amp[j >> 1], 16 * (j & 1), cat) << 16) | key;
                        if ((f.mm >> j) & 1u) {
                            if (count) atomicAdd(&lh[key], 1u);
                            if (slot < lim) dst[slot] = rec;
                            ++slot;
                        }
                    }
                    if (f.needEob) {
                        if (count) atomicAdd(&lh[acSel], 1u);
                        if (slot < lim) dst[slot] = acSel;
                    }
                }
                // The same wave wrote and reads: only the LDS counter has to drain.
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                for (u32 i = lane; i < lim; i += 64)
                    if (written + part + i < (u32)kRecPerTile) out[written + part + i] = dst[i];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
            written += total;
        }
        if (lane == 0) recCount[tile] = written < (u32)kRecPerTile ? written : (u32)kRecPerTile;
    }
    if (__ballot(overflow) != 0 && lane == 0) atomicMax(recFlag, generation);
    __syncthreads();
    // The histograms leave the workgroup as k_histogram's do.
    if (ex.host == nullptr) {
        for (int i = threadIdx.x; i < 1024; i += 256)
            if (lh[i] != 0) atomicAdd(&hist[i], (u64)lh[i]);
        return;
    }
    u64 old[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // all in flight before the first is waited for
        const u32 v = lh[k * 256 + threadIdx.x];
        old[k] = v != 0 ? atomicAdd(&hist[k * 256 + threadIdx.x], (u64)v) : 0;
    }
    u64 seen = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) seen |= old[k];
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the additions have returned (see flush_histograms)
    if (seen == ~0ull) lh[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) lastOne = atomicAdd(ex.counter, 1ull) + 1 == ex.expected;
    __syncthreads();
    if (!lastOne) return;
    for (int i = threadIdx.x; i < 1024; i += 256)
        ex.host[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- pass 2: bits -----------------------------------------------------------------------------
// What a lane makes of its four records: the tokens (code and amplitude in bits 26..0, length in
// bits 31..27, 0 for a record that is not a symbol), and where an interval ends among them (4: nowhere).
struct RecordQuad {
    u32 tok[4];
    u32 len[4];
    int end;
    u32 iv;
};

// The four records of a lane in the step that starts at record `first`; records behind the n-th read as 0.
__device__ __forceinline__ uint4 load_quad(const u32* __restrict__ rec, u32 first, u32 n) {
    const u32 i0 = first + 4u * (u32)(threadIdx.x & 63);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (i0 < n) r = *(const uint4*)(rec + i0);
    return r;
}

__device__ __forceinline__ void decode_quad(uint4 r, u32 first, u32 n, const u32* tab, RecordQuad& q) {
    const u32 i0 = first + 4u * (u32)(threadIdx.x & 63);
    const u32 w[4] = {r.x, r.y, r.z, r.w};
    q.end = 4;
    q.iv = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool there = i0 + (u32)k < n;
        const bool isEnd = there && (w[k] & kRecEnd) != 0;
        if (isEnd) {
            q.end = k;
            q.iv = w[k] >> 11;
        }
        const u32 e = tab[w[k] & 0x3ffu];
        const bool sym = there && !isEnd;
        q.tok[k] = sym ? ((e & 0x07ffffffu) | (w[k] >> 16)) | (e & 0xf8000000u) : 0u;
        q.len[k] = sym ? e >> 27 : 0u;
    }
}

// A tile's records are read four steps ahead: a wave that waits for every kilobyte before it asks
// for the next spends its life waiting.
constexpr int kRecAhead = 4;

// Token templates of one table set in LDS: the code shifted left by the category, so that the
// amplitude bits can be ORed in, and the total number of bits in bits 31..27 (entries of
// `tables`: length << 16 | code).
__device__ __forceinline__ void load_token_templates(const u32* __restrict__ tables, u32* tab) {
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const u32 e = tables[i], cat = (u32)i & 15u;
        const u32 len = e >> 16, code = e & 0xffffu;
        tab[i] = (len == 0 || len + cat > 27u) ? kTokAbsent : ((code << cat) | ((len + cat) << 27));
    }
}

// Bits the interval end behind interval iv adds behind its padding: EOI, or the next interval's RSTn.
__device__ __forceinline__ u32 end_marker_bits(const KGeom& g, u32 iv) {
    const bool last = iv + 1 == (u32)g.intervals;
    return last ? (g.writeEoi ? 16u : 0u) : 16u;
}

// The step's summaries in lane order: inclusive scan of the lanes' own summaries.
__device__ __forceinline__ PosSummary quad_summary(const KGeom& g, const RecordQuad& q) {
    PosSummary inc = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        PosSummary one = {q.len[k], 0u};
        if (k == q.end) {
            one.x = 0x80000000u;
            one.y = end_marker_bits(g, q.iv);
        }
        inc = pos_combine(inc, one);
    }
    return inc;
}

__device__ __forceinline__ PosSummary wave_scan_summary(PosSummary inc) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        PosSummary o;
        o.x = __shfl_up(inc.x, d, 64);
        o.y = __shfl_up(inc.y, d, 64);
        if (lane >= d) inc = pos_combine(o, inc);
    }
    return inc;
}

// Pass 2a: how long each tile's part of the stream is, as a function of where it starts
// (PosSummary, packed as x << 32 | y into summary[tile]).  One wave per tile.
__global__ __launch_bounds__(256) void k_record_lengths(KGeom g, const u32* __restrict__ records, const u32* __restrict__ recCount,
                                                        const u32* __restrict__ recFlag, u32 generation,
                                                        const u32* __restrict__ tables, u64* __restrict__ summary,
                                                        u64* __restrict__ result) {
    __shared__ u32 tab[4 * 256];
    if (*recFlag == generation) {   // a slot of the first pass was too small: the host takes the other bit writer
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr((unsigned long long*)&result[1], 32ull);
        return;
    }
    load_token_templates(tables, tab);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u32 tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    const u32 n = recCount[tile];
    const u32* rec = records + (size_t)tile * kRecPerTile;
    PosSummary run = {0u, 0u};
    u32 maxLen = 0;
    uint4 ahead[kRecAhead];
#pragma unroll
    for (int a = 0; a < kRecAhead; ++a) ahead[a] = load_quad(rec, 256u * a, n);
    for (u32 first = 0; first < n; first += 256) {
        RecordQuad q;
        decode_quad(ahead[0], first, n, tab, q);
#pragma unroll
        for (int a = 0; a + 1 < kRecAhead; ++a) ahead[a] = ahead[a + 1];
        ahead[kRecAhead - 1] = load_quad(rec, first + 256u * kRecAhead, n);
        const u32 m01 = q.len[0] > q.len[1] ? q.len[0] : q.len[1], m23 = q.len[2] > q.len[3] ? q.len[2] : q.len[3];
        maxLen = maxLen > m01 ? maxLen : m01;
        maxLen = maxLen > m23 ? maxLen : m23;
        if (__ballot(q.end != 4) == 0) {
            const u32 len = q.len[0] + q.len[1] + q.len[2] + q.len[3];
            run = pos_combine(run, PosSummary{(u32)__builtin_amdgcn_readlane(wave_scan_add(len), 63), 0u});
        } else {
            const PosSummary inc = wave_scan_summary(quad_summary(g, q));
            run = pos_combine(run, PosSummary{(u32)__builtin_amdgcn_readlane(inc.x, 63), (u32)__builtin_amdgcn_readlane(inc.y, 63)});
        }
    }
    if (lane == 0) summary[tile] = ((u64)run.x << 32) | run.y;
    if (__ballot(maxLen >= 31u) != 0 && lane == 0) atomicOr((unsigned long long*)&result[1], 16ull);   // a symbol without a code
}

// Pass 2b: summaries to start positions in two small steps.  k_group_positions: one wave per
// run of 64 tiles replaces each tile's summary by the summary of the tiles in front of it inside
// the run, and notes the run's own summary.  k_run_positions: one workgroup turns the runs'
// summaries into the positions at which the runs start.  A tile then starts at
// pos_apply(summary[tile], runStart[tile / 64]).
__global__ __launch_bounds__(64) void k_group_positions(u64* __restrict__ summary, u64* __restrict__ runSummary, u32 tiles) {
    const int lane = threadIdx.x;
    const u32 t = blockIdx.x * 64u + (u32)lane;
    const u64 v = t < tiles ? summary[t] : 0ull;
    const PosSummary inc = wave_scan_summary(PosSummary{(u32)(v >> 32), (u32)v});
    PosSummary excl;
    excl.x = __shfl_up(inc.x, 1, 64);
    excl.y = __shfl_up(inc.y, 1, 64);
    if (lane == 0) excl.x = excl.y = 0;
    if (t < tiles) summary[t] = ((u64)excl.x << 32) | excl.y;
    if (lane == 63) runSummary[blockIdx.x] = ((u64)inc.x << 32) | inc.y;
}

__global__ __launch_bounds__(1024) void k_run_positions(KGeom g, u64* __restrict__ runSummary, u32 runs) {
    __shared__ PosSummary part[1024];
    const u32 per = (runs + 1023u) / 1024u;   // 1 up to 65536 tiles
    const u32 t0 = threadIdx.x * per, t1 = t0 + per < runs ? t0 + per : runs;
    PosSummary mine = {0u, 0u};
    for (u32 t = t0; t < t1; ++t) {
        const u64 v = runSummary[t];
        mine = pos_combine(mine, PosSummary{(u32)(v >> 32), (u32)v});
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        PosSummary o = {0u, 0u};
        if ((int)threadIdx.x >= d) o = part[threadIdx.x - d];
        __syncthreads();
        if ((int)threadIdx.x >= d) part[threadIdx.x] = pos_combine(o, part[threadIdx.x]);
        __syncthreads();
    }
    PosSummary before = {0u, 0u};
    if (threadIdx.x > 0) before = part[threadIdx.x - 1];
    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    for (u32 t = t0; t < t1; ++t) {
        const u64 v = runSummary[t];
        runSummary[t] = pos_apply(before, p0);
        before = pos_combine(before, PosSummary{(u32)(v >> 32), (u32)v});
    }
}

// Pass 2c: the bits, at their final positions (see k_group_positions for where a tile starts).
// A step's bits are gathered in the stage (word 0 is the stream word the step starts in) and
// leave as whole words.
__global__ __launch_bounds__(256) void k_emit_records(KGeom g, const u32* __restrict__ records, const u32* __restrict__ recCount,
                                                      const u32* __restrict__ recFlag, u32 generation,
                                                      const u32* __restrict__ tables, const u64* __restrict__ summary,
                                                      const u64* __restrict__ runStart, u64* __restrict__ ivOff, u64* __restrict__ result, u32* __restrict__ words,
                                                      long long zeroedBytes) {
    __shared__ u32 tab[4 * 256];
    __shared__ u32 stage[4][kEmitStageWords];
    if (*recFlag == generation) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * kEmitStageWords; i += 256) (&stage[0][0])[i] = 0;
    load_token_templates(tables, tab);
    __syncthreads();
    const u32 tile = blockIdx.x * 4u + (u32)wave;
    u32* st = stage[wave];
    const u32 n = recCount[tile];
    const u32* rec = records + (size_t)tile * kRecPerTile;
    const u64 inRun = summary[tile];
    u64 pos = pos_apply(PosSummary{(u32)(inRun >> 32), (u32)inRun}, runStart[tile >> 6]);   // position of the step's first bit
    uint4 ahead[kRecAhead];
#pragma unroll
    for (int a = 0; a < kRecAhead; ++a) ahead[a] = load_quad(rec, 256u * a, n);
    for (u32 first = 0; first < n; first += 256) {
        RecordQuad q;
        decode_quad(ahead[0], first, n, tab, q);
#pragma unroll
        for (int a = 0; a + 1 < kRecAhead; ++a) ahead[a] = ahead[a + 1];
        ahead[kRecAhead - 1] = load_quad(rec, first + 256u * kRecAhead, n);
        const u64 origin = pos & ~31ull;
        const u64 ends = __ballot(q.end != 4);
        const u32 len = q.len[0] + q.len[1] + q.len[2] + q.len[3];
        u64 next;
        if (ends == 0 && __ballot(len > 64u) == 0) {
            const u32 incl = wave_scan_add(len);
            const u32 at = (u32)(pos - origin) + incl - len;
            next = pos + __builtin_amdgcn_readlane(incl, 63);
            if (len != 0) {
                u64 acc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = (acc << q.len[k]) | (q.tok[k] & 0x07ffffffu);
                const u64 v = acc << (64u - len);   // the lane's bits, from bit 63 down
                const u32 hi = (u32)(v >> 32), lo = (u32)v, sh = at & 31u;
                u32* w = st + (at >> 5);
                atomicOr(w, __builtin_amdgcn_alignbit(0u, hi, sh));
                const u32 w1 = __builtin_amdgcn_alignbit(hi, lo, sh), w2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
                if (w1 != 0) atomicOr(w + 1, w1);
                if (w2 != 0) atomicOr(w + 2, w2);
            }
        } else {
            // Long tokens or an interval end among the step's records: every token is placed on its own.
            const PosSummary inc = wave_scan_summary(quad_summary(g, q));
            PosSummary excl;
            excl.x = __shfl_up(inc.x, 1, 64);
            excl.y = __shfl_up(inc.y, 1, 64);
            if (lane == 0) excl.x = excl.y = 0;
            next = pos_apply(PosSummary{(u32)__builtin_amdgcn_readlane(inc.x, 63), (u32)__builtin_amdgcn_readlane(inc.y, 63)}, pos);
            u64 p = pos_apply(excl, pos);   // the lane's first bit
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == q.end) {
                    // The end pads the stream to a byte (BitBuffer::fillToByteBorder, BitBuffer.cpp:12-16)
                    // and is followed by EOI (the image ends) or the next interval's RSTn.
                    const bool lastIv = q.iv + 1 == (u32)g.intervals;
                    const u64 padded = (p + 7) & ~7ull;
                    if (g.padOnes && padded != p) stage_put(st, (u32)(p - origin), (1u << (u32)(padded - p)) - 1u, (u32)(padded - p));
                    u64 m = padded;
                    if (g.writeEoi && lastIv) {
                        stage_put(st, (u32)(m - origin), 0xffd9u, 16u);
                        m += 16;
                    }
                    ivOff[q.iv + 1] = m >> 3;
                    if (lastIv) {
                        result[2] = m >> 3;
                        result[3] = ((m >> 3) + kStuffChunk - 1) / kStuffChunk;
                        if ((long long)(m >> 3) > zeroedBytes) atomicOr((unsigned long long*)&result[1], 4ull);
                    } else {
                        stage_put(st, (u32)(m - origin), 0xffd0u + (u32)((g.firstInterval + q.iv) & 7), 16u);
                        m += 16;
                    }
                    p = m;
                } else if (q.len[k] != 0) {
                    stage_put(st, (u32)(p - origin), q.tok[k] & 0x07ffffffu, q.len[k]);
                    p += q.len[k];
                }
            }
        }
        // The same wave wrote and reads: only the LDS counter has to drain.
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // Whole words leave; a last, incomplete one stays behind as word 0 of the next step.  Only
        // the tile's first word can be shared with another tile (an atomic at device scope takes
        // microseconds: one per tile and end, not per step).
        const u32 nFull = (u32)((next >> 5) - (origin >> 5));
        for (u32 i = lane; i < nFull; i += 64) {
            const u32 v = st[i];
            if (v != 0) {
                st[i] = 0;
                const u32 o = __builtin_bswap32(v);
                if (i == 0 && first == 0) atomicOr(&words[(origin >> 5) + i], o);
                else words[(origin >> 5) + i] = o;
            }
        }
        if (nFull != 0 && lane == 0) {
            const u32 rest = st[nFull];
            st[nFull] = 0;
            st[0] = rest;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        pos = next;
    }
    if (lane == 0 && n != 0 && (pos & 31) != 0 && st[0] != 0) atomicOr(&words[pos >> 5], __builtin_bswap32(st[0]));
    if (tile == 0 && lane == 0) {
        ivOff[0] = 0;
        if (interval_marker_bytes(g, 0) != 0) atomicOr(&words[0], __builtin_bswap32((0xffd0u + (u32)((g.firstInterval - 1) & 7)) << 16));
    }
}
