// Bit writer for dense content (long blocks: noise, q = 100) -- included by entropy.hip inside
// namespace jpeghip, behind entropy_groups.h.
//
// One lane per block, a tile is 256 consecutive blocks of one image, tiles chain with the
// decoupled look-back (look_back), as in k_encode_bits.  What differs:
//
//  * A workgroup stays and takes tile after tile (dense_draw_ticket).  The tables are staged once, and
//    the next tile's coefficients are asked for before the current tile's words are copied out, so the
//    round trips to memory that start a tile are hidden behind the end of the one before.
//  * A block's 64 coefficients become 64 tokens in registers without a branch (dense_tokens) and
//    are ORed into the stage in LDS (dense_put).
//  * A tile without an interval boundary is staged from bit 0, whatever its place in the
//    stream, and shifted into place when it is copied out: its look-back comes after its bits have
//    been staged, when most tiles in front have reported, not before.
//  * Tiles with an interval boundary (the first and the last of a picture at least), tiles with a
//    run of sixteen zeros or a symbol without a code, and tiles too long for the stage take the
//    general walk (walk_block) twice, as k_encode_bits does for its long blocks: first for the
//    lengths, here, so that the chain of tile positions goes on, then for the bits, in a second
//    kernel (k_encode_dense_rest) that finds the tiles and their starts in a list.  (In one kernel the
//    second walk's registers and pointers cost the common path spills, and a reload from scratch waits
//    for every load in flight, the next tile's coefficients included: 8192 x 8192 4:4:4 noise
//    324 against 281 us.)
//  * Nothing is ORed into memory.  The first and the last word of a tile's range, which it may share
//    with its neighbours, go to a list (two entries per tile) that k_merge_edges turns into stream
//    words afterwards, everything in between is written with plain stores: the stream need not be
//    cleared beforehand (8192 x 8192 4:4:4 noise: 196 MB, 28 us).
#ifndef JPEGENC_DENSE_LOOK
#define JPEGENC_DENSE_LOOK 1   // tiles per lane and step of the look-back
#endif
constexpr int kDenseStageWords = 6144;   // up to 96 bytes per block on average

// Copies nOut words to the stream, from word w0 on: the stage's bits shifted right by s (0..31; word
// -1 of the stage counts as zero).  Every wave takes a contiguous quarter.  The first and the last
// word go to edge[0], edge[1] (word number << 32 | value, kNoEdge for none; see k_merge_edges), the
// 0xFF bytes of the others are added to their chunks' counts (kChunkCountMask).
__device__ __forceinline__ void dense_copy_out(const u32* stage, u32 nOut, u32 s, u64 w0, u32* __restrict__ words,
                                               u32* __restrict__ ffCount, u64* __restrict__ edge) {
    const int lane = threadIdx.x & 63;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (nOut < 2 && threadIdx.x == 0) edge[1] = kNoEdge;
    if (nOut == 0 && threadIdx.x == 0) edge[0] = kNoEdge;
    const u32 quarter = (nOut + 3u) / 4u;
    const u32 begin = wave * quarter;
    const u32 end = begin + quarter < nOut ? begin + quarter : nOut;
    if (begin >= end) return;
    u64 curChunk = (w0 + begin) >> kChunkWordShift;
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
            const u32 hi = i > 0 ? stage[i - 1] : 0u;
            const u32 v = __builtin_bswap32(__builtin_amdgcn_alignbit(hi, stage[i], s));
            const u64 at = w0 + i;
            if (i == 0) edge[0] = (at << 32) | v;
            else if (i == nOut - 1u) edge[1] = (at << 32) | v;
            else {
                words[at] = v;
                n = (u32)__popc(ff_bits(v));
            }
            inCur = (at >> kChunkWordShift) == curChunk;
        }
        const u32 last = base + 63u < end ? base + 63u : end - 1u;
        const u64 lastChunk = (w0 + last) >> kChunkWordShift;   // the same for the whole wave
        if (lastChunk == curChunk) {
            acc += n;
        } else {   // a chunk ends inside these 64 words (at most one does)
            flush(acc + (inCur ? n : 0u));
            acc = inCur ? 0u : n;
            curChunk = lastChunk;
        }
    }
    flush(acc);
}

// A lane's place in a tile.
struct DenseLane {
    u32 image, bl;
    long long b;
    bool active;
    BlockRole role;
};

__device__ __forceinline__ DenseLane dense_lane(const KGeom& g, u32 tile, u32 tilesPerImage, u32 thread) {
    DenseLane l;
    l.image = g.batch > 1 ? tile / tilesPerImage : 0u;
    l.bl = (tile - l.image * tilesPerImage) * kEntThreads + thread;
    l.b = (long long)l.image * g.blocksPerImage + l.bl;
    l.active = l.bl < (u32)g.blocksPerImage;
    l.role = block_role(g, l.active ? l.bl : (u32)g.blocksPerImage - 1);
    return l;
}

// Does an interval begin or end inside the tile?  (The same answer in every lane.)
__device__ __forceinline__ bool dense_boundary_tile(const KGeom& g, u32 tile, u32 tilesPerImage) {
    const u32 image = g.batch > 1 ? tile / tilesPerImage : 0u;
    const u32 first = (tile - image * tilesPerImage) * kEntThreads;
    const u32 blocks = (u32)g.blocksPerImage;
    const u32 behind = first + kEntThreads < blocks ? first + kEntThreads : blocks;   // the block behind the tile's last
    if (behind == blocks) return true;
    const u32 perInterval = g.intervals == g.batch ? blocks : (u32)(g.mcusPerInterval * g.bpm);
    return first % perInterval == 0 || first / perInterval != behind / perInterval;
}

// A lane's block by the general walk: its length, and what it adds to the stream's position.
struct DenseWalked {
    DenseLane me;
    u32 w[32];
    int pred;
    long long iv;
    bool firstOfInterval, lastOfInterval;
    u32 bits;
    PosSummary mine;
};

// tab: the four raw tables in LDS (a length of kNoCodeLength for symbols without a code).  A block that
// uses such a symbol is reported in result[1] (by both kernels: the flag is a flag) and written as nothing.
__device__ __forceinline__ void dense_walk_lengths(const KGeom& g, const int16_t* __restrict__ coef, const u32* tab, u32 tile,
                                                   u32 tilesPerImage, u64* __restrict__ result, DenseWalked& k) {
    // (Inside k_encode_dense's loop this walk is the rare case: nothing of it is to be worked out ahead of the
    // loop and kept in registers through the common one, so the thread's number is made opaque.)
    u32 thread = threadIdx.x;
    asm volatile("" : "+v"(thread));
    k.me = dense_lane(g, tile, tilesPerImage, thread);
    const bool active = k.me.active;
    const long long b = k.me.b;
    int fetchedDc = 0;
    if (active) {
        load_block(g, coef, b, k.w);
        fetchedDc = load_dc(g, coef, dc_predictor_source(k.me.role, b));
    }
    k.pred = dc_predictor_lanes(k.me.role, active ? coef_at(k.w, 0) : 0, fetchedDc, active);
    k.iv = 0;
    k.firstOfInterval = k.lastOfInterval = false;
    k.bits = 0;
    k.mine = PosSummary{0u, 0u};
    if (!active) return;
    const int t = k.me.role.c == 0 ? 0 : 1;
    const long long blocksPerInterval = g.mcusPerInterval * g.bpm;
    k.iv = g.intervals == g.batch ? (long long)k.me.image : (long long)(k.me.role.mcu / (u32)g.mcusPerInterval);
    const long long ivFirstBlock = k.iv * blocksPerInterval;
    long long ivEnd = ivFirstBlock + blocksPerInterval;
    if (ivEnd > g.totalBlocks) ivEnd = g.totalBlocks;
    k.firstOfInterval = b == ivFirstBlock;
    k.lastOfInterval = b == ivEnd - 1;
    LengthVisitor len{tab + 256 * t, tab + 512 + 256 * t, 0u};
    walk_block(k.w, k.pred, len);
    k.bits = len.bits;
    if (k.bits >= kNoCodeLength) {
        atomicOr((unsigned long long*)&result[1], 16ull);
        k.bits = 0;
    }
    k.mine.x = k.bits;
    if (k.lastOfInterval) {
        k.mine.x |= 0x80000000u;
        k.mine.y = k.iv + 1 < g.intervals ? 8u * (u32)interval_marker_bytes(g, k.iv + 1) : 0u;
        if (interval_has_eoi(g, k.iv)) k.mine.y += 16u;
    }
}

// Where the state words of the two kernels lie behind the tiles' own (`state`, [tiles]); see kDenseHead.
__device__ __forceinline__ u64* dense_counter(u64* state, u32 tiles) { return state + tiles; }
__device__ __forceinline__ u64* dense_list_count(u64* state, u32 tiles) { return state + tiles + 1; }
__device__ __forceinline__ u64* dense_xcd_counter(u64* state, u32 tiles, u32 xcd) { return state + tiles + 2 + xcd * kDenseXcdStride; }
__device__ __forceinline__ u64* dense_xcd_chunks(u64* state, u32 tiles, u32 xcd) {
    return state + tiles + kDenseHead + (u64)xcd * (u64)dense_chunks_per_xcd(tiles);
}
__device__ __forceinline__ u64* dense_edges(u64* state, u32 tiles) { return state + dense_zeroed_words(tiles); }
__device__ __forceinline__ u64* dense_list(u64* state, u32 tiles) { return state + dense_zeroed_words(tiles) + 2ull * tiles; }

// Tile numbers.  One counter for the whole device is one address that a thousand workgroups on eight XCDs add
// to at the same moments, and an answer takes microseconds, which is why a workgroup had to ask a tile ahead --
// and a tile asked for that early is started on out of turn, which the tiles behind it wait for.  So the
// workgroups of an XCD count their tickets on a word of their own, whose answer is quick enough to be waited for
// where the number is needed, and the XCD fetches tile numbers from the device's counter a chunk at a time:
// ticket m is tile chunks[m / kDenseChunk] + m % kDenseChunk.  Who draws ticket 0 fetches the chunks the XCD's
// workgroups start with and kDenseAhead more, who draws the first ticket of chunk j fetches chunk j + kDenseAhead
// if it is not there from the start: just early enough (an XCD that holds many tiles which it is not yet working
// on makes the other XCDs' tiles wait); a table entry is written once (valid bit | first tile).
//   The look-back's wait ends because (1) chunks come from the device's counter in order; (2) an XCD's chunks
// have growing numbers -- who fetches chunk k has seen chunk k - 1 in the table -- so that an XCD gives its tiles
// out in growing order and a workgroup's tiles grow; (3) nobody waits for a tile between drawing a ticket and
// doing the fetching that comes with it.  The smallest tile that nobody has drawn yet belongs to a chunk whose
// fetcher holds a smaller tile and waits, at most, for tiles smaller still, all of them drawn by workgroups of
// which the same is true: they all get on, draw again, and the XCD's tickets reach that tile.  A workgroup
// leaves when its ticket is behind the last tile, and all tickets its XCD gives out from then on are.
//   (Which XCD a workgroup runs on is read from the hardware register; nothing but the sharing depends on it.)
__device__ __forceinline__ u32 this_xcd() {
    return (u32)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (u32)(kDenseXcds - 1);   // HW_REG_XCC_ID, bits 3..0
}
constexpr u64 kChunkValid = 1ull << 63;

// A table entry once it is valid; false: it never came (cannot happen: who writes it waits for nobody but
// writers of earlier entries).
__device__ __forceinline__ bool dense_await_chunk(const u64* entry, u64& value) {
    for (int spins = 0; (value & kChunkValid) == 0; ++spins) {
        if (spins > kLookBackSpins) return false;
        __builtin_amdgcn_s_sleep(1);
        value = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// Draws a ticket of the XCD: the tile number, or ~0u and `failed`.  By one thread.  chunkRoom: entries of `chunks`;
// atStart: chunks that the drawer of ticket 0 fetches (> kDenseAhead).
__device__ __forceinline__ u32 dense_draw_ticket(u64* __restrict__ xcdCounter, u64* __restrict__ chunks, u32 chunkRoom,
                                                 u64* __restrict__ counter, u32 atStart, bool& failed) {
    const u64 m = atomicAdd((unsigned long long*)xcdCounter, 1ull);
    const u64 j = m / kDenseChunk;
    const u32 slot = (u32)(m % kDenseChunk);
    if (j + kDenseAhead >= chunkRoom || atStart > chunkRoom) {
        failed = true;
        return ~0u;
    }
    u64 mine = __hip_atomic_load(chunks + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m == 0) {
        const u64 first = atomicAdd((unsigned long long*)counter, (unsigned long long)atStart * kDenseChunk);
        for (u32 i = 0; i < atStart; ++i)
            __hip_atomic_store(chunks + i, kChunkValid | (first + (u64)i * kDenseChunk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (u32)first;
    }
    if (slot == 0 && j + kDenseAhead >= atStart) {
        const u64 k = j + kDenseAhead;
        u64 before = __hip_atomic_load(chunks + k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!dense_await_chunk(chunks + k - 1, before)) {
            failed = true;
            return ~0u;
        }
        const u64 first = atomicAdd((unsigned long long*)counter, (unsigned long long)kDenseChunk);
        __hip_atomic_store(chunks + k, kChunkValid | first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!dense_await_chunk(chunks + j, mine)) {
        failed = true;
        return ~0u;
    }
    return (u32)mine + slot;
}
constexpr int kDenseListTileBits = 24;

struct DenseShared {
    u32 stage[kDenseStageWords];   // (in front of the templates: dense_tokens forms addresses up to 63 rows below them)
    DenseTables templates;         // token templates (dense_tokens), lengths (dense_lengths)
    u64 span[2];
    u64 tileStart;
    PosSummary waveSum[kEntThreads / 64];
    u32 ticket[2];
    u32 surprise;
};

// state: see kDenseHead.  streamBytes: size of the stream's buffer.  perXcd: tile numbers by dense_draw_ticket
// (pictures of many tiles per workgroup), or straight from the device's counter (0): with few tiles, those an
// XCD holds in hand would be missing on another, whose workgroups would leave early.
__global__ __launch_bounds__(kEntThreads) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_encode_dense(KGeom g, const int16_t* __restrict__ coef, const u32* __restrict__ tables, u64* __restrict__ state,
                    u64* __restrict__ ivOff, u64* __restrict__ result, u32* __restrict__ words, u64* __restrict__ hostResult, long long streamBytes,
                    u32* __restrict__ ffCount, u32 perXcd) {
    __shared__ DenseShared sh;
    const u32 tilesPerImage = (u32)((g.blocksPerImage + kEntThreads - 1) / kEntThreads);
    const u32 tiles = tilesPerImage * (u32)g.batch;
    u64* edges = dense_edges(state, tiles);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);

    auto draw_ticket = [&]() {   // thread 0.  (Nothing of this is kept in registers between two tickets.)
        if (perXcd == 0) return (u32)atomicAdd((unsigned long long*)dense_counter(state, tiles), 1ull);
        const u32 xcd = this_xcd();
        // (the chunks an XCD's workgroups start with, were they spread evenly over eight XCDs, and kDenseAhead more)
        const u32 atStart = (gridDim.x / 8u + kDenseChunk - 1u) / kDenseChunk + kDenseAhead;
        bool failed = false;
        const u32 t = dense_draw_ticket(dense_xcd_counter(state, tiles, xcd), dense_xcd_chunks(state, tiles, xcd),
                                        (u32)dense_chunks_per_xcd(tiles), dense_counter(state, tiles), atStart, failed);
        if (failed) atomicOr((unsigned long long*)&result[1], 2ull);
        return t;
    };
    if (threadIdx.x == 0) {
        sh.ticket[0] = draw_ticket();
        sh.surprise = 0;
    }
    auto stage_tables = [&](const u32* image) { load_dense_tables(image, sh.templates); };
    u32 stagedImage = ~0u;
    if (g.batch == 1) {   // (they come in while the ticket is on its way)
        stage_tables(tables);
        stagedImage = 0;
    }
    __syncthreads();
    u32 tile = (u32)__builtin_amdgcn_readfirstlane(sh.ticket[0]);
    u32 parity = 0;
    // The coefficients of the lane's block in the tile at hand, asked for a tile ahead.
    // (Asked for at one place in the loop and without conditions -- a ticket behind the last tile reads the
    // last tile again, lanes behind the image's end its last block: with several places, or a path on which
    // the registers keep their old values, the compiler keeps copies of all 32 alive round the loop and spills.)
    u32 w[32];
    int fetchedDc = 0;
    auto fetch_tile = [&](u32 which) {
        const DenseLane l = dense_lane(g, which < tiles ? which : tiles - 1u, tilesPerImage, threadIdx.x);
        const long long b = l.active ? l.b : (long long)l.image * g.blocksPerImage + g.blocksPerImage - 1;
        load_block(g, coef, b, w);
        fetchedDc = load_dc(g, coef, dc_predictor_source(l.role, b));
    };
    // A tile's start from the tiles in front of it, and its end for those behind; by wave 0.
    auto find_start = [&](u32 which, PosSummary whole) {
        u64 start = p0;
        bool ok = true;
        if (which > 0) ok = look_back(state, which, p0, start);
        // (Everything this wave has asked memory for is here.  Said explicitly, or the compiler, which cannot
        // tell whether the look-back's last poll is still in flight, makes all four waves wait for whatever they
        // have in flight before they copy a tile out.)
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        if (lane == 0) {
            if (!ok) {
                start = p0;
                atomicOr((unsigned long long*)&result[1], 2ull);
                if (hostResult) hostResult[2] = 2;
            }
            __hip_atomic_store(state + which, kTilePosition | pos_apply(whole, start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.tileStart = start;
        }
        return start;
    };
    // The tile whose bits wait in the stage: a tile is copied out while the next one's tokens sit in registers,
    // a tile's time after it has reported its length.  A tile waits for the tiles in front of it to report theirs,
    // and of a thousand tiles in flight some are always late: looked for right after its own bits were staged, a
    // tile's start cost ten polls of the states in front, each a round trip to memory.
    u32 heldTile = 0, heldBits = 0;
    u64 heldStart = 0;
    int held = 0;   // 0: none, 1: staged from bit 0, start yet to be found, 2: staged where it goes (from heldStart & ~31 on)
    auto copy_out_held = [&]() {   // by the whole workgroup; passes a barrier if the start has to be found
        if (held == 1) {
            if (wave == 0) find_start(heldTile, PosSummary{heldBits, 0u});
            lds_barrier();
            heldStart = sh.tileStart;
        }
        if (held != 0) {
            const u64 firstWord = heldStart >> 5;
            const u32 nOut = (u32)(((heldStart + heldBits + 31) >> 5) - firstWord);
            u64* myEdges = edges + 2ull * heldTile;
            if ((long long)((firstWord + nOut) * 4) <= streamBytes) {
                dense_copy_out(sh.stage, nOut, (u32)(heldStart & 31), firstWord, words, ffCount, myEdges);
            } else if (threadIdx.x == 0) {
                myEdges[0] = myEdges[1] = kNoEdge;
                atomicOr((unsigned long long*)&result[1], 4ull);
            }
        }
        held = 0;
    };
    fetch_tile(tile);
    while (tile < tiles) {
        // (What was worked out about this tile when its coefficients were asked for is worked out again
        // rather than kept in registers all the way round the loop: the number is made opaque here.)
        asm volatile("" : "+s"(tile));
        const u32 image = g.batch > 1 ? tile / tilesPerImage : 0u;
        if (image != stagedImage) {   // (batches: the tile's image has tables of its own)
            stage_tables(tables + (long long)image * g.tableStride);
            stagedImage = image;
            lds_barrier();
        }
        DenseBlock dense;
        u32 offset = 0, total = 0;   // where the lane's bits start in the tile's, and how many the tile has
        bool active;
        u32 bits;
        {
            const DenseLane me = dense_lane(g, tile, tilesPerImage, threadIdx.x);
            active = me.active;
            const int pred = dc_predictor_lanes(me.role, coef_at(w, 0), fetchedDc, active);
            dense_tokens(w, pred, sh.templates, me.role.c == 0 ? 0u : 1u, dense);
            bits = active ? dense.bits : 0u;
        }
        // A run of sixteen zeros or more, or a symbol without a code, somewhere in the tile: see below.
        if (__ballot(active && dense.maxTok >= kTokAbsent) != 0 && lane == 0) sh.surprise = 1;
        // The next tile's number, as late as it can be drawn without the other waves waiting for it for long.
        if (threadIdx.x == 0) sh.ticket[parity ^ 1u] = draw_ticket();
        {
            const u32 inc = wave_scan_add(bits);
            if (lane == 63) sh.waveSum[wave].x = inc;
            lds_barrier();
            u32 before = 0;
#pragma unroll
            for (int i = 0; i < kEntThreads / 64; ++i) {
                const u32 v = sh.waveSum[i].x;
                if (i < wave) before += v;
                total += v;
            }
            offset = before + inc - bits;
        }
        const u32 next = (u32)__builtin_amdgcn_readfirstlane(sh.ticket[parity ^ 1u]);
        parity ^= 1u;
        const bool surprise = sh.surprise != 0;
        const bool general = surprise || dense_boundary_tile(g, tile, tilesPerImage) ||
                             ((total + 31u) >> 5) + 2u > (u32)kDenseStageWords;
        if (!general && threadIdx.x == 0 && tile > 0)
            __hip_atomic_store(state + tile, kTileAggregate | ((u64)total << 30), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The tile before goes out now, and the stage is free for this one.
        copy_out_held();
        // The next tile's coefficients are asked for here, and arrive while this tile's bits are staged.
        fetch_tile(next);
        lds_barrier();   // the stage is free, and everybody has seen the flag and read the sums
        if (threadIdx.x == 0) sh.surprise = 0;
        if (general) {
            PosSummary mine = {0u, 0u};
            long long iv = 0;
            bool firstOfInterval = false, lastOfInterval = false, lastOfAll = false, lastOfImage = false;
            {
                // (the thread's number made opaque: nothing of this rare case is to be worked out ahead of the loop
                // and kept in registers through the common one)
                u32 thread = threadIdx.x;
                asm volatile("" : "+v"(thread));
                const DenseLane me = dense_lane(g, tile, tilesPerImage, thread);
                if (surprise) {   // lengths once more, by the look-up that knows long runs
                    // (The tile goes on the list and its tokens are not used again.  The compiler cannot see that:
                    // giving them a value here ends the life of 65 registers where 32 more are wanted.)
#pragma unroll
                    for (int i = 0; i < 64; ++i) dense.tok[i] = 0;
                    dense.eob = 0;
                    u32 again[32];
                    const long long b = me.active ? me.b : (long long)me.image * g.blocksPerImage + g.blocksPerImage - 1;
                    load_block(g, coef, b, again);
                    const int dcBefore = load_dc(g, coef, dc_predictor_source(me.role, b));
                    const int pred = dc_predictor_lanes(me.role, coef_at(again, 0), dcBefore, me.active);
                    bits = me.active ? dense_lengths(again, pred, sh.templates, me.role.c == 0 ? 0u : 1u) : 0u;
                    if (bits >= kDenseNoCode) {   // a symbol without a code: reported, and written as nothing
                        atomicOr((unsigned long long*)&result[1], 16ull);
                        bits = 0;
                    }
                }
                if (me.active) {
                    const long long perInterval = g.mcusPerInterval * g.bpm;
                    iv = g.intervals == g.batch ? (long long)me.image : (long long)(me.role.mcu / (u32)g.mcusPerInterval);
                    long long ivEnd = iv * perInterval + perInterval;
                    if (ivEnd > g.totalBlocks) ivEnd = g.totalBlocks;
                    firstOfInterval = me.b == iv * perInterval;
                    lastOfInterval = me.b == ivEnd - 1;
                    lastOfAll = me.b == g.totalBlocks - 1;
                    lastOfImage = me.bl == (u32)g.blocksPerImage - 1;
                    mine.x = bits;
                    if (lastOfInterval) {
                        mine.x |= 0x80000000u;
                        mine.y = iv + 1 < g.intervals ? 8u * (u32)interval_marker_bytes(g, iv + 1) : 0u;
                        if (interval_has_eoi(g, iv)) mine.y += 16u;
                    }
                }
            }
            const PosSummary inc = wave_scan_summary(mine);
            if (lane == 63) sh.waveSum[wave] = inc;
            lds_barrier();
            PosSummary before = {0u, 0u}, whole = {0u, 0u};
#pragma unroll
            for (int i = 0; i < kEntThreads / 64; ++i) {
                if (i < wave) before = pos_combine(before, sh.waveSum[i]);
                whole = pos_combine(whole, sh.waveSum[i]);
            }
            // Listed: the general walk has to write the tile (k_encode_dense_rest).  Otherwise its tokens are
            // at hand, and go to the stage once the tile's start is known: where the padding in front of a
            // boundary falls, and the interval offsets in bytes, depend on it.  Either way the start is found
            // here and now.
            const bool listed = surprise || (((whole.x & 0x7fffffffu) + whole.y + 64u + 31u) >> 5) + 2u > (u32)kDenseStageWords;
            if (wave == 0) {
                if (tile > 0 && lane == 0)
                    __hip_atomic_store(state + tile, kTileAggregate | ((u64)whole.x << 30) | whole.y, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                const u64 start = find_start(tile, whole);
                if (listed && lane == 0) {
                    const u64 at = atomicAdd((unsigned long long*)dense_list_count(state, tiles), 1ull);
                    dense_list(state, tiles)[at] = (start << kDenseListTileBits) | tile;
                }
            }
            if (!listed) {
                lds_barrier();
                PosSummary excl;
                excl.x = __shfl_up(inc.x, 1, 64);
                excl.y = __shfl_up(inc.y, 1, 64);
                if (lane == 0) excl.x = excl.y = 0;
                excl = pos_combine(before, excl);
                u64 pos = 0;
                int padBits = 0;
                u32 leadMarker = 0;
                bool tailEoi = false;
                if (active) {
                    pos = pos_apply(excl, sh.tileStart);   // first bit of the block itself
                    u64 begin = pos;
                    u64 end = pos + bits;
                    if (firstOfInterval) {
                        if (interval_marker_bytes(g, iv)) {
                            // (a band's first marker, in front of interval 0, is k_merge_edges' to write)
                            if (iv > 0) leadMarker = 0xffd0u + (u32)((g.firstInterval + iv - 1) & 7);
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
                        if (lastOfAll) {
                            const u64 bytes = end >> 3;
                            ivOff[g.intervals] = bytes;
                            result[2] = bytes;
                            result[3] = (bytes + kStuffChunk - 1) / kStuffChunk;
                            if ((long long)bytes > streamBytes) atomicOr((unsigned long long*)&result[1], 4ull);
                        }
                    }
                    if (threadIdx.x == 0) sh.span[0] = begin;
                }
                // (the tile's last block: the last thread's, or the image's last)
                if (active && (threadIdx.x == kEntThreads - 1 || lastOfImage)) sh.span[1] = pos + bits + (u32)padBits + (tailEoi ? 16u : 0u);
                lds_barrier();
                const u64 origin = sh.span[0] & ~31ull;
                heldTile = tile;
                heldStart = origin;
                heldBits = (u32)(sh.span[1] - origin);
                held = 2;
                for (u32 i = threadIdx.x; i < ((heldBits + 31u) >> 5) + 2u; i += kEntThreads) sh.stage[i] = 0;
                lds_barrier();
                if (active) {
                    const u32 at = (u32)(pos - origin);
                    if (leadMarker) {
                        stage_put(sh.stage, at - 16u, leadMarker, 16u);
                        atomicAdd(&ffCount[(pos - 16) >> 15], kChunkMarkerDelta);   // the marker's 0xFF is no data (bit position -> 4 KiB chunk)
                    }
                    dense_put(dense, sh.stage, at);
                    if (padBits != 0 && g.padOnes) stage_put(sh.stage, at + bits, (1u << padBits) - 1u, (u32)padBits);
                    if (tailEoi) {
                        stage_put(sh.stage, at + bits + (u32)padBits, 0xffd9u, 16u);
                        atomicAdd(&ffCount[(pos + bits + padBits) >> 15], kChunkMarkerDelta);
                    }
                }
            }
        } else {
            const u32 nStage = ((total + 31u) >> 5) + 2u;   // (a token's second word may be the one behind the last)
            for (u32 i = threadIdx.x; i < nStage; i += kEntThreads) sh.stage[i] = 0;
            lds_barrier();
            if (active) dense_put(dense, sh.stage, offset);
            heldTile = tile;
            heldBits = total;
            held = 1;
        }
        lds_barrier();   // the stage is complete
        tile = next;
    }
    copy_out_held();
}

// The tiles k_encode_dense has left to the general walk: their bits, at the starts the list gives.
// A workgroup takes every gridDim.x-th entry of the list.
__global__ __launch_bounds__(kEntThreads)
void k_encode_dense_rest(KGeom g, const int16_t* __restrict__ coef, const u32* __restrict__ tables, u64* __restrict__ state,
                         u64* __restrict__ ivOff, u64* __restrict__ result, u32* __restrict__ words, long long streamBytes,
                         u32* __restrict__ ffCount) {
    __shared__ u32 tab[4 * 256];
    __shared__ u32 stage[kDenseStageWords];
    __shared__ u64 span[2];
    __shared__ PosSummary waveSum[kEntThreads / 64];
    __shared__ u32 edgeWord[2];
    const u32 tilesPerImage = (u32)((g.blocksPerImage + kEntThreads - 1) / kEntThreads);
    const u32 tiles = tilesPerImage * (u32)g.batch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 listed = *dense_list_count(state, tiles);
    if (listed > tiles) listed = tiles;   // (nothing is read or written on the say-so of a count or an entry that cannot be)
    u32 stagedImage = ~0u;
    for (u64 at = blockIdx.x; at < listed; at += gridDim.x) {
        const u64 entry = dense_list(state, tiles)[at];
        const u32 tile = (u32)(entry & ((1u << kDenseListTileBits) - 1u));
        const u64 tileStart = entry >> kDenseListTileBits;
        if (tile >= tiles) continue;
        const u32 image = g.batch > 1 ? tile / tilesPerImage : 0u;
        __syncthreads();   // (the tile before is done with tables and stage)
        if (image != stagedImage) {
            const u32* src = tables + (long long)image * g.tableStride;
            for (int i = threadIdx.x; i < 1024; i += kEntThreads) {
                const u32 e = src[i];
                tab[i] = (e >> 16) == 0 ? kNoCodeLength << 16 : e;
            }
            stagedImage = image;
            __syncthreads();
        }
        DenseWalked k;
        dense_walk_lengths(g, coef, tab, tile, tilesPerImage, result, k);
        const bool active = k.me.active;
        const long long b = k.me.b, iv = k.iv;
        const int t = k.me.role.c == 0 ? 0 : 1;
        const u32 bits = k.bits;
        const PosSummary inc = wave_scan_summary(k.mine);
        if (lane == 63) waveSum[wave] = inc;
        __syncthreads();
        PosSummary before = {0u, 0u};
#pragma unroll
        for (int i = 0; i < kEntThreads / 64; ++i)
            if (i < wave) before = pos_combine(before, waveSum[i]);
        PosSummary excl;
        excl.x = __shfl_up(inc.x, 1, 64);
        excl.y = __shfl_up(inc.y, 1, 64);
        if (lane == 0) excl.x = excl.y = 0;
        excl = pos_combine(before, excl);
        u64 pos = 0;
        int padBits = 0;
        u32 leadMarker = 0;
        bool markerInFront = false, tailEoi = false;
        if (active) {
            pos = pos_apply(excl, tileStart);
            u64 begin = pos;
            u64 end = pos + bits;
            if (k.firstOfInterval) {
                if (interval_marker_bytes(g, iv)) {
                    markerInFront = true;
                    // (a band's first marker, in front of interval 0, is k_merge_edges' to write)
                    if (iv > 0) leadMarker = 0xffd0u + (u32)((g.firstInterval + iv - 1) & 7);
                    begin -= 16;
                }
                ivOff[iv] = begin >> 3;
            }
            if (k.lastOfInterval) {   // pad the interval to a byte boundary (BitBuffer.cpp:12-16)
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
                    if ((long long)bytes > streamBytes) atomicOr((unsigned long long*)&result[1], 4ull);
                }
            }
            if (threadIdx.x == 0) span[0] = begin;
            if (threadIdx.x == kEntThreads - 1 || k.me.bl == (u32)g.blocksPerImage - 1) span[1] = end;
            pos = begin;
        }
        __syncthreads();
        const u64 wordBase = span[0] >> 5;
        const u64 wordEnd = (span[1] + 31) >> 5;
        const u64 nWords = wordEnd - wordBase;
        const bool staged = nWords + 1 < (u64)kDenseStageWords;   // (a token's second word may be the one behind the last)
        const bool writable = (long long)(wordEnd * 4) <= streamBytes;
        if (staged) {
            for (u32 i = threadIdx.x; i < (u32)nWords + 1u; i += kEntThreads) stage[i] = 0;
        } else {
            // Straight to memory, with ORs: the tile clears its words first (all but the two at the ends,
            // which go through edgeWord to the list of edges like everybody's).
            if (writable)
                for (u64 i = 1 + threadIdx.x; i + 1 < nWords; i += kEntThreads) words[wordBase + i] = 0;
            if (threadIdx.x < 2) edgeWord[threadIdx.x] = 0;
            __threadfence();
        }
        __syncthreads();
        if (active && (staged || writable)) {
            const BitTarget out{staged ? stage : words, staged ? (long long)wordBase : 0, !staged, ffCount,
                                staged ? nullptr : edgeWord, (long long)wordBase, (long long)wordEnd - 1};
            const u64 start = pos + (markerInFront ? 16 : 0);   // first bit of the block itself
            if (leadMarker) {
                out.or32(pos, leadMarker << 16);
                atomicAdd(&ffCount[pos >> 15], kChunkMarkerDelta);   // the marker's 0xFF is no data (bit position -> 4 KiB chunk)
            }
            if (bits == 0) {
                // a block with a symbol that has no code: flagged already
            } else if (staged) {
                StageVisitor vis{tab + 256 * t, tab + 512 + 256 * t, stage, (u32)(start - (wordBase << 5))};
                walk_block(k.w, k.pred, vis);
            } else {
                EmitVisitor vis{tab + 256 * t, tab + 512 + 256 * t, out, (long long)(start >> 5), 0, (int)(start & 31), true};
                walk_block(k.w, k.pred, vis);
                vis.finish();
            }
            if (padBits != 0 && g.padOnes) out.or32(start + bits, ((1u << padBits) - 1u) << (32 - padBits));
            if (tailEoi) {
                out.or32(start + bits + padBits, 0xffd9u << 16);
                atomicAdd(&ffCount[(start + bits + padBits) >> 15], kChunkMarkerDelta);
            }
        }
        __syncthreads();
        u64* myEdges = dense_edges(state, tiles) + 2ull * tile;
        if (!staged) {
            if (threadIdx.x == 0) {
                myEdges[0] = writable ? (wordBase << 32) | edgeWord[0] : kNoEdge;
                myEdges[1] = writable ? ((wordEnd - 1) << 32) | edgeWord[1] : kNoEdge;
            }
        } else if (writable) {
            dense_copy_out(stage, (u32)nWords, 0u, wordBase, words, ffCount, myEdges);
        } else if (threadIdx.x == 0) {
            myEdges[0] = myEdges[1] = kNoEdge;
            atomicOr((unsigned long long*)&result[1], 4ull);
        }
    }
}
