// Entropy stage on groups of eight coefficients -- included by entropy.hip inside namespace jpeghip.
//
// A block's 64 zig-zag ordered coefficients are eight groups of eight (16 bytes each).  After
// quantization most groups of a photographic image are all zero (three of eight survive at
// q=90), and an all-zero group codes nothing: its zeros only lengthen the run in front of
// the next non-zero coefficient.  The kernels here therefore work on the groups that are
// not empty.  A wave reads rows of 64 groups (8 blocks x 8 groups, lane l: block l / 8,
// group l % 8 -- one fully coalesced 1 KiB read), keeps the non-empty ones in scan order in a
// small ring in LDS (ballot + lane count give every group its slot), and whenever 64 of them
// have come together one lane takes one group: eight coefficients, no loop, no divergence.
// Zero runs come from bit masks (the eight non-zero flags of the group, one sentinel bit for
// the zeros that earlier groups of the block left behind) and v_ffbh, categories from the
// exponent of the value converted to float (Encoder::lookupBitCategory, Encoder.cpp:354-366),
// DC differences and AC symbols as Encoder::calculateDCValues / calculateACValues define them
// (Encoder.cpp:260-352).
//
// Every wave of a workgroup works on a tile of its own; the waves share the tables only.

constexpr int kGrpQueue = 256;                   // ring capacity in groups (a power of two)
constexpr int kGrpStageWords = 896;              // stage of one wave's tile, words (four workgroups fit a CU's LDS)
constexpr int kGrpStageBits = kGrpStageWords * 32;
constexpr int kGrpMaxFix = 24;                   // interval ends a tile can hold (intervals are >= 16 blocks)
constexpr int kGrpMinIntervalBlocks = 16;

// Symbol tables in LDS, one word per (table, run, category): rows of 16 categories, 23 rows
// per AC table (a run of 16..22 in front of a group's first coefficient aliases run & 15),
// then the DC rows, the end-of-block and the zero-run-length entries.
constexpr int kTabAcWords = 23 * 16;
constexpr int kTabDc = 2 * kTabAcWords;
constexpr int kTabEob = kTabDc + 32;
constexpr int kTabZrl = kTabEob + 2;
constexpr int kTabWords = kTabZrl + 2;
// A token: the symbol's code shifted left by the category, so that the amplitude bits can be
// ORed in, in bits 26..0 and the total number of bits (code + amplitude, at most 27) in bits
// 31..27.  Symbols of category 0 inside the AC rows stand for zero coefficients: empty.
constexpr u32 kTokAbsent = 0xF8000000u;          // the table has no code for a symbol (31 is no length)

// Queue tags: [15:0] DC predictor (group 0 only), [18:16] group, [26:19] block inside the
// tile, [27] chroma, [28] not a group but the end of a restart interval.
constexpr u32 kTagPseudo = 1u << 28, kTagChroma = 1u << 27;
constexpr u32 kTagKeyMask = (0xffu << 19) | kTagPseudo;

__device__ __forceinline__ u32 lanes_below(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

__device__ __forceinline__ u32 dpp_or_zero(u32 v, int ctrl, int rowMask) {
    // lanes without a source lane (or outside rowMask) get 0
    switch (ctrl) {
        case 0x111: return __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);
        case 0x112: return __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);
        case 0x114: return __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);
        case 0x118: return __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);
        case 0x142: return __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    }
}

// Inclusive sum over the 64 lanes of a wave (row shifts, then the two row broadcasts).
__device__ __forceinline__ u32 wave_scan_add(u32 v) {
    v += dpp_or_zero(v, 0x111, 0xf);
    v += dpp_or_zero(v, 0x112, 0xf);
    v += dpp_or_zero(v, 0x114, 0xf);
    v += dpp_or_zero(v, 0x118, 0xf);
    v += dpp_or_zero(v, 0x142, 0xa);
    v += dpp_or_zero(v, 0x143, 0xc);
    return v;
}

// Per block of a tile: bit 0 chroma, bit 1 the DC predictor restarts at 0, bits 6..2 distance
// to the previous block of the channel, bit 7 last block of a restart interval.
__device__ __forceinline__ u32 group_block_role(const KGeom& g, u32 bl) {
    const BlockRole r = block_role(g, bl);
    bool last = bl == (u32)g.blocksPerImage - 1;
    if (g.intervals != g.batch && r.k == g.bpm - 1 && (r.mcu + 1) % (u32)g.mcusPerInterval == 0) last = true;
    return (r.c != 0 ? 1u : 0u) | (r.resets ? 2u : 0u) | ((u32)r.back << 2) | (last ? 0x80u : 0u);
}

// One wave's view of its tile and its ring.
struct GroupTile {
    const int16_t* image;    // coefficients of the tile's image
    u32 first;               // first block of the tile inside its image
    int nBlocks;
    uint4* qData;
    u32* qTag;
    unsigned char* role;
};

struct GroupRows {
    uint4 d[2];
    int pred[2];
    u32 role[2];
};

// Reads rows 2 * it and 2 * it + 1 of the tile (16 blocks): every lane one group per row, group 0
// lanes also the DC term their block is predicted from.
__device__ __forceinline__ void group_load_rows(const GroupTile& t, int it, GroupRows& rows) {
    const int lane = threadIdx.x & 63, k = lane & 7;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lb = it * 16 + r * 8 + (lane >> 3);
        const bool valid = lb < t.nBlocks;
        rows.d[r] = make_uint4(0, 0, 0, 0);
        rows.pred[r] = 0;
        rows.role[r] = 0;
        if (valid) {
            const u32 role = t.role[lb];
            rows.role[r] = role;
            const long long b = (long long)t.first + lb;
            rows.d[r] = *(const uint4*)(t.image + b * 64 + k * 8);
            if (k == 0) rows.pred[r] = t.image[(b - ((role & 2u) ? 0 : (long long)((role >> 2) & 31u))) * 64];
        }
    }
}

// Appends the non-empty groups of two rows to the ring (in scan order) and, behind the last
// group of a block that ends a restart interval, an entry for the interval's end.  Returns the
// new tail.
template <bool kIntervals>
__device__ __forceinline__ u32 group_enqueue(const GroupTile& t, int it, const GroupRows& rows, u32 tail) {
    const int lane = threadIdx.x & 63, k = lane & 7;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int lb = it * 16 + r * 8 + (lane >> 3);
        const bool valid = lb < t.nBlocks;
        const uint4 d = rows.d[r];
        const bool nonempty = valid && (k == 0 || (d.x | d.y | d.z | d.w) != 0);
        const u64 mN = __ballot(nonempty);
        u32 pos = tail + lanes_below(mN);
        u32 added = (u32)__popcll(mN);
        if (kIntervals) {
            const bool bnd = valid && k == 7 && (rows.role[r] & 0x80u) != 0;
            const u64 mB = __ballot(bnd);
            if (mB != 0) {
                pos += lanes_below(mB);
                added += (u32)__popcll(mB);
                if (bnd) t.qTag[(pos + (nonempty ? 1u : 0u)) & (kGrpQueue - 1)] = kTagPseudo | ((u32)lb << 19) | (7u << 16);
            }
        }
        if (nonempty) {
            const u32 slot = pos & (kGrpQueue - 1);
            const u32 pred = (rows.role[r] & 2u) ? 0u : ((u32)rows.pred[r] & 0xffffu);
            t.qData[slot] = d;
            t.qTag[slot] = (k == 0 ? pred : 0u) | ((u32)k << 16) | ((u32)lb << 19) | ((rows.role[r] & 1u) ? kTagChroma : 0u);
        }
        tail += added;
    }
    return tail;
}

// What a lane knows about its group before it looks at single coefficients.
struct GroupFront {
    u32 d[4];        // the eight coefficients (group 0: the DC term replaced by its difference)
    u32 tag;
    u32 mm;          // bits 7..0: coefficient j is not zero (group 0: bit 0 always set)
    u32 X;           // the same flags in bits 23..16 and a sentinel bit below them, see group_front
    u32 val;         // (block key << 6) | index of the group's last non-zero coefficient
    int k;
    int zr;          // ZRL symbols in front of the group's first non-zero coefficient
    bool real;       // a group (not an interval end, not an idle lane)
    bool isK0;
    bool needEob;
    u32 tabBase;     // byte offset of the group's AC rows in the table
};

// carryVal: `val` of the item in front of the chunk (wave uniform).
__device__ __forceinline__ void group_front(const GroupTile& t, u32 head, int n, u32 carryVal, GroupFront& f) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const bool act = lane < n;
    const u32 slot = (head + (u32)lane) & (kGrpQueue - 1);
    const u32 tag = act ? t.qTag[slot] : kTagPseudo;
    const u32 next = t.qTag[(slot + 1) & (kGrpQueue - 1)];
    const bool pseudo = (tag & kTagPseudo) != 0;
    f.real = !pseudo;
    f.tag = tag;
    uint4 q = t.qData[slot];
    if (pseudo) q = make_uint4(0, 0, 0, 0);
    f.k = (int)((tag >> 16) & 7u);
    f.isK0 = f.real && f.k == 0;
    if (f.isK0) {   // Encoder.cpp:268-272: the DC term is coded as the difference to its predecessor
        const int diff = (int)(short)(q.x & 0xffffu) - (int)(short)(tag & 0xffffu);
        q.x = (q.x & 0xffff0000u) | ((u32)diff & 0xffffu);
    }
    f.d[0] = q.x; f.d[1] = q.y; f.d[2] = q.z; f.d[3] = q.w;
    // Non-zero flags: min(value, 1) on both halves of a dword gives bit 0 and bit 16; the four
    // dwords interleave to even bits 0..6 (even coefficients) and 16..22 (odd ones).
    u32 a = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const us2 one = {1, 1};
        const us2 nz = __builtin_elementwise_min(__builtin_bit_cast(us2, f.d[i]), one);
        a |= __builtin_bit_cast(u32, nz) << (2 * i);
    }
    u32 mm = a | (a >> 15);           // bits 7..0 in coefficient order (bits 16..22 are left-overs)
    if (f.isK0) mm |= 1u;             // runs are counted from the DC term whatever its difference is
    f.mm = mm;
    const int tz = (int)__builtin_clz((mm << 24) | 0x00800000u);   // zero coefficients at the group's end, 0..8
    const u32 key = (tag >> 19) & 0x3ffu;
    f.val = (key << 6) | (u32)(f.k * 8 + 7 - tz);
    const u32 prev = __builtin_amdgcn_update_dpp(carryVal, f.val, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    // Zero coefficients between the block's last non-zero one and this group (0 for group 0).
    const int carry = (f.real && (prev >> 6) == key) ? f.k * 8 - 1 - (int)(prev & 63u) : 0;
    // clz(X << (16 - j)) is the zero run in front of coefficient j: flags above, the sentinel
    // (carry & 15) + 1 places below the group's first flag.
    f.X = (mm << 16) | (1u << (15 - (carry & 15)));
    const int firstNz = (int)__builtin_ctz(mm | 0x100u);
    f.zr = f.real ? (firstNz + carry) >> 4 : 0;
    const bool lastOfBlock = ((tag ^ next) & kTagKeyMask) != 0;
    f.needEob = f.real && lastOfBlock && !(f.k == 7 && (mm & 0x80u));   // Encoder.cpp:343-347
    f.tabBase = (tag & kTagChroma) ? kTabAcWords * 4u : 0u;
}

// Byte offset of coefficient j's symbol in the table: AC row `run`, column `category`; the DC
// term goes to the DC rows.
__device__ __forceinline__ u32 group_symbol_offset(const GroupFront& f, int j, int& cat) {
    const int v = (int)(short)(f.d[j >> 1] >> (16 * (j & 1)));
    cat = __builtin_amdgcn_frexp_expf((float)v);   // bits of |v|, 0 for 0
    const u32 run = (u32)__builtin_clz(f.X << (16 - j));
    u32 off = f.tabBase + (run << 6) + ((u32)cat << 2);
    if (j == 0 && f.isK0) off = kTabDc * 4u + ((f.tag & kTagChroma) ? 64u : 0u) + ((u32)cat << 2);
    return off;
}

// ---- histograms ---------------------------------------------------------------------------
// Counters in LDS laid out like the symbol table; the bins of category 0 inside the AC rows
// collect the zero coefficients and are dropped.
__global__ __launch_bounds__(256) void k_histogram_groups(KGeom g, const int16_t* __restrict__ coef,
                                                          u64* __restrict__ hist, HistogramExport ex) {
    __shared__ u32 lh[kTabWords];
    __shared__ __attribute__((aligned(16))) uint4 qData[4][kGrpQueue];
    __shared__ u32 qTag[4][kGrpQueue];
    __shared__ unsigned char roleTab[4][256];
    __shared__ bool lastOne;
    for (int i = threadIdx.x; i < kTabWords; i += 256) lh[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GroupTile t;
    t.image = coef + (long long)blockIdx.y * g.blocksPerImage * 64;
    t.qData = qData[wave];
    t.qTag = qTag[wave];
    t.role = roleTab[wave];
    const u32 blocks = (u32)g.blocksPerImage;
    const u32 tiles = (blocks + 255) / 256;
    for (u32 tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
        t.first = tile * 256;
        t.nBlocks = (int)(blocks - t.first < 256u ? blocks - t.first : 256u);
        for (int lb = lane; lb < t.nBlocks; lb += 64) t.role[lb] = (unsigned char)group_block_role(g, t.first + lb);
        u32 head = 0, tail = 0, carryVal = ~0u;
        const int iters = (t.nBlocks + 15) / 16;
        GroupRows cur, nxt;
        group_load_rows(t, 0, cur);
        for (int it = 0; it <= iters; ++it) {
            if (it < iters) {
                if (it + 1 < iters) group_load_rows(t, it + 1, nxt);
                tail = __builtin_amdgcn_readfirstlane(group_enqueue<false>(t, it, cur, tail));
                if (lane == 0) t.qTag[tail & (kGrpQueue - 1)] = ~0u;   // nothing follows the last item yet
                cur = nxt;
            }
            // A full chunk needs the tag of the item behind it (end-of-block test); the last
            // chunks take what is left.
            while (tail - head >= 65 || (it == iters && tail != head)) {
                const int n = tail - head < 64 ? (int)(tail - head) : 64;
                GroupFront f;
                group_front(t, head, n, carryVal, f);
                carryVal = __builtin_amdgcn_readlane(f.val, n - 1);
                head += (u32)n;
                if (f.real) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        int cat;
                        const u32 off = group_symbol_offset(f, j, cat);
                        atomicAdd(&lh[off >> 2], 1u);
                    }
                    const u32 c = (f.tag & kTagChroma) ? 1u : 0u;
                    if (f.needEob) atomicAdd(&lh[kTabEob + c], 1u);
                    if (f.zr != 0) atomicAdd(&lh[kTabZrl + c], (u32)f.zr);
                }
            }
        }
    }
    __syncthreads();
    // Counter i of the LDS layout -> bin of the histogram set (DC-Y, DC-C, AC-Y, AC-C), or none.
    auto bin_of = [](int i) -> int {
        if (i < kTabDc) {
            const int c = i / kTabAcWords, rem = i - c * kTabAcWords, cat = rem & 15;
            return cat == 0 ? -1 : (2 + c) * 256 + ((((rem >> 4) & 15) << 4) | cat);
        }
        if (i < kTabEob) return ((i - kTabDc) >> 4) * 256 + ((i - kTabDc) & 15);
        if (i < kTabZrl) return (2 + (i - kTabEob)) * 256;
        return (2 + (i - kTabZrl)) * 256 + 0xF0;
    };
    u64* mine = hist + (long long)blockIdx.y * 1024;
    // The additions return their old values so that they have been performed when the
    // workgroup counts itself off (see k_histogram).
    u64 seen = 0;
    for (int i = threadIdx.x; i < kTabWords; i += 256) {
        const u32 v = lh[i];
        const int bin = bin_of(i);
        if (v != 0 && bin >= 0) {
            if (ex.host != nullptr) seen |= atomicAdd(&mine[bin], (u64)v);
            else atomicAdd(&mine[bin], (u64)v);
        }
    }
    if (ex.host == nullptr) return;
    if (seen == ~0ull) lh[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) lastOne = atomicAdd(ex.counter, 1ull) + 1 == ex.expected;
    __syncthreads();
    if (!lastOne) return;
    for (int i = threadIdx.x; i < 1024; i += 256)
        ex.host[i] = __hip_atomic_load(&hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- bit writer -----------------------------------------------------------------------------
// ORs the n low bits of `bits` into the stage so that they start at bit position pos
// (position 0 is bit 31 of word 0: the stage holds the stream MSB first in native words).
__device__ __forceinline__ void stage_put(u32* st, u32 pos, u32 bits, u32 n) {
    const u32 sh = pos & 31u;
    const u64 v = (u64)bits << ((64u - sh - n) & 63u);   // bits == 0 whenever n == 0
    u32* w = st + (pos >> 5);
    atomicOr(w, (u32)(v >> 32));
    atomicOr(w + 1, (u32)v);
}

// Copies `bits` bits of the stage, starting at word srcWord, to the stream at bit position
// dst, and clears them in the stage.  The stream is zero where nothing has been written yet;
// the first and the last word of the range may be shared with neighbours and are merged.
__device__ __forceinline__ void stage_copy_out(u32* st, u32 srcWord, u32 bits, u64 dst, u32* __restrict__ words) {
    if (bits == 0) return;
    const int lane = threadIdx.x & 63;
    const u32 s = (u32)(dst & 31);
    const u64 w0 = dst >> 5;
    const u32 nSrc = (bits + 31) >> 5, nOut = (s + bits + 31) >> 5;
    for (u32 i = lane; i < nOut; i += 64) {
        const u32 hi = i > 0 ? st[srcWord + i - 1] : 0u;
        const u32 lo = i < nSrc ? st[srcWord + i] : 0u;
        const u32 v = __builtin_bswap32(__builtin_amdgcn_alignbit(hi, lo, s));
        if (i == 0 || i == nOut - 1) {
            if (v != 0) atomicOr(&words[w0 + i], v);
        } else {
            words[w0 + i] = v;
        }
    }
    for (u32 i = lane; i < nSrc; i += 64) st[srcWord + i] = 0;
}

__global__ __launch_bounds__(256) void k_encode_groups(KGeom g, const int16_t* __restrict__ coef,
                                                       const u32* __restrict__ tables, u64* __restrict__ state,
                                                       u64* __restrict__ ivOff, u64* __restrict__ result,
                                                       u32* __restrict__ words, int tileBlocks, long long zeroedBytes) {
    __shared__ u32 tab[kTabWords];
    __shared__ __attribute__((aligned(16))) uint4 qData[4][kGrpQueue];
    __shared__ u32 qTag[4][kGrpQueue];
    __shared__ u32 stage[4][kGrpStageWords];
    __shared__ unsigned char roleTab[4][256];
    __shared__ u32 fixBlock[4][kGrpMaxFix], fixRel[4][kGrpMaxFix];
    __shared__ unsigned long long ticket;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Tiles per image in fours, so that the four tiles of a workgroup share their image's tables.
    const u32 tilesPerImage = (((u32)g.blocksPerImage + (u32)tileBlocks - 1) / (u32)tileBlocks + 3u) & ~3u;
    const long long tiles = (long long)tilesPerImage * g.batch;
    // Tile numbers come from a counter: every predecessor of a running tile is running or done.
    if (threadIdx.x == 0) ticket = atomicAdd((unsigned long long*)(state + tiles), 4ull);
    for (int i = threadIdx.x; i < 4 * kGrpStageWords; i += 256) (&stage[0][0])[i] = 0;
    __syncthreads();
    const u32 tile = (u32)__builtin_amdgcn_readfirstlane((u32)ticket) + (u32)wave;
    const u32 image = tile / tilesPerImage;
    {   // token templates from the image's tables (entries: length << 16 | code)
        const u32* src = tables + (long long)image * g.tableStride;
        for (int i = threadIdx.x; i < kTabWords; i += 256) {
            u32 e, cat = 0;
            bool zeroCoef = false;
            if (i < kTabDc) {
                const int c = i / kTabAcWords, rem = i - c * kTabAcWords;
                cat = (u32)rem & 15u;
                zeroCoef = cat == 0;
                e = src[(2 + c) * 256 + ((((rem >> 4) & 15) << 4) | (int)cat)];
            } else if (i < kTabEob) {
                cat = (u32)(i - kTabDc) & 15u;
                e = src[((i - kTabDc) >> 4) * 256 + (int)cat];
            } else {
                e = src[(2 + ((i - kTabEob) & 1)) * 256 + (i < kTabZrl ? 0 : 0xF0)];
            }
            const u32 len = e >> 16, code = e & 0xffffu;
            tab[i] = zeroCoef ? 0u : ((len == 0 || len + cat > 27u) ? kTokAbsent : ((code << cat) | ((len + cat) << 27)));
        }
    }
    __syncthreads();
    // No workgroup barrier from here on: the waves go their own ways (a wave's look-back may
    // wait for a tile of another wave of the same workgroup).

    GroupTile t;
    t.image = coef + (long long)image * g.blocksPerImage * 64;
    t.qData = qData[wave];
    t.qTag = qTag[wave];
    t.role = roleTab[wave];
    u32* st = stage[wave];
    const u32 tileInImage = tile - image * tilesPerImage;
    const long long firstLL = (long long)tileInImage * tileBlocks;
    t.first = (u32)(firstLL < g.blocksPerImage ? firstLL : g.blocksPerImage);
    {
        const long long left = g.blocksPerImage - (long long)t.first;
        t.nBlocks = (int)(left < tileBlocks ? left : tileBlocks);
    }
    for (int lb = lane; lb < t.nBlocks; lb += 64) t.role[lb] = (unsigned char)group_block_role(g, t.first + lb);

    // Position inside the tile's part of the stream, as a function of the position P at
    // which the tile starts: P + runX while no interval has ended inside the tile, after that
    // roundup8(P + runX) + runY (PosSummary).  Bits in front of the first interval end are
    // staged from bit 0, the rest from a word boundary behind them, and the two pieces go to the
    // stream separately once P is known.
    u32 runX = 0, runY = 0;
    u32 head = 0, tail = 0, carryVal = ~0u, nFix = 0;
    u32 maxTok = 0;
    bool overflow = false;
    const int iters = (t.nBlocks + 15) / 16;
    GroupRows cur, nxt;
    group_load_rows(t, 0, cur);
    for (int it = 0; it <= iters; ++it) {
        if (it < iters) {
            if (it + 1 < iters) group_load_rows(t, it + 1, nxt);
            tail = __builtin_amdgcn_readfirstlane(group_enqueue<true>(t, it, cur, tail));
            if (lane == 0) t.qTag[tail & (kGrpQueue - 1)] = ~0u;
            cur = nxt;
        }
        while (tail - head >= 65 || (it == iters && tail != head)) {
            const int n = tail - head < 64 ? (int)(tail - head) : 64;
            GroupFront f;
            group_front(t, head, n, carryVal, f);
            carryVal = __builtin_amdgcn_readlane(f.val, n - 1);
            head += (u32)n;
            const bool boundary = lane < n && !f.real;

            // Tokens of the eight coefficients, with their bit counts.
            typedef short s2 __attribute__((ext_vector_type(2)));
            u32 amp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // v for v >= 0, v - 1 (the one's complement of |v|) for v < 0, Encoder.cpp:281-284
                const s2 v = __builtin_bit_cast(s2, f.d[i]);
                amp[i] = __builtin_bit_cast(u32, (s2)(v + (v >> 15)));
            }
            u32 tok[8], off[9];
            const u32 c = (f.tag & kTagChroma) ? 1u : 0u;
            const u32 zrlTok = tab[kTabZrl + c], eobTok = tab[kTabEob + c];
            off[0] = (u32)f.zr * (zrlTok >> 27);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int cat;
                const u32 o = group_symbol_offset(f, j, cat);
                const u32 e = *(const u32*)((const char*)tab + o);
                tok[j] = e | __builtin_amdgcn_ubfe(amp[j >> 1], 16 * (j & 1), (u32)cat);
                maxTok = tok[j] > maxTok ? tok[j] : maxTok;
                off[j + 1] = off[j] + (tok[j] >> 27);
            }
            if (f.zr != 0 && zrlTok > maxTok) maxTok = zrlTok;
            if (f.needEob && eobTok > maxTok) maxTok = eobTok;
            const u32 eobBits = f.needEob ? eobTok >> 27 : 0u;
            const u32 len = f.real ? off[8] + eobBits : 0u;

            // Where the lane's bits go in the stage.
            u32 base;
            if (__ballot(boundary) == 0) {
                const u32 incl = wave_scan_add(len);
                const u32 total = __builtin_amdgcn_readlane(incl, 63);
                if ((runX >> 31) == 0) {
                    base = runX + incl - len;
                    runX += total;
                } else {
                    const u32 h = runX & 0x7fffffffu;
                    base = ((h + 31u) & ~31u) + 32u + runY + incl - len;
                    runY += total;
                }
            } else {
                // An interval ends inside the chunk: the general form of the scan.  The end pads
                // the stream to a byte (BitBuffer::fillToByteBorder, BitBuffer.cpp:12-16) and
                // is followed by EOI (the image ends) or the next interval's RSTn.
                const u32 bl = t.first + ((f.tag >> 19) & 0xffu);
                const bool lastBlock = bl == (u32)g.blocksPerImage - 1;
                const u32 eoiBits = boundary && g.writeEoi && (g.batch > 1 || lastBlock) ? 16u : 0u;
                const u32 rstBits = boundary && g.batch == 1 && !lastBlock ? 16u : 0u;
                PosSummary mine = {len, 0u};
                if (boundary) {
                    mine.x = 0x80000000u;
                    mine.y = eoiBits + rstBits;
                }
                PosSummary inc = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    PosSummary o;
                    o.x = __shfl_up(inc.x, d, 64);
                    o.y = __shfl_up(inc.y, d, 64);
                    if (lane >= d) inc = pos_combine(o, inc);
                }
                PosSummary excl;
                excl.x = __shfl_up(inc.x, 1, 64);
                excl.y = __shfl_up(inc.y, 1, 64);
                if (lane == 0) excl.x = excl.y = 0;
                const PosSummary run = {runX, runY};
                const PosSummary at = pos_combine(run, excl);
                const u32 h = at.x & 0x7fffffffu;
                const u32 partB = ((h + 31u) & ~31u) + 32u;
                base = (at.x >> 31) ? partB + at.y : at.x;
                if (boundary) {
                    // Offset, inside the part behind the tile's first interval end, at which this
                    // end's padding stops.  The padding of the first end depends on P and is
                    // left to the copy.
                    u32 padded = 0;
                    if (at.x >> 31) {
                        padded = (at.y + 7u) & ~7u;
                        if (g.padOnes && padded != at.y && partB + padded + 64u <= (u32)kGrpStageBits)
                            stage_put(st, partB + at.y, (1u << (padded - at.y)) - 1u, padded - at.y);
                    }
                    const u32 iv = g.batch > 1 ? image : (bl / (u32)g.bpm) / (u32)g.mcusPerInterval;
                    u32 marks = 0, nMarks = eoiBits + rstBits;
                    if (eoiBits) marks = 0xffd9u;
                    if (rstBits) marks = (marks << 16) | (0xffd0u + (u32)((g.firstInterval + iv) & 7));
                    if (partB + padded + nMarks + 64u <= (u32)kGrpStageBits) {
                        if (nMarks == 32u) {
                            stage_put(st, partB + padded, marks >> 16, 16u);
                            stage_put(st, partB + padded + 16u, marks & 0xffffu, 16u);
                        } else if (nMarks != 0) {
                            stage_put(st, partB + padded, marks, nMarks);
                        }
                    } else {
                        overflow = true;
                    }
                    const u32 slot = nFix + lanes_below(__ballot(boundary));
                    if (slot < (u32)kGrpMaxFix) {
                        fixBlock[wave][slot] = bl;
                        fixRel[wave][slot] = padded + eoiBits;
                    } else {
                        overflow = true;
                    }
                }
                nFix += (u32)__popcll(__ballot(boundary));
                const PosSummary whole = pos_combine(run, PosSummary{(u32)__builtin_amdgcn_readlane(inc.x, 63),
                                                                     (u32)__builtin_amdgcn_readlane(inc.y, 63)});
                runX = whole.x;
                runY = whole.y;
            }
            runX = __builtin_amdgcn_readfirstlane(runX);
            runY = __builtin_amdgcn_readfirstlane(runY);

            if (f.real) {
                if (base + len + 64u <= (u32)kGrpStageBits) {
                    if (f.zr != 0) {
                        const u32 zl = zrlTok >> 27;
                        for (int z = 0; z < f.zr; ++z) stage_put(st, base + (u32)z * zl, zrlTok & 0x07ffffffu, zl);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) stage_put(st, base + off[j], tok[j] & 0x07ffffffu, tok[j] >> 27);
                    if (f.needEob) stage_put(st, base + off[8], eobTok & 0x07ffffffu, eobBits);
                } else {
                    overflow = true;
                }
            }
        }
    }

    // The tile's summary is complete: publish it, find the tile's start, publish that.
    const PosSummary whole = {runX, runY};
    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    u64 start = p0;
    bool ok = true;
    if (tile > 0) {
        if (lane == 0)
            __hip_atomic_store(state + tile, kTileAggregate | ((u64)whole.x << 30) | whole.y, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        ok = look_back(state, tile, p0, start);
    }
    if (!ok) start = p0;
    if (lane == 0)
        __hip_atomic_store(state + tile, kTilePosition | pos_apply(whole, start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    u32 flags = ok ? 0u : 2u;
    if (__ballot(overflow) != 0) flags |= 8u;
    if (__ballot(maxTok >= kTokAbsent) != 0) flags |= 16u;
    const u32 h = whole.x & 0x7fffffffu;
    stage_copy_out(st, 0, h, start, words);
    if (whole.x >> 31) {
        const u64 partStart = (start + h + 7) & ~7ull;
        stage_copy_out(st, (((h + 31u) & ~31u) + 32u) >> 5, whole.y, partStart, words);
        if (g.padOnes && partStart != start + h && lane == 0) {   // the first end's padding, as one bits
            const u64 at = start + h;
            const u32 nPad = (u32)(partStart - at);
            atomicOr(&words[at >> 5], __builtin_bswap32(((1u << nPad) - 1u) << (32u - (u32)(at & 31) - nPad)));
        }
        if ((u32)lane < nFix && (u32)lane < (u32)kGrpMaxFix) {
            const u32 bl = fixBlock[wave][lane];
            const long long iv = g.batch > 1 ? (long long)image : (long long)((bl / (u32)g.bpm) / (u32)g.mcusPerInterval);
            const u64 bytes = (partStart + fixRel[wave][lane]) >> 3;
            ivOff[iv + 1] = bytes;
            if (iv + 1 == g.intervals) {
                result[2] = bytes;
                result[3] = (bytes + kStuffChunk - 1) / kStuffChunk;
                if ((long long)bytes > zeroedBytes) atomicOr((unsigned long long*)&result[1], 4ull);
            }
        }
    }
    if (tile == 0 && lane == 0) {
        ivOff[0] = 0;
        if (p0 != 0) atomicOr(&words[0], __builtin_bswap32((0xffd0u + (u32)((g.firstInterval - 1) & 7)) << 16));
    }
    if (flags != 0 && lane == 0) atomicOr((unsigned long long*)&result[1], (unsigned long long)flags);
}
