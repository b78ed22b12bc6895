// Entropy stage on groups of eight coefficients -- included by entropy.hip inside namespace jpeghip.
//
// A block's 64 zig-zag ordered coefficients are eight groups of eight (16 bytes each).  After
// quantization most groups of a photographic image are all zero (three of eight survive at
// q=90), and an all-zero group codes nothing: its zeros only lengthen the run in front of
// the next non-zero coefficient.  The transform leaves one byte per block that says which
// groups are not empty; the kernels here read those bytes, list the non-empty groups of a
// tile in scan order (a prefix sum over the bytes' bit counts gives every group its place), and
// then one lane takes one group: a 16-byte read of eight coefficients, no loop over them, no
// divergence, and the empty groups are never read at all.
// (The histogram pass of large images is built the same way, k_histogram_tiles below; small images
// keep k_histogram, whose one-lane-per-block walk has no list to build.)
// Zero runs come from bit masks (the eight non-zero flags of the group, one sentinel bit for
// the zeros that earlier groups of the block left behind) and v_ffbh, categories from the
// exponent of the value converted to float (Encoder::lookupBitCategory, Encoder.cpp:354-366),
// DC differences and AC symbols as Encoder::calculateDCValues / calculateACValues define them
// (Encoder.cpp:260-352).
//
// Every wave of a workgroup works on a tile of its own; the waves share the tables only.

constexpr int kGrpTileBlocks = 256;              // largest tile, blocks (four per lane)
// Room in LDS for a tile's list of items and for its bits.  Both are sized for photographic
// content, not for the worst case (nine items and a few hundred bits per block): six workgroups
// then fit a CU's LDS, and a wave issues an instruction only every eighth cycle or so, so the
// kernel's speed follows the number of resident waves.  A tile that needs more is reported
// (flag 8) and the host runs the stage again with smaller tiles; 64 blocks always fit the list.
constexpr int kGrpMaxItems = 1278;
constexpr int kGrpStageWords = 704;              // stage of one wave's tile, words
constexpr int kGrpStageBits = kGrpStageWords * 32;
constexpr int kGrpMaxFix = 24;                   // interval ends a tile can hold (intervals are >= 16 blocks)
constexpr int kGrpMinIntervalBlocks = 16;
constexpr long long kGrpMinBlocks = 150000;       // images (or batches) below this many blocks keep the one-lane-per-block writer (entropy stage at 98 k blocks: 39 against 42 us; at 197 k: 49 against 45)
constexpr u32 kGrpBaseBlocks = 32;               // the tile's base address lies this many blocks in front of its first block

// Token templates in LDS: see kTabPitch in kernels.h.
// An item of a tile's list: (block inside the tile << 4) | group; group 8 stands for the end
// of a restart interval behind the block's last group.
constexpr u32 kItemNone = 0xffffu;

// min(a, b) on both 16-bit halves (the compiler turns the vector form into compares and selects).
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
    u32 r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ u32 dpp_or_zero(u32 v, int ctrl) {
    // lanes without a source lane (or outside the row mask) get 0
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
    v += dpp_or_zero(v, 0x111);
    v += dpp_or_zero(v, 0x112);
    v += dpp_or_zero(v, 0x114);
    v += dpp_or_zero(v, 0x118);
    v += dpp_or_zero(v, 0x142);
    v += dpp_or_zero(v, 0x143);
    return v;
}

// One wave's view of its tile.
struct GroupTile {
    const char* base;        // address of the tile's first block minus kGrpBaseBlocks blocks: predictors lie in front
    u32 shift;               // log2 of the bytes from one block's groups 0..3 to the next block's: 7 or 6 (KGeom::coefPitch)
    u32 high;                // bytes from a block's groups 0..3 to its groups 4..7 (KGeom::coefHigh)
    const unsigned char* mask;   // group masks of the tile's blocks
    u32 first;               // first block of the tile inside its image
    int nBlocks;
    unsigned short* items;   // [cap + 2]
    int cap;                 // items the list has room for
    unsigned char* role;     // [kGrpTileBlocks]
    int nItems;
    bool tooMany;            // the tile has more items than the list holds
};

// Lists the tile's non-empty groups in scan order and notes every block's role: bit 0 chroma,
// bit 1 the DC predictor restarts at 0, bits 6..2 distance to the previous block of the
// channel, bit 7 last block of a restart interval (kIntervals only; such a block gets an
// item of its own behind its groups).  Lane l looks after blocks 4l..4l+3.
// kSkipFull: a tile all of whose blocks have all eight groups gets no list (t.nItems says so: eight per block).
template <bool kIntervals, bool kSkipFull = false>
__device__ __forceinline__ void group_index_tile(const KGeom& g, GroupTile& t) {
    const int lane = threadIdx.x & 63;
    const int lb0 = 4 * lane;
    // Position of the lane's first block: two divisions, the other three blocks follow by counting.
    const u32 bpm = (u32)g.bpm;
    const u32 bl0 = t.first + (u32)lb0;
    u32 mcu = bl0 / bpm;
    int k = (int)(bl0 - mcu * bpm);
    const bool oneInterval = g.intervals == g.batch;   // no restarts, or a batch: the image is the interval
    u32 inInterval = oneInterval ? mcu : mcu % (u32)g.mcusPerInterval;
    const u32 lastInInterval = oneInterval ? ~0u : (u32)g.mcusPerInterval - 1;
    u32 counts = 0, total = 0;
    u32 masks[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lb = lb0 + i;
        const bool valid = lb < t.nBlocks;
        const int c = k < g.off1 ? 0 : (k < g.off2 ? 1 : 2);
        const int first = c == 0 ? 0 : (c == 1 ? g.off1 : g.off2);
        const int end = c == 0 ? g.off1 : (c == 1 ? g.off2 : g.bpm);
        const u32 back = k > first ? 1u : (u32)(g.bpm - (end - 1 - first));
        const bool last = kIntervals && (t.first + (u32)lb == (u32)g.blocksPerImage - 1 ||
                                         (inInterval == lastInInterval && k == g.bpm - 1));
        const u32 role = (c != 0 ? 1u : 0u) | (k == first && inInterval == 0 ? 2u : 0u) | (back << 2) | (last ? 0x80u : 0u);
        if (valid) t.role[lb] = (unsigned char)role;
        const u32 m = valid ? ((u32)t.mask[lb] | 1u | (last ? 0x100u : 0u)) : 0u;   // group 0 always: it carries the DC term
        masks[i] = m;
        counts |= total << (8 * i);   // items of the lane's earlier blocks (at most 27)
        total += (u32)__popc(m);
        if (++k == (int)bpm) {
            k = 0;
            ++mcu;
            if (oneInterval) inInterval = mcu;
            else if (++inInterval == (u32)g.mcusPerInterval) inInterval = 0;
        }
    }
    const u32 incl = wave_scan_add(total);
    const u32 base = incl - total;
    t.nItems = (int)__builtin_amdgcn_readlane(incl, 63);
    t.tooMany = t.nItems > t.cap;
    if (t.tooMany) t.nItems = t.cap;   // the list stops where its room ends; the tile's output is void
    if (kSkipFull && t.nItems == 8 * t.nBlocks) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 slot = base + ((counts >> (8 * i)) & 0xffu);
        const u32 m = masks[i];
        const u32 item = (u32)(lb0 + i) << 4;
#pragma unroll
        for (int grp = 0; grp < (kIntervals ? 9 : 8); ++grp) {
            if (((m >> grp) & 1u) && slot < (u32)t.cap) t.items[slot] = (unsigned short)(item | (u32)grp);
            slot += (m >> grp) & 1u;
        }
    }
    if (lane == 0) {   // nothing follows the last item
        t.items[t.nItems] = (unsigned short)kItemNone;
        t.items[t.nItems + 1] = (unsigned short)kItemNone;
    }
}

// The loads one lane issues for its item: the group's 16 bytes, and the DC term the block is
// predicted from (used by group 0 only).
struct GroupLoad {
    uint4 d;
    u32 pred;
};

template <bool kStream = false>
__device__ __forceinline__ void group_issue_loads(const GroupTile& t, int firstItem, GroupLoad& ld) {
    const int lane = threadIdx.x & 63;
    int i = firstItem + lane;
    i = i < t.nItems ? i : t.nItems - 1;    // idle lanes read what the last item reads
    const u32 item = t.items[i];
    const u32 lb = item >> 4, grp = item & 7u;
    const u32 role = t.role[lb];
    const u32 back = (role & 2u) ? 0u : ((role >> 2) & 31u);
    const char* at = t.base + (((lb + kGrpBaseBlocks) << t.shift) + (grp & 3u) * 16u + (grp >= 4u ? t.high : 0u));
    if (kStream) {   // read once and not again soon: leave the caches to what will be
        typedef u32 u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load((const u32x4*)at);
        ld.d = make_uint4(v.x, v.y, v.z, v.w);
    } else {
        ld.d = *(const uint4*)at;
    }
    ld.pred = *(const unsigned short*)(t.base + ((lb + kGrpBaseBlocks - back) << t.shift));
}

// What a lane knows about its group before it looks at single coefficients.
struct GroupFront {
    u32 d[4];        // the eight coefficients (group 0: the DC term replaced by its difference)
    u32 lb;          // block inside the tile
    u32 mm;          // bits 7..0: coefficient j is not zero (group 0: bit 0 always set)
    u32 X;           // the same flags in bits 23..16 and a sentinel bit below them, see group_front
    u32 val;         // (block << 6) | index of the group's last non-zero coefficient
    int zr;          // ZRL symbols in front of the group's first non-zero coefficient
    bool real;       // a group (not an interval end, not an idle lane)
    bool boundary;   // an interval end
    bool isK0;
    bool needEob;
    u32 chroma;      // 0 / 1
    u32 tabBase;     // byte offset of the group's AC rows in the table
};

// carryVal: `val` of the item in front of the chunk (wave uniform).
__device__ __forceinline__ void group_front(const GroupTile& t, int firstItem, const GroupLoad& ld, u32 carryVal,
                                            GroupFront& f) {
    const int lane = threadIdx.x & 63;
    const int i = firstItem + lane;
    const bool act = i < t.nItems;
    const int ic = act ? i : t.nItems - 1;
    const u32 item = t.items[ic];
    const u32 next = t.items[ic + 1];
    const u32 grp = item & 15u;
    f.lb = item >> 4;
    f.boundary = act && grp == 8u;
    f.real = act && grp != 8u;
    const u32 role = t.role[f.lb];
    f.chroma = role & 1u;
    f.isK0 = f.real && grp == 0u;
    uint4 q = ld.d;
    q.x = f.real ? q.x : 0u;
    q.y = f.real ? q.y : 0u;
    q.z = f.real ? q.z : 0u;
    q.w = f.real ? q.w : 0u;
    if (f.isK0) {   // Encoder.cpp:268-272: the DC term is coded as the difference to its predecessor (0 where a new interval starts)
        const int pred = (role & 2u) ? 0 : (int)(short)ld.pred;
        const int diff = (int)(short)(q.x & 0xffffu) - pred;
        q.x = (q.x & 0xffff0000u) | ((u32)diff & 0xffffu);
    }
    f.d[0] = q.x; f.d[1] = q.y; f.d[2] = q.z; f.d[3] = q.w;
    // Non-zero flags: min(value, 1) on both halves of a dword gives bit 0 and bit 16; the four
    // dwords interleave to even bits 0..6 (even coefficients) and 16..22 (odd ones).
    u32 a = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) a |= pk_min_u16(f.d[w], 0x00010001u) << (2 * w);
    u32 mm = a | (a >> 15);           // bits 7..0 in coefficient order (bits 16..22 are left-overs)
    if (f.isK0) mm |= 1u;             // runs are counted from the DC term whatever its difference is
    f.mm = mm;
    const int tz = (int)__builtin_clz((mm << 24) | 0x00800000u);   // zero coefficients at the group's end, 0..8
    f.val = (f.lb << 6) | (u32)((int)(grp & 7u) * 8 + 7 - tz);
    const u32 prev = __builtin_amdgcn_update_dpp(carryVal, f.val, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    // Zero coefficients between the block's last non-zero one and this group (0 for group 0:
    // the item in front of it belongs to another block).
    const int carry = (f.real && (prev >> 6) == f.lb) ? (int)grp * 8 - 1 - (int)(prev & 63u) : 0;
    // clz(X << (16 - j)) is the zero run in front of coefficient j: flags above, the sentinel
    // (carry & 15) + 1 places below the group's first flag.
    f.X = (mm << 16) | (1u << (15 - (carry & 15)));
    const int firstNz = (int)__builtin_ctz(mm | 0x100u);
    f.zr = f.real ? (firstNz + carry) >> 4 : 0;
    // Last group of its block: the next item belongs to another block, or is the interval's end.
    const bool lastOfBlock = (next >> 4) != f.lb || (next & 15u) == 8u;
    f.needEob = f.real && lastOfBlock && !(grp == 7u && (mm & 0x80u));   // Encoder.cpp:343-347
    f.tabBase = f.chroma * (kTabAcWords * 4u);
}

// Byte offset of coefficient j's symbol in the table: AC row `run`, column `category`; the DC
// term goes to the DC rows.
__device__ __forceinline__ u32 group_symbol_offset(const GroupFront& f, int j, int& cat) {
    const int v = (int)(short)(f.d[j >> 1] >> (16 * (j & 1)));
    cat = __builtin_amdgcn_frexp_expf((float)v);   // bits of |v|, 0 for 0
    const u32 run = (u32)__builtin_clz(f.X << (16 - j));
    u32 off = f.tabBase + run * (kTabPitch * 4u) + ((u32)cat << 2);
    if (j == 0 && f.isK0) off = kTabDc * 4u + f.chroma * 64u + ((u32)cat << 2);
    return off;
}

// The token templates of a table image (kernels.h) into LDS, by the 256 threads of a workgroup; the
// caller passes a barrier afterwards.
__device__ __forceinline__ void load_token_templates(const u32* __restrict__ image, u32* tab) {
    for (int i = threadIdx.x; i < kTabWords; i += 256) tab[i] = image[kTabTemplates + i];
}

// ---- one lane per block, dense content ----------------------------------------------------------
// For blocks in which most coefficients are not zero (noise, q = 100) the block walk with a
// branch per coefficient (walk_block) spends its time on branches all lanes take anyway, and
// walking twice -- once for the length, once for the bits -- does everything twice.  Here a
// lane turns its block's 64 coefficients into 64 tokens (template | amplitude bits, length in
// bits 31..27, 0 for a zero coefficient) without a branch, keeps them in registers, and the
// second pass only puts them where they belong.
//
// The kernel is bound by the number of vector instructions, so the look-up is arranged to need
// few: the templates of a table lie in rows of kTabPitch words per run, 63 rows (a run of 16 and
// more, for which the stream wants ZRL symbols (Encoder.cpp:318-328) that this pass does not write,
// finds kTokAbsent there, like a symbol without a code: the caller takes the general walk for
// the tile then).  The lane carries 63 minus the index of its last non-zero coefficient; coefficient
// i reads row (that + i - 64), the constant part of which sits in the instruction's offset
// field, and a non-zero coefficient replaces the number by a constant.  Eleven instructions
// per coefficient.
constexpr int kDenseRows = 63;
constexpr int kDenseAcWords = kDenseRows * kTabPitch;

struct DenseTables {
    u32 ac[2 * kDenseAcWords];   // luma, chroma: [run][category], column 0 (a zero coefficient) empty
    u32 dc[32];                  // luma, chroma: [category]
    u32 eob[2];
    // Bits a coefficient of a category costs behind a run of zeros, ZRL symbols for runs of sixteen and
    // more included (Encoder.cpp:318-328), laid out like `ac`; kDenseNoCode when a symbol it needs has no code.
    unsigned short acBits[2 * kDenseAcWords];
};
constexpr u32 kDenseNoCode = 0x4000u;   // (as kNoCodeLength: no block has that many bits)

// The templates of a table image (kernels.h) into LDS, by the 256 threads of a workgroup; the caller
// passes a barrier afterwards.
__device__ __forceinline__ void load_dense_tables(const u32* __restrict__ image, DenseTables& t) {
    for (int i = threadIdx.x; i < 2 * kDenseAcWords; i += 256) {
        const int c = i >= kDenseAcWords ? 1 : 0, rem = i - c * kDenseAcWords;
        const int row = rem / kTabPitch, col = rem - row * kTabPitch;
        t.ac[i] = row < 16 ? image[kTabTemplates + c * kTabAcWords + rem] : (col == 0 ? 0u : kTokAbsent);
    }
    if (threadIdx.x < 32) t.dc[threadIdx.x] = image[kTabTemplates + kTabDc + threadIdx.x];
    if (threadIdx.x < 2) t.eob[threadIdx.x] = image[kTabTemplates + kTabEob + threadIdx.x];
    for (int i = threadIdx.x; i < 2 * kDenseAcWords; i += 256) {
        const int c = i >= kDenseAcWords ? 1 : 0, rem = i - c * kDenseAcWords;
        const int row = rem / kTabPitch, col = rem - row * kTabPitch;
        const u32* acTable = image + (2 + c) * 256;   // length << 16 | code, by symbol
        const u32 symLen = acTable[((row & 15) << 4) | (col & 15)] >> 16, zrlLen = acTable[0xF0] >> 16;
        u32 bits = symLen + (u32)col + (u32)(row >> 4) * zrlLen;
        if (symLen == 0 || (row >= 16 && zrlLen == 0)) bits = kDenseNoCode;
        t.acBits[i] = (unsigned short)(col == 0 ? 0u : bits);
    }
}

struct DenseBlock {
    u32 tok[64];    // [0] the DC token, [i] coefficient i's, 0 for zero coefficients
    u32 eob;        // the end-of-block token, 0 when coefficient 63 is not zero (Encoder.cpp:343-347)
    u32 bits;       // length of the block's bit string
    u32 maxTok;     // >= kTokAbsent: a symbol without a code, or a run of sixteen zeros and more
};

// Coefficients kFirst .. kFirst + kCount - 1 of a block (kFirst even; not the DC term): their look-ups
// are all on their way before the first token is put together (one at a time every look-up is
// waited for alone), and what they add to the sums is added there and then.  (The tokens are not
// used before the second pass; left alone the compiler sinks their assembly and the sums down to
// there and keeps template, category and coefficient of all 63 alive instead.)
// back: 63 - (index of the last non-zero coefficient).
template <int kFirst, int kCount>
__device__ __forceinline__ void dense_batch(const u32 (&w)[32], const char* rows, u32 chromaAt, DenseBlock& d, u32& bits,
                                            u32& maxTok, u32& back) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    u32 e[kCount], cat[kCount], amp[kCount / 2];
#pragma unroll
    for (int k = 0; k < kCount / 2; ++k) {   // v for v >= 0, v - 1 (the one's complement of |v|) for v < 0, Encoder.cpp:281-284
        const s2 v = __builtin_bit_cast(s2, w[(kFirst >> 1) + k]);
        amp[k] = __builtin_bit_cast(u32, (s2)(v + (v >> 15)));
    }
#pragma unroll
    for (int k = 0; k < kCount; ++k) {
        const int i = kFirst + k;
        if (i == 0) continue;
        const int v = (int)(short)(w[i >> 1] >> (16 * (i & 1)));
        cat[k] = (u32)__builtin_amdgcn_frexp_expf((float)v);   // 0 for 0
        // row (i - 1 - last) = back + i - 64, column cat
        u32 at;   // back * pitch + (cat << 2) + chromaAt in two instructions (the compiler makes three of them)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(at) : "v"(back), "s"((u32)(kTabPitch * 4)), "v"((cat[k] << 2) + chromaAt));
        e[k] = *(const u32*)(rows + at + (i - 64) * (kTabPitch * 4));
        back = cat[k] != 0 ? (u32)(63 - i) : back;
    }
#pragma unroll
    for (int k = 0; k < kCount; ++k) {
        const int i = kFirst + k;
        if (i == 0) continue;
        const u32 t = e[k] | __builtin_amdgcn_ubfe(amp[k >> 1], 16u * (i & 1), cat[k]);
        d.tok[i] = t;
        bits += t >> 27;
        maxTok = max(maxTok, t);
    }
    asm volatile("" : "+v"(bits), "+v"(maxTok), "+v"(back));
#pragma unroll
    for (int k = 0; k < kCount; ++k)
        if (kFirst + k != 0) asm volatile("" : "+v"(d.tok[kFirst + k]));
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void dense_tokens(const u32 (&w)[32], int pred, const DenseTables& tab, u32 chroma, DenseBlock& d) {
    {   // Encoder.cpp:268-284: the DC term's difference, its category, its amplitude bits
        const int diff = (int)(short)(w[0] & 0xffffu) - pred;
        const int cat = __builtin_amdgcn_frexp_expf((float)diff);
        const u32 e = tab.dc[chroma * 16u + (u32)cat];
        d.tok[0] = e | __builtin_amdgcn_ubfe((u32)(diff + (diff >> 31)), 0u, (u32)cat);
    }
    u32 bits = d.tok[0] >> 27, maxTok = d.tok[0];
    u32 back = 63;   // (the DC term counts as the last non-zero coefficient, Encoder.cpp:304-306)
    const char* rows = (const char*)tab.ac;
    const u32 chromaAt = chroma * (u32)(kDenseAcWords * 4);
    dense_batch<0, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<8, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<16, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<24, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<32, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<40, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<48, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    dense_batch<56, 8>(w, rows, chromaAt, d, bits, maxTok, back);
    d.eob = back != 0 ? tab.eob[chroma] : 0u;
    d.bits = bits + (d.eob >> 27);
    d.maxTok = max(maxTok, d.eob);
}

// The length of a block's bit string alone, long runs of zeros included, with the same arithmetic
// (for the tiles dense_tokens cannot code; what walk_block with a LengthVisitor finds).
__device__ __forceinline__ u32 dense_lengths(const u32 (&w)[32], int pred, const DenseTables& tab, u32 chroma) {
    u32 bits;
    {
        const int diff = (int)(short)(w[0] & 0xffffu) - pred;
        const int cat = __builtin_amdgcn_frexp_expf((float)diff);
        const u32 e = tab.dc[chroma * 16u + (u32)cat];
        bits = e >= kTokAbsent ? kDenseNoCode : e >> 27;
    }
    u32 back = 63;
    const char* rows = (const char*)tab.acBits;
    const u32 chromaAt = chroma * (u32)(kDenseAcWords * 2);
#pragma unroll
    for (int i0 = 0; i0 < 64; i0 += 8) {
        u32 e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k;
            if (i == 0) continue;
            const int v = (int)(short)(w[i >> 1] >> (16 * (i & 1)));
            const u32 cat = (u32)__builtin_amdgcn_frexp_expf((float)v);
            const u32 at = back * (u32)(kTabPitch * 2) + ((cat << 1) + chromaAt);
            e[k] = *(const unsigned short*)(rows + at + (i - 64) * (kTabPitch * 2));
            back = cat != 0 ? (u32)(63 - i) : back;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k != 0) bits += e[k];
    }
    if (back != 0) {
        const u32 e = tab.eob[chroma];
        bits += e >= kTokAbsent ? kDenseNoCode : e >> 27;
    }
    return bits;
}

// ORs the block's tokens into the (zeroed) stage in LDS, the first at bit position pos (counted
// from bit 31 of stage word 0).
__device__ __forceinline__ void dense_put(const DenseBlock& d, u32* stage, u32 pos) {
#pragma unroll
    for (int i = 0; i < 65; ++i) {
        const u32 t = i < 64 ? d.tok[i] : d.eob;
        // The token's bits end at bit `end` of the stream, i.e. 64 - (pos & 31) - len bits above the bottom of
        // the two words at pos >> 5; the shift takes its amount modulo 64.  (Nothing to shift when len == 0.)
        const u32 end = pos + (t >> 27);
        const u64 v = (u64)(t & 0x07ffffffu) << (((pos & 32u) - end) & 63u);
        u32* at = stage + (pos >> 5);
        atomicOr(at, (u32)(v >> 32));
        atomicOr(at + 1, (u32)v);
        pos = end;
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- histograms ------------------------------------------------------------------------------
// The symbol histograms on the same footing: a wave lists the non-empty groups of a tile of
// blocks and counts the symbols of 64 groups at a time.  The lanes of a wave meet the same few
// symbols in the same instruction (the first coefficients of most blocks have category 1 or 2),
// and additions to one LDS address queue up one behind the other, so every counter exists
// kHistCopies times, side by side (on different banks), and a lane adds to the copy of its number
// (16384 x 16384 4:2:0: 211 us with one copy, 196 with two, 191 with four or eight).
constexpr int kHistTileBlocks = 256;              // four blocks per lane while the list is made
constexpr int kHistMaxItems = kHistTileBlocks * 8;   // a list of the pass's own: every group of a tile fits

// A tile without an empty group (noise, q = 100).  Its list says nothing -- item i is group i & 7 of block i >> 3,
// a wave's 64 lanes hold eight whole blocks -- and nearly every coefficient follows a non-zero one: the symbol is
// its category (run 0).  Those a lane counts in registers, without a branch and without LDS (an addition to a
// counter in LDS per coefficient is what the pass on lists waits for on such content): ten fields of six bits,
// luma and chroma apart, widened to twelve bits every seven groups and added to the workgroup's counters at the
// tile's end.  The few coefficients that follow zeros are walked one by one afterwards, and so is a whole group
// with a category beyond ten (pixels out of range).  8192 x 8192 noise at q = 100: 4:4:4 182 -> 135 us, 4:2:0
// 100 -> 86 us.  (Measured and dropped: the same counting in words of LDS of the lane's own, 202 us -- the LDS
// adds no faster to 64 addresses than to one; a lane per block, which reads 16 bytes of every line eight times
// over: 259 us.)
constexpr int kFullAhead = 2;   // chunks of a full tile asked for ahead (one to eight measured: no difference up to four, slower beyond)

template <int kHistCopies>
__device__ __forceinline__ void histogram_full_tile(const GroupTile& t, u32* mine, bool wantStats, u32* tileStats) {
    const int lane = threadIdx.x & 63;
    const u32 grp = (u32)lane & 7u;
    constexpr u64 kEvenFields = 0x003F03F03F03F03Full;   // fields 0, 2, 4, 6, 8 of six bits, twelve apart
    u64 evenY = 0, oddY = 0, evenC = 0, oddC = 0;        // categories 1, 3, .. and 2, 4, .. in fields of twelve bits
    u64 y6 = 0, c6 = 0;                                  // category k + 1 in bits 6 k + 5 .. 6 k, of up to seven groups
    u64 nzQuarters = 0;   // non-zero AC coefficients the lane has met in the four quarters of the tile, 16 bits each
    auto widen = [&]() {
        evenY += y6 & kEvenFields;
        oddY += (y6 >> 6) & kEvenFields;
        evenC += c6 & kEvenFields;
        oddC += (c6 >> 6) & kEvenFields;
        y6 = c6 = 0;
    };
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    struct Loaded {
        u32x4 d;
        u32 pred;
    };
    const int chunks = (t.nBlocks + 7) >> 3;
    auto block_of = [&](int chunk, bool& valid) {
        const int lb = 8 * chunk + (lane >> 3);
        valid = lb < t.nBlocks;
        return (u32)(valid ? lb : t.nBlocks - 1);   // (idle lanes read the last block and count nothing)
    };
    auto issue = [&](int chunk, Loaded& ld) {
        bool valid;
        const u32 lbc = block_of(chunk < chunks ? chunk : chunks - 1, valid);
        const u32 role = t.role[lbc];
        const u32 back = (role & 2u) ? 0u : ((role >> 2) & 31u);
        ld.d = __builtin_nontemporal_load((const u32x4*)(t.base + (((lbc + kGrpBaseBlocks) << t.shift) + (grp & 3u) * 16u + (grp >= 4u ? t.high : 0u))));
        ld.pred = *(const unsigned short*)(t.base + ((lbc + kGrpBaseBlocks - back) << t.shift));
    };
    // The groups of the next chunks are on their way while one is counted.
    Loaded ring[kFullAhead];
#pragma unroll
    for (int u = 0; u < kFullAhead; ++u) issue(u, ring[u]);
#pragma unroll 1
    for (int first = 0; first < chunks; first += kFullAhead) {
#pragma unroll
    for (int u = 0; u < kFullAhead; ++u) {
        const int chunk = first + u;
        if (chunk >= chunks) break;
        const Loaded got = ring[u];
        issue(chunk + kFullAhead, ring[u]);
        struct {
            u32x4 d;
            u32 pred, role;
            bool valid;
        } cur;
        cur.d = got.d;
        cur.pred = got.pred;
        cur.role = t.role[block_of(chunk, cur.valid)];
        const u32 chroma = cur.role & 1u;
        const bool isK0 = cur.valid && grp == 0u;
        u32 d[4] = {cur.valid ? cur.d.x : 0u, cur.valid ? cur.d.y : 0u, cur.valid ? cur.d.z : 0u, cur.valid ? cur.d.w : 0u};
        if (isK0) {   // Encoder.cpp:268-272: the DC term is coded as the difference to its predecessor (0 where a new interval starts)
            const int pred = (cur.role & 2u) ? 0 : (int)(short)cur.pred;
            const int diff = (int)(short)(d[0] & 0xffffu) - pred;
            d[0] = (d[0] & 0xffff0000u) | ((u32)diff & 0xffffu);
        }
        u32 a = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) a |= pk_min_u16(d[w], 0x00010001u) << (2 * w);
        u32 mm = (a | (a >> 15)) & 0xffu;   // coefficient j is not zero
        if (isK0) mm |= 1u;                 // runs are counted from the DC term whatever its difference is, Encoder.cpp:304-306
        // Zero coefficients at the end of the group in front (the lane in front: same block unless this is group 0).
        const u32 tz = (u32)__builtin_clz((mm << 24) | 0x00800000u);
        // (Fetched by all lanes: a lane that sits out has nothing to give to the lane behind it.)
        const u32 tzInFront = (u32)__builtin_amdgcn_update_dpp(0u, tz, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const u32 carry = grp == 0u ? 0u : tzInFront;
        const u32 acMask = isK0 ? mm & ~1u : mm;
        const u32 run0 = acMask & ((mm << 1) | (carry == 0u ? 1u : 0u));   // right behind a non-zero coefficient
        u32 cat[8];
        u64 acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = (int)(short)(d[j >> 1] >> (16 * (j & 1)));
            cat[j] = (u32)__builtin_amdgcn_frexp_expf((float)v);   // bits of |v|
            u32 field;   // 6 (category - 1) modulo 64 (a zero coefficient shifts a zero)
            asm("v_mad_u32_u24 %0, %1, 6, 58" : "=v"(field) : "v"(cat[j]));
            acc += (u64)__builtin_amdgcn_ubfe(run0, (u32)j, 1u) << (field & 63u);
        }
        const u32 largest = max(max(max(isK0 ? 0u : cat[0], cat[1]), max(cat[2], cat[3])), max(max(cat[4], cat[5]), max(cat[6], cat[7])));
        const bool counted = largest <= 10u;
        u32 rest = counted ? acMask & ~run0 : acMask;
        if (!counted) acc = 0;
        y6 += chroma ? 0ull : acc;
        c6 += chroma ? acc : 0ull;
        if ((chunk % 7) == 6) widen();
        u32* ac = mine + chroma * (256 * kHistCopies);
        if (isK0) atomicAdd(&mine[(2 * 256 + chroma * 16 + cat[0]) * kHistCopies], 1u);
        if (cur.valid && grp == 7u && (mm & 0x80u) == 0u) atomicAdd(&ac[0], 1u);   // Encoder.cpp:343-347
        if (rest != 0) {   // Encoder.cpp:304-341 for the coefficients behind zeros
            // clz(X << (16 - j)) is the zero run in front of coefficient j: flags above, a sentinel carry + 1 places
            // below the group's first flag.  (At most 14: sixteen zeros in a row would hold an empty group.)
            const u32 X = (mm << 16) | (1u << (15u - carry));
            do {   // (as many turns as the lane with the most of them has: one or two)
                const u32 j = (u32)__builtin_ctz(rest);
                rest &= rest - 1u;
                const u32 pair = j < 4u ? (j < 2u ? d[0] : d[1]) : (j < 6u ? d[2] : d[3]);
                const int v = (int)(short)(pair >> (16u * (j & 1u)));
                const u32 c = (u32)__builtin_amdgcn_frexp_expf((float)v);
                const u32 run = (u32)__builtin_clz(X << (16u - j)) & 15u;
                atomicAdd(&ac[((run << 4) | c) * kHistCopies], 1u);
            } while (rest != 0);
        }
        nzQuarters += (u64)(u32)__popc(acMask) << ((chunk >> 3) * 16);
    }
    }
    widen();
#pragma unroll
    for (int k = 0; k < 10; ++k) {   // category k + 1
        const u32 y = (u32)((k & 1 ? oddY : evenY) >> (12 * (k >> 1))) & 0xfffu;
        const u32 c = (u32)((k & 1 ? oddC : evenC) >> (12 * (k >> 1))) & 0xfffu;
        if (y) atomicAdd(&mine[(k + 1) * kHistCopies], y);
        if (c) atomicAdd(&mine[(256 + k + 1) * kHistCopies], c);
    }
    if (wantStats) {
        const u32 lo = (u32)__builtin_amdgcn_readlane(wave_scan_add((u32)nzQuarters), 63);
        const u32 hi = (u32)__builtin_amdgcn_readlane(wave_scan_add((u32)(nzQuarters >> 32)), 63);
        if (lane == 0) {
            atomicMax(&tileStats[0], (u32)t.nItems);
            atomicMax(&tileStats[1], (lo & 0xffffu) + (lo >> 16) + (hi & 0xffffu) + (hi >> 16));
            atomicMax(&tileStats[2], max(max(lo & 0xffffu, lo >> 16), max(hi & 0xffffu, hi >> 16)));
        }
    }
}

template <int kHistCopies>
__global__ __launch_bounds__(256) void k_histogram_tiles(KGeom g, const int16_t* __restrict__ coef, u64* __restrict__ hist,
                                                         HistogramExport ex) {
    // AC-Y, AC-C: 256 counters each, then DC-Y, DC-C: 16 each; kHistCopies of everything.
    __shared__ u32 cnt[(2 * 256 + 2 * 16) * kHistCopies];
    __shared__ unsigned short items[4][kHistMaxItems + 2];
    u32* lh = (u32*)&items[0][0];   // at the end: the sums, laid out as flush_histograms expects them (DC-Y, DC-C, AC-Y, AC-C)
    static_assert(sizeof(items) >= 4 * 256 * sizeof(u32), "the sums take the lists' place");
    __shared__ unsigned char roleTab[4][kGrpTileBlocks];
    __shared__ u32 tileStats[4];   // most items in a tile, most non-zero AC coefficients in a tile, in a quarter of a tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < (2 * 256 + 2 * 16) * kHistCopies; i += 256) cnt[i] = 0;
    if (threadIdx.x < 4) tileStats[threadIdx.x] = 0;
    __syncthreads();
    const bool wantStats = ex.stats != nullptr;
    u32* mine = cnt + (lane & (kHistCopies - 1));
    const u32 image = blockIdx.y;
    const u32 tiles = ((u32)g.blocksPerImage + kHistTileBlocks - 1) / kHistTileBlocks;
    GroupTile t;
    t.items = items[wave];
    t.cap = kHistMaxItems;
    t.role = roleTab[wave];
    for (u32 turn = blockIdx.x * 4u + (u32)wave; turn < tiles; turn += gridDim.x * 4u) {
        // From the image's end to its start: what the transform wrote last is still in the memory-side cache.
        const u32 tile = tiles - 1u - turn;
        t.first = tile * kHistTileBlocks;
        const u32 left = (u32)g.blocksPerImage - t.first;
        t.nBlocks = (int)(left < (u32)kHistTileBlocks ? left : (u32)kHistTileBlocks);
        t.shift = g.coefPitch == 64 ? 6u : 7u;
        t.high = (u32)g.coefHigh;
        t.base = (const char*)coef + (((long long)image * g.blocksPerImage + t.first - (long long)kGrpBaseBlocks) << t.shift);
        t.mask = g.groupMask + (long long)image * g.blocksPerImage + t.first;
        t.nItems = 0;
        t.tooMany = false;
        group_index_tile<false, true>(g, t);
        if (t.nItems == 8 * t.nBlocks) {
            histogram_full_tile<kHistCopies>(t, mine, wantStats, tileStats);
            continue;
        }
        u32 carryVal = ~0u;
        u64 nzQuarters = 0;   // non-zero coefficients the lane has met in the four quarters of the tile, 16 bits each
        GroupLoad ld;
        group_issue_loads<true>(t, 0, ld);
        for (int first = 0; first < t.nItems; first += 64) {
            GroupFront f;
            group_front(t, first, ld, carryVal, f);
            const int n = t.nItems - first < 64 ? t.nItems - first : 64;
            carryVal = __builtin_amdgcn_readlane(f.val, n - 1);
            group_issue_loads<true>(t, first + 64, ld);   // the next chunk's groups arrive while this one is counted
            if (wantStats)   // (idle lanes repeat the tile's last item with all flags clear; group 0's forced flag is taken off below)
                nzQuarters += (u64)(u32)__popc(f.mm & 0xffu) << ((f.lb >> 2) & 48u);
            u32* ac = mine + f.chroma * (256 * kHistCopies);
            if (f.zr != 0) atomicAdd(&ac[0xF0 * kHistCopies], (u32)f.zr);          // Encoder.cpp:318-328
            if (f.needEob) atomicAdd(&ac[0], 1u);                                  // Encoder.cpp:343-347
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = (int)(short)(f.d[j >> 1] >> (16 * (j & 1)));
                const u32 cat = (u32)__builtin_amdgcn_frexp_expf((float)v);   // bits of |v|
                if (j == 0 && f.isK0) {
                    atomicAdd(&mine[(2 * 256 + f.chroma * 16 + cat) * kHistCopies], 1u);
                } else if ((f.mm >> j) & 1u) {
                    const u32 run = (u32)__builtin_clz(f.X << (16 - j)) & 15u;
                    atomicAdd(&ac[((run << 4) | cat) * kHistCopies], 1u);
                }
            }
        }
        if (wantStats) {
            // A quarter of a tile has at most 64 * 64 of them: the sums of two quarters travel in one register.
            const u32 lo = (u32)__builtin_amdgcn_readlane(wave_scan_add((u32)nzQuarters), 63);
            const u32 hi = (u32)__builtin_amdgcn_readlane(wave_scan_add((u32)(nzQuarters >> 32)), 63);
            // Every block's group 0 counted its DC term as well: a quarter has 64 blocks (the image's last tile fewer).
            auto ac = [&](u32 sum, int quarter) {
                const int blocks = t.nBlocks - 64 * quarter;
                const u32 dc = (u32)(blocks < 0 ? 0 : (blocks > 64 ? 64 : blocks));
                return sum > dc ? sum - dc : 0u;
            };
            const u32 q0 = ac(lo & 0xffffu, 0), q1 = ac(lo >> 16, 1), q2 = ac(hi & 0xffffu, 2), q3 = ac(hi >> 16, 3);
            if (lane == 0) {
                atomicMax(&tileStats[0], (u32)t.nItems);
                atomicMax(&tileStats[1], q0 + q1 + q2 + q3);
                atomicMax(&tileStats[2], max(max(q0, q1), max(q2, q3)));
            }
        }
        // The list is this wave's own: the wave's next tile may overwrite it once the wave is here.
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int tab = i >> 8, sym = i & 255;   // DC-Y, DC-C, AC-Y, AC-C
        u32 v = 0;
        if (tab >= 2 || sym < 16) {
            const u32* c = cnt + (tab >= 2 ? (tab - 2) * 256 + sym : 2 * 256 + tab * 16 + sym) * kHistCopies;
#pragma unroll
            for (int k = 0; k < kHistCopies; ++k) v += c[k];
        }
        lh[i] = v;
    }
    __syncthreads();
    flush_histograms(lh, hist, ex, tileStats);
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
// dst, and (kClear) clears them in the stage.  The stream is zero where nothing has been
// written yet; the first and the last word of the range may be shared with neighbours and are merged.
// ffCount (may be null): the 0xFF bytes of the words are added to their chunks' counts on the way
// (see kChunkCountMask), one addition per chunk; the shared words at the ends count what their OR
// changed (or_counted).  Call with all lanes of the wave.
template <bool kClear>
__device__ __forceinline__ void stage_copy_out(u32* st, u32 srcWord, u32 bits, u64 dst, u32* __restrict__ words,
                                               u32* __restrict__ ffCount = nullptr) {
    if (bits == 0) return;
    const int lane = threadIdx.x & 63;
    const u32 s = (u32)(dst & 31);
    const u64 w0 = dst >> 5;
    const u32 nSrc = (bits + 31) >> 5, nOut = (s + bits + 31) >> 5;
    u64 curChunk = w0 >> kChunkWordShift;
    u32 acc = 0;
    auto flush = [&](u32 mine) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += (u32)__shfl_xor((int)mine, d, 64);
        if (lane == 0 && mine != 0) atomicAdd(&ffCount[curChunk], mine);
    };
    for (u32 base = 0; base < nOut; base += 64) {
        const u32 i = base + (u32)lane;
        u32 n = 0;
        bool inCur = true;
        if (i < nOut) {
            const u32 hi = i > 0 ? st[srcWord + i - 1] : 0u;
            const u32 lo = i < nSrc ? st[srcWord + i] : 0u;
            const u32 v = __builtin_bswap32(__builtin_amdgcn_alignbit(hi, lo, s));
            if (i == 0 || i == nOut - 1) {
                if (v != 0) {
                    if (ffCount != nullptr) {
                        const u32 old = atomicOr(&words[w0 + i], v);
                        n = (u32)__popc(ff_bits(old | v)) - (u32)__popc(ff_bits(old));
                    } else {
                        atomicOr(&words[w0 + i], v);
                    }
                }
            } else {
                words[w0 + i] = v;
                n = (u32)__popc(ff_bits(v));
            }
            inCur = ((w0 + i) >> kChunkWordShift) == curChunk;
        }
        if (ffCount == nullptr) continue;
        const u32 last = base + 63u < nOut ? base + 63u : nOut - 1u;
        const u64 lastChunk = (w0 + last) >> kChunkWordShift;   // the same for the whole wave
        if (lastChunk == curChunk) {
            acc += n;
        } else {   // a chunk ends inside these 64 words (at most one does)
            flush(acc + (inCur ? n : 0u));
            acc = inCur ? 0u : n;
            curChunk = lastChunk;
        }
    }
    if (ffCount != nullptr) flush(acc);
    if (kClear)
        for (u32 i = lane; i < nSrc; i += 64) st[srcWord + i] = 0;
}

// The same from a tile's slot in global memory.  All reads are issued before the first word is
// written: a loop that reads, merges and writes word by word waits for its own stores and
// atomics (one counter tracks them all), i.e. for several round trips to memory instead of one.
constexpr int kSlotCopyRounds = (kGrpStageWords + 1 + 63) / 64;
constexpr int kGrpSlotHead = kGrpStageWords + 64;   // where a slot's header starts
constexpr int kGrpSlotEdges = kGrpSlotHead + 4 + 2 * 24;   // where the four edge entries of a slot lie (8-byte aligned)
constexpr int kGrpSlotWords = kGrpSlotEdges + 8;   // stage and a row of slack; number of ends, image, summary; blocks and offsets of the ends; edges

__device__ __forceinline__ void slot_load(const u32* __restrict__ slot, u32 srcWord, u32 (&lo)[kSlotCopyRounds]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kSlotCopyRounds; ++k) {   // reads behind the part stay inside the slot and are dropped later
        const u32 i = srcWord + (u32)lane + 64u * k;
        lo[k] = slot[i < (u32)kGrpSlotWords ? i : (u32)kGrpSlotWords - 1u];
    }
}

// edge: where the first and the last word of the range go when they may be shared with a
// neighbour (two 64-bit entries, word number << 32 | bits, kNoEdge for none); k_merge_edges puts
// them into the stream.  An atomic at device scope instead costs the placing kernel half its time.
constexpr u64 kNoEdge = ~0ull;

// padN one bits at bit position padAt (the padding behind an interval end, when it is made of ones)
// are merged into the word they fall into, which is the range's last.
// ffCount (may be null): the 0xFF bytes of the words written here are added to their chunks'
// counts (see kChunkCountMask); the words at the edges are counted by k_merge_edges.
__device__ __forceinline__ void slot_place(u32 (&lo)[kSlotCopyRounds], u32 bits, u64 dst, u32* __restrict__ words,
                                           u64* __restrict__ edge, u32* __restrict__ ffCount, u64 padAt = 0, u32 padN = 0) {
    const int lane = threadIdx.x & 63;
    if (bits == 0) {
        if (lane == 0) edge[0] = edge[1] = kNoEdge;
        return;
    }
    const u32 s = (u32)(dst & 31);
    const u64 w0 = dst >> 5;
    const u32 nSrc = (bits + 31) >> 5, nOut = (s + bits + 31) >> 5;
#pragma unroll
    for (int k = 0; k < kSlotCopyRounds; ++k)
        if ((u32)lane + 64u * k >= nSrc) lo[k] = 0;
    if (lane == 0 && nOut == 1) edge[1] = kNoEdge;
    // A part is shorter than a chunk: its words lie in the chunk of its first word or in the next.
    const u64 firstChunk = w0 >> kChunkWordShift;
    u32 ffHere = 0, ffNext = 0;
#pragma unroll
    for (int k = 0; k < kSlotCopyRounds; ++k) {
        const u32 i = (u32)lane + 64u * k;
        // the word in front of this one: the neighbour lane's, or lane 63's of the round before
        const u32 before = k > 0 ? (u32)__builtin_amdgcn_readlane(lo[k > 0 ? k - 1 : 0], 63) : 0u;
        const u32 hi = __builtin_amdgcn_update_dpp(before, lo[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (i < nOut) {
            u32 v = __builtin_bswap32(__builtin_amdgcn_alignbit(hi, lo[k], s));
            if (padN != 0 && w0 + i == (padAt >> 5)) v |= __builtin_bswap32(((1u << padN) - 1u) << (32u - (u32)(padAt & 31) - padN));
            if (i == 0) edge[0] = ((w0 + i) << 32) | v;
            else if (i == nOut - 1) edge[1] = ((w0 + i) << 32) | v;
            else {
                words[w0 + i] = v;
                const u32 n = (u32)__popc(ff_bits(v));
                if (((w0 + i) >> kChunkWordShift) == firstChunk) ffHere += n; else ffNext += n;
            }
        }
    }
    if (ffCount != nullptr) {
        // counts of at most 4 * 11 per lane: both sums travel in one register
        u32 both = ffHere | (ffNext << 16);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) both += (u32)__shfl_xor((int)both, d, 64);
        if (lane == 0) {
            if ((both & 0xffffu) != 0) atomicAdd(&ffCount[firstChunk], both & 0xffffu);
            if ((both >> 16) != 0) atomicAdd(&ffCount[firstChunk + 1], both >> 16);
        }
    }
}

__device__ __forceinline__ void slot_copy_out(const u32* __restrict__ slot, u32 srcWord, u32 bits, u64 dst,
                                              u32* __restrict__ words, u64* __restrict__ edge, u32* __restrict__ ffCount) {
    u32 lo[kSlotCopyRounds];
    slot_load(slot, srcWord, lo);
    slot_place(lo, bits, dst, words, edge, ffCount);
}

// What a tile leaves behind when its bits are placed by a later kernel (k_place_tiles): the
// words of its stage, and the interval ends inside it.
static_assert(kGrpMaxFix == 24, "the slot's size is written out above");

// Places a tile's staged bits at `start` and notes where its restart intervals end.  `st`: the
// stage (LDS or the tile's slot), fixBlock / fixRel: nFix interval ends.
template <bool kClear>
__device__ __forceinline__ void place_tile(const KGeom& g, u32* st, PosSummary whole, u64 start, u32 nFix, const u32* fixBlock,
                                           const u32* fixRel, u32 image, u64* __restrict__ ivOff, u64* __restrict__ result,
                                           u32* __restrict__ words, long long zeroedBytes, u32 partWord = ~0u,
                                           u32* __restrict__ ffCount = nullptr) {
    // partWord: word of the stage at which the part behind the first interval end starts (given
    // by a caller that has placed the part in front of it already and passes h = 0 with its end as `start`).
    const int lane = threadIdx.x & 63;
    const u32 h = whole.x & 0x7fffffffu;
    const bool frontDone = partWord != ~0u;
    if (!frontDone) partWord = (((h + 31u) & ~31u) + 32u) >> 5;
    u64* edges = (u64*)(st + kGrpSlotEdges);   // (slots only)
    if (kClear) stage_copy_out<true>(st, 0, h, start, words, ffCount);   // (kClear: the stage in LDS; otherwise a slot in memory)
    else if (!frontDone) slot_copy_out(st, 0, h, start, words, edges, ffCount);
    if (whole.x >> 31) {
        const u64 partStart = (start + h + 7) & ~7ull;
        if (kClear) stage_copy_out<true>(st, partWord, whole.y, partStart, words, ffCount);
        else slot_copy_out(st, partWord, whole.y, partStart, words, edges + 2, ffCount);
        if (kClear && g.padOnes && partStart != start + h && lane == 0) {   // the first end's padding, as one bits (slots: see slot_place)
            const u64 at = start + h;
            const u32 nPad = (u32)(partStart - at);
            const u32 ones = __builtin_bswap32(((1u << nPad) - 1u) << (32u - (u32)(at & 31) - nPad));
            if (ffCount != nullptr) or_counted(words, at >> 5, ones, ffCount);
            else atomicOr(&words[at >> 5], ones);
        }
        if ((u32)lane < nFix && (u32)lane < (u32)kGrpMaxFix) {
            const u32 bl = fixBlock[lane];
            const long long iv = g.batch > 1 ? (long long)image : (long long)((bl / (u32)g.bpm) / (u32)g.mcusPerInterval);
            const u64 bytes = (partStart + fixRel[lane]) >> 3;
            ivOff[iv + 1] = bytes;
            if (ffCount != nullptr) {
                // The markers at this interval's end are no data: EOI in front of `bytes` (the image
                // ends), or the next interval's RSTn at `bytes`.
                const bool lastOne = g.batch > 1 || iv + 1 == g.intervals;
                if (g.writeEoi && lastOne) atomicAdd(&ffCount[(bytes - 2) >> 12], kChunkMarkerDelta);
                if (g.batch == 1 && !lastOne) atomicAdd(&ffCount[bytes >> 12], kChunkMarkerDelta);
            }
            if (iv + 1 == g.intervals) {
                result[2] = bytes;
                result[3] = (bytes + kStuffChunk - 1) / kStuffChunk;
                if ((long long)bytes > zeroedBytes) atomicOr((unsigned long long*)&result[1], 4ull);
            }
        }
    }
}

// kDeferred: the tile does not wait for its start position; it leaves its staged bits and its
// summary behind (slots, state) and k_place_tiles puts them into the stream once every tile's
// start is known.  Otherwise the tiles chain with a look-back and place their bits themselves.
template <bool kDeferred>
__global__ __launch_bounds__(256) void k_encode_groups(KGeom g, const int16_t* __restrict__ coef,
                                                       const u32* __restrict__ tables, u64* __restrict__ state,
                                                       u64* __restrict__ ivOff, u64* __restrict__ result,
                                                       u32* __restrict__ words, int tileBlocks, long long zeroedBytes,
                                                       u32* __restrict__ slots, u32* __restrict__ ffCount) {
    // ffCount (tiles that place their bits themselves; may be null): per-chunk counts of the stream's
    // 0xFF bytes, zero at the start, kept up to date as the words are written.
    __shared__ u32 tab[kTabWords];
    __shared__ unsigned short items[4][kGrpMaxItems + 2];
    __shared__ unsigned char roleTab[4][kGrpTileBlocks];
    __shared__ u32 stage[4][kGrpStageWords];
    __shared__ u32 fixBlock[4][kGrpMaxFix], fixRel[4][kGrpMaxFix];
    __shared__ unsigned long long ticket;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Tiles per image in fours, so that the four tiles of a workgroup share their image's tables.
    const u32 tilesPerImage = (((u32)g.blocksPerImage + (u32)tileBlocks - 1) / (u32)tileBlocks + 3u) & ~3u;
    const long long tiles = (long long)tilesPerImage * g.batch;
    // Tile numbers come from a counter: every predecessor of a running tile is running or done
    // (nobody waits for anybody when the placement is deferred).
    // Deferred tiles do not depend on each other: they are taken from the band's end, where the memory-side
    // cache still holds what the transform wrote last (the histogram pass before this one reads with
    // streaming loads and leaves it there).
    if (threadIdx.x == 0) ticket = kDeferred ? 4ull * (gridDim.x - 1u - blockIdx.x) : atomicAdd((unsigned long long*)(state + tiles), 4ull);
    for (int i = threadIdx.x; i < 4 * kGrpStageWords; i += 256) (&stage[0][0])[i] = 0;
    if (g.batch == 1) load_token_templates(tables, tab);   // (one picture: they come in while the ticket is on its way)
    __syncthreads();
    const u32 tile = (u32)__builtin_amdgcn_readfirstlane((u32)ticket) + (u32)wave;
    const u32 image = tile / tilesPerImage;
    if (g.batch > 1) {
        load_token_templates(tables + (long long)image * g.tableStride, tab);
        __syncthreads();
    }
    // No workgroup barrier from here on: the waves go their own ways (a wave's look-back may
    // wait for a tile of another wave of the same workgroup).

    GroupTile t;
    t.items = items[wave];
    t.cap = kGrpMaxItems;
    t.role = roleTab[wave];
    u32* st = stage[wave];
    const u32 tileInImage = tile - image * tilesPerImage;
    const long long firstLL = (long long)tileInImage * tileBlocks;
    t.first = (u32)(firstLL < g.blocksPerImage ? firstLL : g.blocksPerImage);
    {
        const long long left = g.blocksPerImage - (long long)t.first;
        t.nBlocks = (int)(left < tileBlocks ? left : tileBlocks);
    }
    t.shift = g.coefPitch == 64 ? 6u : 7u;
    t.high = (u32)g.coefHigh;
    t.base = (const char*)coef + (((long long)image * g.blocksPerImage + t.first - (long long)kGrpBaseBlocks) << t.shift);
    t.mask = g.groupMask + (long long)image * g.blocksPerImage + t.first;
    t.nItems = 0;
    t.tooMany = false;
    if (t.nBlocks > 0) group_index_tile<true>(g, t);   // (a tile behind the image's last one is empty)

    // Position inside the tile's part of the stream, as a function of the position P at
    // which the tile starts: P + runX while no interval has ended inside the tile, after that
    // roundup8(P + runX) + runY (PosSummary).  Bits in front of the first interval end are
    // staged from bit 0, the rest from a word boundary behind them, and the two pieces go to the
    // stream separately once P is known.
    u32 runX = 0, runY = 0;
    u32 carryVal = ~0u, nFix = 0;
    u32 maxTok = 0;
    bool overflow = t.tooMany;
    GroupLoad ld;
    if (t.nItems > 0) group_issue_loads<kDeferred>(t, 0, ld);
    for (int first = 0; first < t.nItems; first += 64) {
        GroupFront f;
        group_front(t, first, ld, carryVal, f);
        const int n = t.nItems - first < 64 ? t.nItems - first : 64;
        carryVal = __builtin_amdgcn_readlane(f.val, n - 1);
        group_issue_loads<kDeferred>(t, first + 64, ld);   // the next chunk's groups arrive while this one is coded

        // Tokens of the eight coefficients, with their bit counts.
        typedef short s2 __attribute__((ext_vector_type(2)));
        u32 amp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // v for v >= 0, v - 1 (the one's complement of |v|) for v < 0, Encoder.cpp:281-284
            const s2 v = __builtin_bit_cast(s2, f.d[i]);
            amp[i] = __builtin_bit_cast(u32, (s2)(v + (v >> 15)));
        }
        u32 tok[8], off[9];
        const u32 zrlTok = tab[kTabZrl + f.chroma], eobTok = tab[kTabEob + f.chroma];
        off[0] = (u32)f.zr * (zrlTok >> 27);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int cat;
            const u32 o = group_symbol_offset(f, j, cat);
            const u32 e = *(const u32*)((const char*)tab + o);
            tok[j] = e | __builtin_amdgcn_ubfe(amp[j >> 1], 16 * (j & 1), (u32)cat);
            off[j + 1] = off[j] + (tok[j] >> 27);
        }
        {   // a token of the "no code" kind is larger than any other (four v_max3_u32)
            u32 m = max(max(maxTok, tok[0]), tok[1]);
            m = max(max(m, tok[2]), tok[3]);
            m = max(max(m, tok[4]), tok[5]);
            maxTok = max(max(m, tok[6]), tok[7]);
        }
        if (f.zr != 0 && zrlTok > maxTok) maxTok = zrlTok;
        if (f.needEob && eobTok > maxTok) maxTok = eobTok;
        const u32 eobBits = f.needEob ? eobTok >> 27 : 0u;
        const u32 len = f.real ? off[8] + eobBits : 0u;

        // Where the lane's bits go in the stage.
        u32 base;
        if (__ballot(f.boundary) == 0) {
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
            const u32 bl = t.first + f.lb;
            const bool lastBlock = bl == (u32)g.blocksPerImage - 1;
            const u32 eoiBits = f.boundary && g.writeEoi && (g.batch > 1 || lastBlock) ? 16u : 0u;
            const u32 rstBits = f.boundary && g.batch == 1 && !lastBlock ? 16u : 0u;
            PosSummary mine = {len, 0u};
            if (f.boundary) {
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
            const u64 ends = __ballot(f.boundary);
            if (f.boundary) {
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
                const u32 slot = nFix + __builtin_amdgcn_mbcnt_hi((u32)(ends >> 32), __builtin_amdgcn_mbcnt_lo((u32)ends, 0u));
                if (slot < (u32)kGrpMaxFix) {
                    fixBlock[wave][slot] = bl;
                    fixRel[wave][slot] = padded + eoiBits;
                } else {
                    overflow = true;
                }
            }
            nFix += (u32)__popcll(ends);
            const PosSummary whole = pos_combine(run, PosSummary{(u32)__builtin_amdgcn_readlane(inc.x, 63),
                                                                 (u32)__builtin_amdgcn_readlane(inc.y, 63)});
            runX = whole.x;
            runY = whole.y;
        }
        runX = __builtin_amdgcn_readfirstlane(runX);
        runY = __builtin_amdgcn_readfirstlane(runY);

        if (f.real && base + len + 64u > (u32)kGrpStageBits) overflow = true;
        const bool fits = f.real && base + len + 64u <= (u32)kGrpStageBits;
        if (__ballot(fits && len > 64u) == 0) {
            // No lane of the chunk has more than 64 bits (photographic content at usual qualities
            // never has): the lane strings its tokens together in a register pair and puts them into
            // the stage with three ORs instead of two per token.
            if (fits && len != 0) {
                u64 acc = 0;
                const u32 zl = zrlTok >> 27, zb = zrlTok & 0x07ffffffu;
#pragma unroll
                for (int z = 0; z < 3; ++z)
                    if (z < f.zr) acc = (acc << zl) | zb;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = (acc << (tok[j] >> 27)) | (tok[j] & 0x07ffffffu);
                acc = (acc << eobBits) | (f.needEob ? eobTok & 0x07ffffffu : 0u);
                const u64 v = acc << (64u - len);   // the string, from bit 63 down
                const u32 hi = (u32)(v >> 32), lo = (u32)v, sh = base & 31u;
                u32* w = st + (base >> 5);
                atomicOr(w, __builtin_amdgcn_alignbit(0u, hi, sh));
                const u32 w1 = __builtin_amdgcn_alignbit(hi, lo, sh), w2 = __builtin_amdgcn_alignbit(lo, 0u, sh);
                if (w1 != 0) atomicOr(w + 1, w1);
                if (w2 != 0) atomicOr(w + 2, w2);
            }
        } else if (fits) {
            if (f.zr != 0) {
                const u32 zl = zrlTok >> 27;
                for (int z = 0; z < f.zr; ++z) stage_put(st, base + (u32)z * zl, zrlTok & 0x07ffffffu, zl);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) stage_put(st, base + off[j], tok[j] & 0x07ffffffu, tok[j] >> 27);
            if (f.needEob) stage_put(st, base + off[8], eobTok & 0x07ffffffu, eobBits);
        }
    }

    const PosSummary whole = {runX, runY};
    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    u32 flags = 0;
    if (__ballot(overflow) != 0) flags |= 8u;
    if (__ballot(maxTok >= kTokAbsent) != 0) flags |= 16u;
    if (kDeferred) {
        // The tile's summary, its staged bits and its interval ends, for k_place_tiles.
        u32* slot = slots + (size_t)tile * kGrpSlotWords;
        const u32 h = whole.x & 0x7fffffffu;
        const u32 used = (whole.x >> 31) ? ((((h + 31u) & ~31u) + 32u + whole.y + 31u) >> 5) : ((h + 31u) >> 5);
        for (u32 i = lane; i < used && i < (u32)kGrpStageWords; i += 64) slot[i] = st[i];
        if (lane == 0) {
            state[tile] = ((u64)whole.x << 32) | whole.y;
            slot[kGrpSlotHead] = nFix;
            slot[kGrpSlotHead + 1] = image;
            slot[kGrpSlotHead + 2] = whole.x;
            slot[kGrpSlotHead + 3] = whole.y;
        }
        if ((u32)lane < nFix && (u32)lane < (u32)kGrpMaxFix) {
            slot[kGrpSlotHead + 4 + lane] = fixBlock[wave][lane];
            slot[kGrpSlotHead + 4 + kGrpMaxFix + lane] = fixRel[wave][lane];
        }
        if (flags != 0 && lane == 0) atomicOr((unsigned long long*)&result[1], (unsigned long long)flags);
        return;
    }
    // The tile's summary is complete: publish it, find the tile's start, publish that.
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
    if (!ok) flags |= 2u;
    place_tile<true>(g, st, whole, start, nFix, fixBlock[wave], fixRel[wave], image, ivOff, result, words, zeroedBytes, ~0u,
                     ffCount);
    if (tile == 0 && lane == 0) {
        ivOff[0] = 0;
        if (p0 != 0) {   // the band does not start the image: its first interval's RSTn
            const u32 mark = __builtin_bswap32((0xffd0u + (u32)((g.firstInterval - 1) & 7)) << 16);
            if (ffCount != nullptr) {
                or_counted(words, 0, mark, ffCount);
                atomicAdd(&ffCount[0], kChunkMarkerDelta);
            } else {
                atomicOr(&words[0], mark);
            }
        }
    }
    if (flags != 0 && lane == 0) atomicOr((unsigned long long*)&result[1], (unsigned long long)flags);
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

// Deferred placement, positions: the tiles' summaries to start positions, in one launch.  One wave
// per run of 64 tiles replaces each tile's summary by the summary of the tiles in front of it
// inside the run and notes the run's own summary; the wave that finishes last turns the runs'
// summaries into the positions at which the runs start.  A tile then starts at
// pos_apply(summary[tile], runSummary[tile / 64]).
// `done` counts the waves that have finished and is zero at the start.  No fences (a release at
// device scope writes a whole L2 back): a run's summary is put down with an exchange that
// returns, so it has been performed when the wave counts itself off, and the last wave reads the
// summaries where the exchanges were made.
constexpr int kPositionRunsPerWave = 4;
__global__ __launch_bounds__(64) void k_tile_positions(KGeom g, u64* __restrict__ summary, u64* __restrict__ runSummary,
                                                       u32 tiles, u64* __restrict__ done) {
    // (A wave takes kPositionRunsPerWave runs: the count-off is one address for all waves, and additions to one
    // address from all over the device are performed one behind the other, some twenty nanoseconds each.)
    const int lane = threadIdx.x;
    const u32 runs = (tiles + 63u) / 64u;
    u64 v[kPositionRunsPerWave];
#pragma unroll
    for (int k = 0; k < kPositionRunsPerWave; ++k) {
        const u32 t = (blockIdx.x * (u32)kPositionRunsPerWave + (u32)k) * 64u + (u32)lane;
        v[k] = t < tiles ? summary[t] : 0ull;
    }
    u64 old = 0;
#pragma unroll
    for (int k = 0; k < kPositionRunsPerWave; ++k) {
        const u32 r = blockIdx.x * (u32)kPositionRunsPerWave + (u32)k;
        const u32 t = r * 64u + (u32)lane;
        const PosSummary inc = wave_scan_summary(PosSummary{(u32)(v[k] >> 32), (u32)v[k]});
        PosSummary excl;
        excl.x = __shfl_up(inc.x, 1, 64);
        excl.y = __shfl_up(inc.y, 1, 64);
        if (lane == 0) excl.x = excl.y = 0;
        if (t < tiles) summary[t] = ((u64)excl.x << 32) | excl.y;
        if (lane == 63 && r < runs) old |= atomicExch((unsigned long long*)&runSummary[r], ((u64)inc.x << 32) | inc.y);
    }
    u64 seen = 0;
    if (lane == 63) {
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the exchanges have been performed
        // (the returned values are of no interest -- whatever the buffer held before -- but asking for them
        // here keeps the exchanges, and the wait for them, in front of the count-off)
        asm volatile("" : : "v"((u32)old), "v"((u32)(old >> 32)) : "memory");
        seen = atomicAdd((unsigned long long*)done, 1ull) + 1;
    }
    const u32 count = (u32)__builtin_amdgcn_readlane((u32)seen, 63);
    if (count != gridDim.x) return;
    // Last wave: 64 runs at a time, the runs in front carried along.  The summaries of up to kAhead * 64 runs
    // are asked for in one go (one round trip to memory instead of one per 64 runs: 16384 x 16384 4:2:0 has 384).
    const u64 p0 = 8ull * (u64)interval_marker_bytes(g, 0);
    PosSummary before = {0u, 0u};
    constexpr int kAhead = 8;
    for (u32 first = 0; first < runs; first += 64 * kAhead) {
        u64 w[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const u32 r = first + 64u * k + (u32)lane;
            w[k] = r < runs ? __hip_atomic_load(&runSummary[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            if (first + 64u * k >= runs) break;
            const u32 r = first + 64u * k + (u32)lane;
            const PosSummary in = wave_scan_summary(PosSummary{(u32)(w[k] >> 32), (u32)w[k]});
            PosSummary ex;
            ex.x = __shfl_up(in.x, 1, 64);
            ex.y = __shfl_up(in.y, 1, 64);
            if (lane == 0) ex.x = ex.y = 0;
            if (r < runs) runSummary[r] = pos_apply(pos_combine(before, ex), p0);
            before = pos_combine(before, PosSummary{(u32)__builtin_amdgcn_readlane(in.x, 63), (u32)__builtin_amdgcn_readlane(in.y, 63)});
        }
    }
}

// Deferred placement (see k_encode_groups): one wave per tile; state[tile] holds the summary of the
// tiles in front of it inside its run of 64, runStart the position at which the run starts
// (k_tile_positions).
__global__ __launch_bounds__(256) void k_place_tiles(KGeom g, const u64* __restrict__ state, const u64* __restrict__ runStart,
                                                     u32* __restrict__ slots, u64* __restrict__ ivOff, u64* __restrict__ result,
                                                     u32* __restrict__ words, u32 tiles, long long zeroedBytes,
                                                     u32* __restrict__ ffCount) {
    const int lane = threadIdx.x & 63;
    const u32 tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    u32* slot = slots + (size_t)tile * kGrpSlotWords;
    // Everything that does not depend on anything else is asked for at once: the tile's place
    // in its run, the run's start, the slot's header, and the words of the part in front of the
    // first interval end (as many as a stage holds; the length decides later which count).
    u32 lo[kSlotCopyRounds];
    slot_load(slot, 0, lo);
    const u64 before = state[tile];
    const u64 runAt = runStart[tile >> 6];
    const u32 nFix = slot[kGrpSlotHead], image = slot[kGrpSlotHead + 1];
    const PosSummary whole = {slot[kGrpSlotHead + 2], slot[kGrpSlotHead + 3]};
    const u64 start = pos_apply(PosSummary{(u32)(before >> 32), (u32)before}, runAt);
    const u32 h = whole.x & 0x7fffffffu;
    u64* edges = (u64*)(slot + kGrpSlotEdges);
    const u64 at = start + h;
    // Nothing is written outside the stream's buffer, whatever the positions say: a tile whose
    // part would end behind it reports a broken chain instead (the host then fails the call).
    if (((at + 7) >> 3) + ((whole.x >> 31) ? ((u64)whole.y >> 3) + 1 : 0) + 8 > (u64)zeroedBytes) {
        if (lane == 0) {
            edges[0] = edges[1] = edges[2] = edges[3] = kNoEdge;
            atomicOr((unsigned long long*)&result[1], 2ull);
        }
        return;
    }
    const u32 padN = (whole.x >> 31) && g.padOnes ? (u32)(((at + 7) & ~7ull) - at) : 0u;
    slot_place(lo, h, start, words, edges, ffCount, at, padN);
    if ((whole.x >> 31) == 0) {
        if (lane == 0) edges[2] = edges[3] = kNoEdge;
    } else {
        place_tile<false>(g, slot, PosSummary{0x80000000u, whole.y}, at, nFix, slot + kGrpSlotHead + 4,
                          slot + kGrpSlotHead + 4 + kGrpMaxFix, image, ivOff, result, words, zeroedBytes, (((h + 31u) & ~31u) + 32u) >> 5,
                          ffCount);
    }
    if (tile == 0 && lane == 0) ivOff[0] = 0;
}

// The stream words at the edges of the tiles' parts: entries with the same word number are
// ORed together (neighbours meet inside a word) and the first of such a run writes the word.
// A band that does not begin the image starts with the RSTn of the interval in front of it.
__global__ __launch_bounds__(256) void k_merge_edges(KGeom g, const u32* __restrict__ list, u32 strideWords, u32 perTileLog2,
                                                     u32* __restrict__ words, u32 tiles, u32* __restrict__ ffCount) {
    // list: the first tile's entries; a tile has 1 << perTileLog2 of them, strideWords 32-bit words behind the tile's in front.
    const long long n = (long long)tiles << perTileLog2;
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    auto entry = [&](long long k) {
        return ((const u64*)(list + (size_t)(k >> perTileLog2) * strideWords))[k & ((1u << perTileLog2) - 1u)];
    };
    if (c == 0 && interval_marker_bytes(g, 0) != 0) {
        // bits 31..16 of word 0; the first tile's first entry, if it lies in word 0 too, brings the rest
        u32 v = __builtin_bswap32((0xffd0u + (u32)((g.firstInterval - 1) & 7)) << 16);
        for (long long k = 0; k < n; ++k) {
            const u64 e = entry(k);
            if (e == kNoEdge) continue;
            if ((e >> 32) != 0) break;
            v |= (u32)e;
        }
        words[0] = v;
        // (the RSTn's own 0xFF is a marker's)
        atomicAdd(&ffCount[0], (u32)__popc(ff_bits(v)) + kChunkMarkerDelta);
        return;
    }
    if (c >= n) return;
    const u64 mine = entry(c);
    if (mine == kNoEdge) return;
    const u64 w = mine >> 32;
    if (w == 0 && interval_marker_bytes(g, 0) != 0) return;   // written above
    for (long long k = c - 1; k >= 0; --k) {   // is there an entry for the same word in front?  then that one writes
        const u64 e = entry(k);
        if (e == kNoEdge) continue;
        if ((e >> 32) == w) return;
        break;
    }
    u32 v = (u32)mine;
    for (long long k = c + 1; k < n; ++k) {
        const u64 e = entry(k);
        if (e == kNoEdge) continue;
        if ((e >> 32) != w) break;
        v |= (u32)e;
    }
    words[w] = v;
    const u32 ffs = (u32)__popc(ff_bits(v));
    if (ffs != 0) atomicAdd(&ffCount[w >> kChunkWordShift], ffs);
}
