// spl_k_tile.h -- part of spl_kernels.hip (included there, in this order; one translation unit): tile geometry, the tile kernel's LDS layout, and the tail that finishes a tile's long chunks (bpe_tail_segments).
#pragma once

namespace spl {

// Tile geometry is a template parameter: small batches use small tiles (many wavefronts, 4 bytes
// per lane, latency hidden by occupancy), large batches use 4 KiB tiles (less halo overhead).
// Every phase maps ONE 4-byte word of the window to one lane, so LDS traffic is bank-conflict free.
template <int TB_, int RH_> struct TileGeom {
    static constexpr int TBv = TB_;
    static constexpr int Wv = LH + TB_ + RH_;            // staged bytes
    static constexpr int NW32 = (Wv + WPAD) / 4;         // dwords of text / records
    static constexpr int NBW = Wv / 32 + 1;              // bitmap words incl. the bit for position W
    static constexpr int C16 = Wv / 2 + 1;               // miss list capacities: chunks of 2..16 bytes,
    static constexpr int C64 = Wv / 17 + 2;              //   17..64 bytes
    static constexpr int QCAP = C16 + C64;
    static_assert(Wv % 32 == 0 && NBW <= NT, "window must be a multiple of 32 bytes and fit one scan");
};

// EXPORT_MEDIUM (large batches): the 17..64-byte misses are not merged here but appended to the
// global q64 (one atomic per workgroup) for k_bpe_lanes64, which works them one lane per chunk --
// with hundreds of thousands of such chunks in flight (CJK text) that is the throughput-optimal
// shape; for small batches the latency-optimal in-kernel groups are used instead.
//
// DIRECT (tile-owned mode, batches without special tokens): the workgroup finishes EVERYTHING that
// starts in its tile and leaves a self-contained record.  Text-start bits come from a search of
// doc_off (no k_mark_docs, no bitmap to clear), token ids are kept in LDS, chunks longer than 64
// bytes and the (at most one) chain that outgrows the window are finished right here; the tile's
// window tokens are packed into tile_ids[] at a slot taken from one atomic cursor, its token
// count is added to the sum of its 64-tile group, and the documents that start in the tile get
// their LOCAL rank.  k_tile_out then only has to add each tile's base.  Two launches instead of
// seven, no workgroup ever waits for another one, and stage[] / tbits[] in HBM are touched only
// by tokens that start beyond the window (k_tile_out cleans those bits up again).
// (A decoupled look-back inside this kernel -- ONE launch -- measured 1.4 us faster on the 1 MB
//  bench batch but collapses when tile times vary: tiles wait, resident, for the slowest
//  predecessor.  8 MB of CJK-heavy text took 4.4 ms that way and 1.3 ms this way.)
#ifndef SPL_WORK_PRIO
#define SPL_WORK_PRIO 3
#endif
#ifndef SPL_MERGE_PRIO
#define SPL_MERGE_PRIO 2        /* (1 was right while the chains ran at 3; since the start masks: 2, k_pretok 34.5 -> 33.7 us) */
#endif
#ifndef SPL_MEDIUM_PRIO
#define SPL_MEDIUM_PRIO 2
#endif
#ifndef SPL_PRETOK_WAVES
#define SPL_PRETOK_WAVES 6
#endif
#ifndef SPL_MEDIUM_PAIRS
#define SPL_MEDIUM_PAIRS 1        /* 1: chunks of 17..32 bytes merge two to a wavefront (32 lanes each) */
#endif
#ifndef SPL_MASK_STARTS
#define SPL_MASK_STARTS 1         /* 1: cl100k tiles take their match starts from the bit-vector computation of spl_scan_starts.h */
#endif
#ifndef SPL_TILE_MISS_LIST
#define SPL_TILE_MISS_LIST 0      /* 1: EVERY miss of a tile through the workgroup-wide segment pass of the tail instead of the
                                     per-wavefront merge loops.  Measured on the bench batch: 60 us against 42 us per launch --
                                     fewer instructions, but the tail's ~20 workgroup barriers serialise what the wavefronts
                                     otherwise do independently (profiles/r02_notes.md).  Kept for A/B builds. */
#endif
#ifndef SPL_LQ_MEDIUM
#define SPL_LQ_MEDIUM 16
#endif
constexpr int DIRECT_LQ_MEDIUM = SPL_LQ_MEDIUM;      // of which, from the back: medium chunks of multi-byte text
constexpr int DIRECT_LQCAP = 32;          // long-chunk list of one workgroup (refilled while a chain is continued)
constexpr int DIRECT_WIN = 2048;           // bytes staged per turn for a chain that continues beyond the window
constexpr int DIRECT_WAVE_NMAX = 256;     // nodes of one wavefront's LDS slab in the single-pass tail

template <int TB_, int RH_> struct PretokScanLds {           // dead once the merge loop is done
    using G = TileGeom<TB_, RH_>;
    uint32_t rec32[G::NW32];
    uint32_t mk[MK_COUNT * (G::NBW + 1)];                // class bitmasks of the window (spl_scan_masks.h)
    uint32_t sub[NT / 16][16 * SUB_W];                   // per 16-lane group: tabulated substring ids
    uint32_t miss[G::QCAP];                              // p | n << 16, one region per size class
};
constexpr int DIRECT_TAB_NMAX = 128;      // chunks up to this size: tabulated wavefront merge (bpe_wave_tab<2>)
struct PretokTailLds {                                   // tile-owned tail: one slab per wavefront, used either as
    uint32_t slab[NT / 64][DIRECT_TAB_NMAX * SUB_W];     // bpe_wave_tab's table or as bpe_wave's node arrays
};
static_assert(DIRECT_TAB_NMAX * SUB_W >= 3 * DIRECT_WAVE_NMAX, "a slab must hold bpe_wave's id, rank and link arrays");
static_assert((NT / 64) * DIRECT_TAB_NMAX * SUB_W * 4 >= 2 * (DIRECT_WIN + 32), "the slab must hold a chain window's text and records");
static_assert(2 * DIRECT_TAB_NMAX * SUB_W >= 3 * 512, "two slabs must hold bpe_wave's arrays for 512 nodes");
constexpr int DIRECT_BLOCK_NMAX = 1024;   // workgroup-wide LDS node list in the whole slab
static_assert((NT / 64) * DIRECT_TAB_NMAX * SUB_W >= 3 * DIRECT_BLOCK_NMAX, "the slab must hold the workgroup-wide list");

// Tile-owned tail: the long chunks of a tile, several at a time, through the table of substring
// ids and the segments between boundaries that no token spans (see bpe_wave64_tab).  The chunks
// are laid end to end over up to SEG_ROWS table rows, one row per thread -- a chunk's end is such a
// boundary by construction -- and filled with two batches of probes for ALL of them together;
// then the 16 groups of 16 lanes take the segments of up to 16 bytes (each group those that start
// in its 16 rows), wavefronts take those of 17..64 bytes, and a chunk with a longer segment is
// left on the list for the merge loops below.  Chinese text is chunks of 60..200 bytes made of
// 3-byte segments: two memory round trips and a few two-step loops per tile, where the node-list
// loops pay a round trip per merge.
#ifndef SPL_SEG_ROWS
#define SPL_SEG_ROWS 448         /* rows (bytes) of one pass: up to two per thread since round 5 (one per thread up to round 4).  448, not 512:
                                    448 rows of seven cells and the 448 words of sid[] fill the tail's slab exactly -- with 512 the kernel
                                    needs 2 KB of LDS more, and a batch that fills the GPU several times over loses 1.5 % (C2, 8 MB) */
#endif
#ifndef SPL_TAIL_SKIP_EMPTY
#define SPL_TAIL_SKIP_EMPTY 1    /* 1: no chunk left behind the segment passes (nearly always): none of the caller's three node-list loops, nor their barriers */
#endif
// Round 5: a pass works per CHARACTER where round 4 worked per byte.  Two in three bytes of CJK text are continuation bytes, where
// (almost) no token starts -- yet every byte row ran the full tabulation: a prefix entry, a filter byte, six probes and the p8
// bucket, ~200 branch-free instructions, and a pass of 256 rows held 85 characters (profiles/r04_tail_cost.txt: the tail was 35 %
// of C3's kernel time, 40 % of C5's).  Now a pass holds up to SEG_ROWS = 448 byte rows and fills them in two steps:
//   1. every row reads its prefix entry (one 8-byte load).  A continuation byte whose two-byte prefix allows nothing beyond the
//      two-byte token (98 % of them, measured on the CJK generator for cl100k / o200k / deepseek_v3) is LIGHT: cell 0 is that
//      token's id, the other cells are empty, done.  Every other row -- lead bytes, ASCII, a chunk's first byte, the rare
//      continuation byte that starts something longer -- is HEAVY and goes on a dense list;
//   2. the heavy rows, one per lane off that list, get the full tabulation (row_head + row_fill + p8): a wavefront's 64 lanes
//      are 64 rows that need it.  512 bytes of Chinese text: 171 + a few heavy rows -- three wavefronts, one round of probes --
//      where round 4 ran two passes of four wavefronts each.
// The rows stay one per BYTE (node i of a chunk lives at byte offset i: the merge loops are the ones of round 4); the segments
// of up to 8 bytes -- nearly all of CJK text -- are likewise put on a dense list first and merged one per lane.
#ifndef SPL_TAIL_CUT
#define SPL_TAIL_CUT 0           /* timing experiments only (tokens missing): 1 no segment merges, 2 no heavy-row tabulation either, 3 no per-row step either */
#endif
constexpr int SEG_ROWS = SPL_SEG_ROWS;
constexpr int SEG_RPT = (SEG_ROWS + NT - 1) / NT;     // rows per thread in the row-indexed steps: thread t has rows t, t + NT, ...
constexpr int SEG_PAD = SEG_RPT * NT;                 // (bitmaps over the rows are sized for whole rounds of threads)
static_assert(SEG_ROWS % 32 == 0 && SEG_RPT >= 1 && SEG_RPT <= 2, "one or two rows per thread");
constexpr int SG_OFF = 0;                         // [33] row of each packed chunk's first byte (+ total)
constexpr int SG_ITEM = 33;                       // [32] its index on the long list
constexpr int SG_HARD = 65;                       // [SEG_PAD / 32 + 2 zero words] bit r: nothing spans the boundary after row r
constexpr int SG_LONG = SG_HARD + SEG_PAD / 32 + 2;   // [32] segments of 17..64 bytes: first row | length << 16
constexpr int SG_CTL = SG_LONG + 32;              // [13] packed chunks, long segments, chunks to leave (bit = packing slot), chunks tried
                                                  //      (bit = list index), cut, chunks appended, mid segments, x segments, any left, heavy rows, short segments
constexpr int SG_WS = SG_CTL + 13;                // [8] running-maximum totals per wavefront and half
constexpr int SG_MID = SG_WS + 8;                 // [64] segments of 9..16 bytes
constexpr int SG_XSEG = SG_MID + 64;              // [8] segments of 65 .. 64 XNPL bytes
constexpr int SG_SBITS = SG_XSEG + 8;             // [SEG_PAD / 32] bit r: row r is the first row of a packed chunk
constexpr int SG_LIST = SG_SBITS + SEG_PAD / 32; // [SEG_PAD / 2] u16 lists, one after the other: heavy rows; then segments of up to 8 bytes (row | len - 1 << 9)
constexpr int SG_SPRE = SG_LIST + SEG_PAD / 2;     // [SEG_PAD / 32] chunk starts in the bitmap words before this one
constexpr int SG_WORDS = SG_SPRE + SEG_PAD / 32;
// sid[SEG_ROWS] (the caller's): each row's byte | its chunk's packing slot << 8 | min(longest token that starts there, 255) << 24 (255: to the
// chunk's end); bid_tab[256]: DeviceTables::byte_id in LDS (the id of a row's byte is needed where a lone byte stays a token)
template <int XNPL, class EmitG>
__device__ __forceinline__ uint32_t bpe_tail_segments(const DeviceTables& T, const Batch& b, uint32_t* s_lq, uint32_t nl,
                                                      uint32_t* slab, uint32_t* scr, uint32_t* sid, const uint32_t* bid_tab,
                                                      const uint8_t* win_txt, int64_t win_lo, int64_t win_hi, EmitG emit_g) {
    const int tid = tidx(), lane = tid & 63, wv = tid >> 6;
    uint32_t* const off = scr + SG_OFF;
    uint32_t* const item = scr + SG_ITEM;
    uint32_t* const hard = scr + SG_HARD;
    uint32_t* const lseg = scr + SG_LONG;
    uint32_t* const mseg = scr + SG_MID;
    uint32_t* const xseg = scr + SG_XSEG;
    uint32_t* const ctl = scr + SG_CTL;
    uint32_t* const wsum = scr + SG_WS;
    uint16_t* const list = reinterpret_cast<uint16_t*>(scr + SG_LIST);
    auto hbits = [&](int pos) {                              // 32 boundary bits from row `pos` on
        const int w = pos >> 5, sh = pos & 31;
        return (hard[w] >> sh) | (sh ? hard[w + 1] << (32 - sh) : 0u);
    };
    uint32_t* const sbits = scr + SG_SBITS;
    uint32_t* const spre = scr + SG_SPRE;
    auto chunk_of = [&](int row) {                           // packing slot of the chunk that owns a row: chunk starts at or below it, minus one
        // (two LDS reads and a popcount; up to round 4 a linear search of off[] -- a chain of dependent LDS reads that was, with two rows
        //  per thread, a tenth of a pass's instructions)
        const int rw = row >> 5;
        return spre[rw] + (uint32_t)__popc(sbits[rw] & (0xFFFFFFFFu >> (31 - (row & 31)))) - 1u;
    };
    auto first_byte_of = [&](int row) {                      // global position of a row's byte (its chunk's packing slot: sid, since step 1)
        const uint32_t k = (sid[row] >> 8) & 31u;
        return s_lq[2 * item[k]] + ((uint32_t)row - off[k]);
    };
    // the eight text bytes at a row (w1 only on request): from the tile's window where it is staged, else from memory
    auto row_text = [&](uint64_t g, bool want1, uint32_t& w0, uint32_t& w1) {
        const uint64_t B = b.n_bytes;
        w0 = 0; w1 = 0;
        if ((int64_t)g >= win_lo && (int64_t)g + 8 <= win_hi) {            // staged with the tile's window: no trip to HBM
            const LdsAcc wt{nullptr, win_txt};
            const int q = (int)((int64_t)g - win_lo);
            w0 = wt.load32(q);
            if (want1) w1 = wt.load32(q + 4);
        } else if (g + 8 <= B) { __builtin_memcpy(&w0, b.text + g, 4); if (want1) __builtin_memcpy(&w1, b.text + g + 4, 4); }
        else for (int q = 0; q < 8; q++) if (g + q < B) (q < 4 ? w0 : w1) |= (uint32_t)b.text[g + q] << (8 * (q & 3));
    };
    // (the list's length lives in ctl[12] from here on: as a value it was kept -- in scratch -- across the whole pass for its three uses)
    {
        uint32_t* c0 = scr + SG_CTL;                         // (an address of its own: the one LDS base register that served this store and the
        asm volatile("" : "+v"(c0));                         //  reads at the function's end was kept -- spilled -- across everything in between)
        if (tid == 0) { c0[3] = 0; c0[5] = 0; c0[12] = nl; hard[SEG_PAD / 32] = 0; hard[SEG_PAD / 32 + 1] = 0; }
    }
#ifndef SPL_TT_MASK
#define SPL_TT_MASK 31u
#endif
#ifdef SPL_STAMP_TAIL      /* profiling: wall clock of a pass's steps as thread 0 sees them, summed over all workgroups and passes of
                              a launch (tools/dev/gpu_tail_steps.py; the atomics inflate every step) */
    unsigned long long tt_prev = 0;
#define TT(k) do { if (b.dbg && tid == 0 && (blockIdx.x & SPL_TT_MASK) == 0u) { const unsigned long long tt_now = wall_clock64(); \
                   if ((k) >= 0) atomicAdd(&b.dbg[16 + 4 * (SPL_DEBUG_BLOCKS - 32) + (k)], tt_now - tt_prev); tt_prev = tt_now; } } while (0)
#define TT_COUNT(k, v) do { if (b.dbg && tid == 0 && (blockIdx.x & SPL_TT_MASK) == 0u) atomicAdd(&b.dbg[16 + 4 * (SPL_DEBUG_BLOCKS - 32) + (k)], (unsigned long long)(v)); } while (0)
#else
#define TT(k) do { } while (0)
#define TT_COUNT(k, v) do { } while (0)
#endif
    for (;;) {
        TT(-1);
        __syncthreads();
        if (wv == 0) {
            // pack: the untried chunks in list order while they fit (lane = list index); a chunk beyond
            // SEG_ROWS goes alone -- its first SEG_ROWS bytes -- once it is the first one left
            const uint32_t tried = ctl[3], nlv = ctl[12];
            const uint32_t n = (lane < 32 && (uint32_t)lane < nlv) ? s_lq[2 * lane + 1] : 0u;
            const bool elig = n >= 2u && !((tried >> (lane & 31)) & 1u);
            const unsigned long long em = __ballot(elig);
            bool take = false;
            uint32_t offv = 0, cut = 0;
            if (em) {
                const int first = __builtin_ctzll(em);
                if ((uint32_t)__builtin_amdgcn_readlane((int)n, first) > (uint32_t)SEG_ROWS) { cut = 1; take = lane == first; }
                else {
                    const uint32_t v = (elig && n <= (uint32_t)SEG_ROWS) ? n : 0u;
                    const uint32_t x = wave_scan_incl(v);
                    take = v != 0 && x <= (uint32_t)SEG_ROWS;
                    offv = x - v;
                }
            }
            const unsigned long long tm = __ballot(take);
            const uint32_t k = mbcnt64(tm), nk = (uint32_t)__popcll(tm);
            if (lane < SEG_PAD / 32) sbits[lane] = 0u;
            wave_lds_sync();
            if (take) { off[k] = offv; item[k] = (uint32_t)lane; atomicOr(&sbits[offv >> 5], 1u << (offv & 31)); }
            wave_lds_sync();
            {
                const uint32_t sc = lane < SEG_PAD / 32 ? (uint32_t)__popc(sbits[lane]) : 0u;
                const uint32_t sx = wave_scan_incl(sc);
                if (lane < SEG_PAD / 32) spre[lane] = sx - sc;
            }
            const uint32_t endv = offv + (n < (uint32_t)SEG_ROWS ? n : (uint32_t)SEG_ROWS);
            const uint32_t total = tm ? (uint32_t)__builtin_amdgcn_readlane((int)endv, 63 - __builtin_clzll(tm)) : 0u;
            // (for the caller: is ANY chunk of two bytes or more on the list -- packed now, left by an earlier pass, or a long
            //  segment set aside behind its end?  Nearly always not once the last pass is done, and the caller then skips
            //  its three node-list loops and their barriers.)
            const uint32_t nl_now = nlv + ctl[5] < (uint32_t)DIRECT_LQCAP ? nlv + ctl[5] : (uint32_t)DIRECT_LQCAP;
            const unsigned long long any_m = __ballot(lane < 32 && (uint32_t)lane < nl_now && s_lq[2 * lane + 1] >= 2u);
            const unsigned long long left_m = __ballot(lane < 32 && (uint32_t)lane < nl_now && s_lq[2 * lane + 1] >= 2u && !take);
            if (lane == 0) {
                off[nk] = total;
                ctl[0] = nk; ctl[1] = 0; ctl[2] = 0; ctl[3] = tried | (uint32_t)tm; ctl[4] = cut; ctl[6] = 0; ctl[7] = 0;
                ctl[8] = any_m != 0ull; ctl[9] = 0; ctl[10] = 0;
                // this pass takes everything that is left (nothing untried stays behind, nothing an earlier pass gave up on): if it
                // finishes all of it -- the usual case -- the closing empty pass (a pack round and three barriers) is not needed
                ctl[11] = (left_m == 0ull && !cut) ? 0x80000000u | ctl[5] : 0u;
            }
        }
        __syncthreads();
        const uint32_t nk = ctl[0];
        if (nk == 0) break;
        TT(0);                                               // pack (and the wait for the previous pass's stragglers)
        const uint32_t total = off[nk];
        TT_COUNT(12, 1); TT_COUNT(13, total);
        const bool cut = ctl[4] != 0;                        // the (one) chunk continues beyond the rows
        // ---- step 1, per row: the prefix entry; light rows are finished here, heavy rows go on the list -------------
        // sid[r] = the row's byte | its chunk's packing slot << 8 | (longest token that starts there, 255: to the chunk's end) << 24
        {
            uint32_t rk[SEG_RPT], rw0[SEG_RPT], rcap[SEG_RPT], rci[SEG_RPT];
            PfxEnt rpe[SEG_RPT];
#pragma unroll
            for (int h = 0; h < SEG_RPT; h++) {                  // (the loads of all of a thread's rows in flight together)
                const int r = tid + h * NT;
                rk[h] = 0; rw0[h] = 0; rcap[h] = 0; rci[h] = 0; rpe[h] = PfxEnt{0u, SPL_NO_RANK};
                if ((uint32_t)r < total && SPL_TAIL_CUT < 3) {
                    const uint32_t k = chunk_of(r);
                    const uint32_t ci = (uint32_t)r - off[k], cn = s_lq[2 * item[k] + 1];
                    const uint64_t g = (uint64_t)s_lq[2 * item[k]] + ci;
                    uint32_t w0, w1;
                    row_text(g, false, w0, w1);
                    rk[h] = k; rw0[h] = w0; rcap[h] = cn - ci; rci[h] = ci;
                    rpe[h] = T.pfx[w0 & 0xFFFFu];
                }
            }
#pragma unroll
            for (int h = 0; h < SEG_RPT; h++) {
                const int r = tid + h * NT;
                const bool own = (uint32_t)r < total;
                bool heavy = false;
                if (own && SPL_TAIL_CUT >= 3) { sid[r] = 1u << 24; uint32_t* const row = slab + r * SUB_W; for (int c = 0; c < SUB_W; c++) row[c] = SPL_NO_RANK; }
                if (own && SPL_TAIL_CUT < 3) {
                    const uint32_t w0 = rw0[h], cap = rcap[h];
                    const PfxEnt pe = rpe[h];
                    // lengths 3 .. min(cap, 8) that exist behind these two bytes, and "longer" if the chunk has room for it
                    const uint32_t maxlen = cap < (uint32_t)SUB_LMAX ? cap : (uint32_t)SUB_LMAX;
                    const uint32_t allow = maxlen >= 2u ? ((((1u << (maxlen - 1u)) - 1u) & 0x7Eu) | (cap > (uint32_t)SUB_LMAX ? 0x80u : 0u)) : 0u;
                    heavy = rci[h] == 0u || (w0 & 0xC0u) != 0x80u || cap < 2u || (pe.lm & allow) != 0u;
                    uint32_t* const row = slab + r * SUB_W;
                    const uint32_t sw = (w0 & 0xFFu) | (rk[h] << 8);
                    if (!heavy) {
                        row[0] = pe.id2;
#pragma unroll
                        for (int c = 1; c < SUB_W; c++) row[c] = SPL_NO_RANK;
                        sid[r] = sw | ((pe.id2 != SPL_NO_RANK ? 2u : 1u) << 24);
                    } else {                                     // the head, for the lane that tabulates this row in step 2
                        row[0] = pe.lm; row[1] = pe.id2;
                        sid[r] = sw;
                    }
                }
                const unsigned long long hm = __ballot(heavy);
                uint32_t hb = 0;
                if (lane == 0 && hm) hb = atomicAdd(&ctl[9], (uint32_t)__popcll(hm));
                hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
                if (heavy) list[hb + mbcnt64(hm)] = (uint16_t)r;
            }
        }
        __syncthreads();
        TT(1);                                               // step 1
        TT_COUNT(14, ctl[9]);
        // ---- step 2, per heavy row off the list: the full tabulation (ONE call site of the probe code) -------------------
        {
            const uint32_t nheavy = SPL_TAIL_CUT >= 2 ? 0u : ctl[9];
            if (SPL_TAIL_CUT >= 2) for (uint32_t i = (uint32_t)tid; i < ctl[9]; i += NT) { const int r = (int)list[i]; sid[r] = 1u << 24; uint32_t* const row = slab + r * SUB_W; for (int c = 0; c < SUB_W; c++) row[c] = SPL_NO_RANK; }
#pragma nounroll
            for (uint32_t i0 = 0; i0 < nheavy; i0 += NT) {
                if (i0 + (uint32_t)(tid & ~63) >= nheavy) break;             // (wave-uniform: this wavefront has no row of the round)
                const uint32_t i = i0 + (uint32_t)tid;
                const bool own = i < nheavy;
                const int r = own ? (int)list[i] : 0;
                int maxlen = 0, cap = 0;
                uint32_t w0 = 0, w1 = 0;
                RowHead rh{0u, SPL_NO_RANK, 0u, 0u};                 // (as row_head() makes it, from the words step 1 left in the row's cells)
                uint32_t sw0 = 0;
                if (own) {
                    sw0 = sid[r];
                    const uint32_t k = (sw0 >> 8) & 31u;
                    const uint32_t ci = (uint32_t)r - off[k], cn = s_lq[2 * item[k] + 1];
                    const uint64_t g = (uint64_t)s_lq[2 * item[k]] + ci;
                    cap = (int)(cn - ci);
                    maxlen = cap < SUB_LMAX ? cap : SUB_LMAX;
                    row_text(g, true, w0, w1);
                    const uint32_t* const st = slab + r * SUB_W;
                    // (the filter entry is fetched HERE, for heavy rows only: fetched in step 1 for every row that is heavy by its byte alone --
                    //  one dependent round trip less -- the pass was 0.8 us SLOWER: it is the count of scattered loads that costs, not their latency)
                    const uint32_t plm = st[0], f = maxlen >= 4 ? (uint32_t)T.filt4[hash_f4(w0) >> T.filt4_shift] : 0u;
                    const uint32_t f4 = SPL_ROW_FILTER ? f & 0x3Fu : (maxlen >= 4 ? 0x3Fu : 0u);
                    rh.lm = plm & 0xFFu & (0x03u | (f4 << 2));
                    rh.id2 = st[1];
                    rh.tsalt = plm >> 16;
                    rh.fsalt = f >> SPL_F4_MASK_BITS;
                }
                int ml = 1;
                P8Bucket e8{0u, 0u};
                const bool want8 = maxlen >= 2 && cap > SUB_LMAX && (rh.lm & 0x80u);
                if (want8) e8 = T.p8_tab[hash_p8(w0, w1) & T.p8_mask];
                uint32_t rr[7];
                row_fill(T, rh, w0, w1, maxlen, rr);
                if (own) {
                    uint32_t* const row = slab + r * SUB_W;
#pragma unroll
                    for (int c = 0; c < SUB_W; c++) {
                        row[c] = rr[c];                              // (maxlen < 2: all empty)
                        ml = (rr[c] != SPL_NO_RANK && maxlen >= c + 2) ? c + 2 : ml;
                    }
                    if (want8) {
                        const int l8 = (int)p8_match(e8.a, e8.b, p8_tag(w0, w1));
                        if (l8) ml = (l8 == 255 || l8 > cap) ? cap : l8;
                    }
                    sid[r] = (sw0 & 0x1FFFu) | ((uint32_t)(ml >= cap ? (cap < 255 ? cap : 255) : ml) << 24);   // (ml < 255 unless it is the chunk's end)
                }
            }
        }
        __syncthreads();
        TT(2);                                               // step 2
        // ---- boundaries: a running maximum of "last row covered by a token that starts at or before this row" ----------------
        {
            uint32_t cov[SEG_RPT];
            bool ownr[SEG_RPT];
#pragma unroll
            for (int h = 0; h < SEG_RPT; h++) {
                const int r = tid + h * NT;
                ownr[h] = (uint32_t)r < total;
                uint32_t reach = 0;
                if (ownr[h]) {
                    const uint32_t sw = sid[r], mlb = sw >> 24;
                    if (mlb == 255u) {                               // to the end of the row's chunk
                        const uint32_t k = (sw >> 8) & 31u;
                        const uint32_t cn = s_lq[2 * item[k] + 1];
                        const uint32_t endr = off[k] + (cn < (uint32_t)SEG_ROWS ? cn : (uint32_t)SEG_ROWS);
                        reach = (endr < total ? endr : total) - 1u;
                        if (cut) reach = (uint32_t)SEG_ROWS;          // (the one cut chunk: beyond the rows)
                    } else reach = (uint32_t)r + mlb - 1u;
                }
                cov[h] = wave_scan_max(reach);
                if (lane == 63) wsum[h * 4 + wv] = cov[h];
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < SEG_RPT; h++) {
                uint32_t c = cov[h];
                for (int k = 0; k < wv; k++) c = wsum[h * 4 + k] > c ? wsum[h * 4 + k] : c;
                if (h == 1) for (int k = 0; k < NT / 64; k++) c = wsum[k] > c ? wsum[k] : c;
                const unsigned long long hb = __ballot(ownr[h] && c == (uint32_t)(tid + h * NT));
                if (lane == 0) { hard[h * (NT / 32) + 2 * wv] = (uint32_t)hb; hard[h * (NT / 32) + 2 * wv + 1] = (uint32_t)(hb >> 32); }
            }
        }
        __syncthreads();
        TT(3);                                               // boundaries
        // a cut chunk: only what lies before the last boundary among the rows is complete; the rest
        // goes back on the list as a chunk of its own (nothing spans that boundary)
        uint32_t rows = total;
        if (cut) {
            int last = -1;
            for (int w = SEG_PAD / 32 - 1; w >= 0 && last < 0; w--) if (hard[w]) last = 32 * w + 31 - __clz((int)hard[w]);
            rows = (uint32_t)(last + 1);
            if (last < 0 && tid == 0) ctl[2] = 1u;           // no boundary at all: left to the node-list loops
        }
        // ---- every row that starts a segment: up to 8 bytes go on the short list (one lane each below), longer ones to a
        //      group of 16 lanes, a wavefront, or back on the list
#pragma unroll
        for (int h = 0; h < SEG_RPT; h++) {
            const int r = tid + h * NT;
            bool shortseg = false;
            uint32_t slen = 0;
            if ((uint32_t)r < rows && (r == 0 || ((hard[(r - 1) >> 5] >> ((r - 1) & 31)) & 1u))) {
                const uint32_t h0 = hbits(r);                    // the first boundary at or after the start ends the segment
                if (h0 & 0xFFu) { shortseg = true; slen = (uint32_t)__ffs((int)h0); }
                else if (h0 & 0xFFFFu) {
                    mseg[atomicAdd(&ctl[6], 1u)] = (uint32_t)r | (uint32_t)__ffs((int)h0) << 16;
                } else {
                    const uint32_t h1 = hbits(r + 32);
                    const uint32_t l2 = h0 ? (uint32_t)__ffs((int)h0) : h1 ? 32u + (uint32_t)__ffs((int)h1) : 65u;
                    if (l2 <= 64u) lseg[atomicAdd(&ctl[1], 1u)] = (uint32_t)r | l2 << 16;
                    else {
                        int q = r + 64;
                        uint32_t hq;
                        while ((hq = hbits(q)) == 0) q += 32;    // (the last row of a chunk is a boundary)
                        const uint32_t l3 = (uint32_t)(q - r) + (uint32_t)__ffs((int)hq);
                        const uint32_t qi = ctl[12] + (l3 <= 64u * XNPL ? 0u : atomicAdd(&ctl[5], 1u));
                        if (l3 <= 64u * XNPL) xseg[atomicAdd(&ctl[7], 1u)] = (uint32_t)r | l3 << 16;   // a wavefront, several nodes per lane
                        else if (qi < (uint32_t)DIRECT_LQCAP) {  // longer still: a chunk of its own for the loops below
                            s_lq[2 * qi] = first_byte_of(r);
                            s_lq[2 * qi + 1] = l3;
                            atomicOr(&ctl[3], 1u << qi);         // (not to be packed again)
                        } else {                                 // no room: the whole chunk stays on the list
                            atomicOr(&ctl[2], 1u << ((sid[r] >> 8) & 31u));
                        }
                    }
                }
            }
            const unsigned long long sm = __ballot(shortseg);
            uint32_t sb = 0;
            if (lane == 0 && sm) sb = atomicAdd(&ctl[10], (uint32_t)__popcll(sm));
            sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)sb);
            if (shortseg) list[sb + mbcnt64(sm)] = (uint16_t)((uint32_t)r | (slen - 1u) << 9);
        }
        __syncthreads();
        TT(4);                                               // classification of the segments
        TT_COUNT(15, ctl[10]);
        // ---- segments of up to 8 bytes, one lane each off the dense list: all their spans are in the table ----------------
        {
            const uint32_t nshort = SPL_TAIL_CUT >= 1 ? 0u : ctl[10];
#pragma nounroll
            for (uint32_t i = (uint32_t)tid; i < nshort; i += NT) {
                const uint32_t ent = list[i];
                const int r = (int)(ent & 511u), len = (int)(ent >> 9) + 1;
                const uint32_t gpos = first_byte_of(r);
                const uint32_t* const cells = slab + r * SUB_W;      // node x of the segment: cells + x * SUB_W
                uint32_t alive = (1u << len) - 1u;
                for (;;) {                                   // bpe.rs:118-190 on at most 8 nodes in a bit mask
                    uint32_t best = SPL_NO_RANK, kill = 0, m = alive;
                    int x = __ffs((int)m) - 1;
                    m &= m - 1u;
                    while (m) {
                        const int y = __ffs((int)m) - 1;
                        const uint32_t m2 = m & (m - 1u);
                        const int e2 = m2 ? __ffs((int)m2) - 1 : len;
                        const uint32_t rk = cells[x * SUB_W + (e2 - x - 2)];
                        if (rk < best) { best = rk; kill = 1u << y; }
                        x = y;
                        m = m2;
                    }
                    if (!kill) break;
                    alive &= ~kill;
                }
                for (uint32_t m = alive; m;) {
                    const int x = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const int e2 = m ? __ffs((int)m) - 1 : len;
                    emit_g(gpos + (uint32_t)x, e2 - x == 1 ? bid_tab[sid[r + x] & 0xFFu] : cells[x * SUB_W + (e2 - x - 2)]);
                }
            }
        }
        TT(5);                                               // segments of up to 8 bytes (thread 0's wavefront)
        // ---- segments of 9..16 bytes: a group of 16 lanes each ------------------------------------------
        {
            const int gi = tid >> 4, gl = tid & 15;
            const uint32_t nmid = SPL_TAIL_CUT >= 1 ? 0u : ctl[6];
            for (uint32_t q0 = 0; q0 < nmid; q0 += NT / 16) {
                const uint32_t q = q0 + (uint32_t)gi;
                const int s0 = q < nmid ? (int)(mseg[q] & 0xFFFFu) : 0, len = q < nmid ? (int)(mseg[q] >> 16) : 0;
                const uint32_t gpos = len ? first_byte_of(s0) : 0u;
                const bool gown = gl < len;
                // (far_max: the longest token that can start at the lane's byte -- the row's own bound, from its cells and the p8 table --
                //  so that a span beyond it ranks "none" without a trip to the pair table; up to round 4 every span of more than 8
                //  bytes went there, a dependent round trip per merge round: profiles/r05_tail_cost.txt)
                const uint32_t sw = gown ? sid[s0 + gl] : 0u;
                group16_merge(T, slab + (gown ? s0 + gl : 0) * SUB_W, gown ? bid_tab[sw & 0xFFu] : SPL_DEAD, len,
                              (sw >> 24) == 255u ? FAR_UNBOUNDED : (int)(sw >> 24),
                              [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
            }
        }
        TT(6);                                               // 9..16 (thread 0's wavefront)
        TT_COUNT(16, ctl[6]); TT_COUNT(17, ctl[1]);
        // ---- segments of 17..64 bytes: one wavefront each ------------------------------------------
        for (uint32_t q = (uint32_t)wv; q < (SPL_TAIL_CUT >= 1 ? 0u : ctl[1]); q += NT / 64) {
            const int s0 = (int)(lseg[q] & 0xFFFFu), len = (int)(lseg[q] >> 16);
            const uint32_t gpos = first_byte_of(s0);
            const bool lown = lane < len;
            const uint32_t* const lrow = slab + (lown ? s0 + lane : 0) * SUB_W;
            const uint32_t sw = lown ? sid[s0 + lane] : 0u;
            wave64_merge(T, lrow, len >= 64 ? ~0ull : ((1ull << len) - 1ull), len, lane + 1 < len ? lrow[0] : SPL_NO_RANK,
                         lown ? bid_tab[sw & 0xFFu] : SPL_DEAD, (sw >> 24) == 255u ? FAR_UNBOUNDED : (int)(sw >> 24),
                         [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
        }
        for (uint32_t q = (uint32_t)wv; q < (SPL_TAIL_CUT >= 1 ? 0u : ctl[7]); q += NT / 64) {          // 65 .. 64 XNPL bytes
            const int s0 = (int)(xseg[q] & 0xFFFFu), len = (int)(xseg[q] >> 16);
            const uint32_t gpos = first_byte_of(s0);
            wave_tab_merge<XNPL>(T, len, slab + s0 * SUB_W, [&](int i) { return bid_tab[sid[s0 + i] & 0xFFu]; },
                                 [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
        }
        __syncthreads();
        TT(7);                                               // 17..64, 65.. and the wait for the other wavefronts
        if ((uint32_t)tid < nk && !((ctl[2] >> tid) & 1u)) {
            if (!cut) s_lq[2 * item[tid] + 1] = 0;          // done: off the list
            else {                                           // the rest of a cut chunk: to be packed again
                const uint32_t rest = s_lq[2 * item[tid] + 1] - rows, at = s_lq[2 * item[tid]] + rows;
                if (rest == 1u) emit_g(at, T.byte_id[b.text[at]]);       // a lone last byte is its own token
                s_lq[2 * item[tid]] = at;
                s_lq[2 * item[tid] + 1] = rest == 1u ? 0u : rest;
                ctl[3] &= ~(1u << item[tid]);
            }
        }
        // (ctl[2], ctl[5], ctl[11]: final since the barrier behind the classification of the segments -- the same in every thread)
        if (SPL_TAIL_SKIP_EMPTY && (ctl[11] >> 31) && ctl[2] == 0u && ctl[5] == (ctl[11] & 0x7FFFFFFFu)) { __syncthreads(); return 0u; }
    }
    __syncthreads();
#undef TT
#undef TT_COUNT
    const uint32_t* c1 = scr + SG_CTL;
    asm volatile("" : "+v"(c1));
    const uint32_t nl2 = c1[12] + c1[5];                    // the list grew by the segments set aside
    if (SPL_TAIL_SKIP_EMPTY && !c1[8]) return 0u;           // (as of the last, empty pass: nothing of two bytes or more is left)
    return nl2 < (uint32_t)DIRECT_LQCAP ? nl2 : (uint32_t)DIRECT_LQCAP;
}

}  // namespace spl
