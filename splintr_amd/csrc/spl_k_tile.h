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
#define SPL_SEG_ROWS NT
#endif
#ifndef SPL_TAIL_SKIP_EMPTY
#define SPL_TAIL_SKIP_EMPTY 1    /* 1: no chunk left behind the segment passes (nearly always): none of the caller's three node-list loops, nor their barriers */
#endif
constexpr int SEG_ROWS = SPL_SEG_ROWS;     // rows of a pass: one per thread ((A/B) 128: twice the passes -- what a pass costs)
constexpr int SG_OFF = 0;        // [33] row of each packed chunk's first byte (+ total)
constexpr int SG_ITEM = 33;      // [32] its index on the long list
constexpr int SG_HARD = 65;      // [8 + 2 zero words] bit r: nothing spans the boundary after row r
constexpr int SG_LONG = 75;      // [16] segments of 17..64 bytes: first row | length << 16
constexpr int SG_CTL = 91;       // [9] packed chunks, long segments, chunks to leave (bit = packing slot), chunks tried
                                 //     (bit = list index), cut, chunks appended, mid segments
constexpr int SG_ID = 100;       // [SEG_ROWS] id of each row's byte
constexpr int SG_MID = SG_ID + SEG_ROWS;     // [32] segments of 9..16 bytes
constexpr int SG_XSEG = SG_MID + 32;          // [4] segments of 65 .. 64 XNPL bytes
constexpr int SG_SBITS = SG_XSEG + 4;         // [8] bit r: row r is the first row of a packed chunk
constexpr int SG_WORDS = SG_SBITS + 8;
template <int XNPL, class EmitG>
__device__ __forceinline__ uint32_t bpe_tail_segments(const DeviceTables& T, const Batch& b, uint32_t* s_lq, uint32_t nl,
                                                      uint32_t* slab, uint32_t* scr, uint32_t* s_wsum4, const uint8_t* win_txt,
                                                      int64_t win_lo, int64_t win_hi, EmitG emit_g) {
    const int tid = tidx(), lane = tid & 63, wv = tid >> 6;
    uint32_t* const off = scr + SG_OFF;
    uint32_t* const item = scr + SG_ITEM;
    uint32_t* const hard = scr + SG_HARD;
    uint32_t* const lseg = scr + SG_LONG;
    uint32_t* const mseg = scr + SG_MID;
    uint32_t* const xseg = scr + SG_XSEG;
    uint32_t* const ctl = scr + SG_CTL;
    uint32_t* const sid = scr + SG_ID;
    auto hbits = [&](int pos) {                              // 32 boundary bits from row `pos` on
        const int w = pos >> 5, sh = pos & 31;
        return (hard[w] >> sh) | (sh ? hard[w + 1] << (32 - sh) : 0u);
    };
    uint32_t* const sbits = scr + SG_SBITS;
    auto chunk_of = [&](int row) {                           // packing slot of the chunk that owns a row:
#if !SPL_TILE_MISS_LIST
        // a handful of packed chunks (long chunks of a tile): a linear search beats the popcounts below
        // (X1 121 -> 116 us, C3 498 -> 486 us); the bitmap is for the many-chunk packing of SPL_TILE_MISS_LIST
        { uint32_t kk = 0; while (off[kk + 1] <= (uint32_t)row) kk++; return kk; }
#endif
        uint32_t k = 0;                                      // chunk starts at or below it, minus one
        const int rw = row >> 5;
#pragma unroll
        for (int w = 0; w < SEG_ROWS / 32; w++) {
            const uint32_t x = sbits[w];
            k += w < rw ? __popc(x) : w == rw ? __popc(x & (0xFFFFFFFFu >> (31 - (row & 31)))) : 0u;
        }
        return k - 1u;
    };
    auto first_byte_of = [&](int row) {                      // global position of a row's byte
        const uint32_t k = chunk_of(row);
        return s_lq[2 * item[k]] + ((uint32_t)row - off[k]);
    };
    if (tid == 0) { ctl[3] = 0; ctl[5] = 0; hard[8] = 0; hard[9] = 0; }
#ifdef SPL_STAMP_TAIL      /* profiling: wall clock of a pass's steps as thread 0 sees them, summed over all workgroups and passes of
                              a launch of at most ~4000 tiles (tools/dev/gpu_tail_steps.py; the atomics inflate every step) */
    unsigned long long tt_prev = 0;
#define TT(k) do { if (b.dbg && tid == 0) { const unsigned long long tt_now = wall_clock64(); \
                   if ((k) >= 0) atomicAdd(&b.dbg[16 + 4 * (SPL_DEBUG_BLOCKS - 32) + (k)], tt_now - tt_prev); tt_prev = tt_now; } } while (0)
#define TT_COUNT() do { if (b.dbg && tid == 0) { atomicAdd(&b.dbg[16 + 4 * (SPL_DEBUG_BLOCKS - 32) + 7], 1ull); \
                        atomicAdd(&b.dbg[16 + 4 * (SPL_DEBUG_BLOCKS - 32) + 6], (unsigned long long)total); } } while (0)
#else
#define TT(k) do { } while (0)
#define TT_COUNT() do { } while (0)
#endif
    for (;;) {
        TT(-1);
        __syncthreads();
        if (wv == 0) {
            // pack: the untried chunks in list order while they fit (lane = list index); a chunk beyond
            // SEG_ROWS goes alone -- its first SEG_ROWS bytes -- once it is the first one left
            const uint32_t tried = ctl[3];
            const uint32_t n = (lane < 32 && (uint32_t)lane < nl) ? s_lq[2 * lane + 1] : 0u;
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
            if (lane < SEG_ROWS / 32) sbits[lane] = 0u;
            wave_lds_sync();
            if (take) { off[k] = offv; item[k] = (uint32_t)lane; atomicOr(&sbits[offv >> 5], 1u << (offv & 31)); }
            const uint32_t endv = offv + (n < (uint32_t)SEG_ROWS ? n : (uint32_t)SEG_ROWS);
            const uint32_t total = tm ? (uint32_t)__builtin_amdgcn_readlane((int)endv, 63 - __builtin_clzll(tm)) : 0u;
            // (for the caller: is ANY chunk of two bytes or more on the list -- packed now, left by an earlier pass, or a long
            //  segment set aside behind its end?  Nearly always not once the last pass is done, and the caller then skips
            //  its three node-list loops and their barriers.)
            const uint32_t nl_now = nl + ctl[5] < (uint32_t)DIRECT_LQCAP ? nl + ctl[5] : (uint32_t)DIRECT_LQCAP;
            const unsigned long long any_m = __ballot(lane < 32 && (uint32_t)lane < nl_now && s_lq[2 * lane + 1] >= 2u);
            if (lane == 0) {
                off[nk] = total;
                ctl[0] = nk; ctl[1] = 0; ctl[2] = 0; ctl[3] = tried | (uint32_t)tm; ctl[4] = cut; ctl[6] = 0; ctl[7] = 0;
                ctl[8] = any_m != 0ull;
            }
        }
        __syncthreads();
        const uint32_t nk = ctl[0];
        if (nk == 0) break;
        TT(0);                                               // pack (and the wait for the previous pass's stragglers)
        const uint32_t total = off[nk];
        // ---- table rows, longest token per row, boundaries ------------------------------------------
        const bool own = (uint32_t)tid < total;
        const bool cut = ctl[4] != 0;                        // the (one) chunk continues beyond the rows
        int maxlen = 0, cap = 0;                             // cap: bytes left in the row's chunk
        uint32_t w0 = 0, w1 = 0, bid = SPL_DEAD, lm = 0;
        uint32_t my_gpos = 0;                                // global position of this row's byte (kept: the row's segment starts there)
        if (own) {
            const uint32_t k = chunk_of(tid);
            const uint32_t ci = (uint32_t)tid - off[k], cn = s_lq[2 * item[k] + 1];
            const uint64_t g = (uint64_t)s_lq[2 * item[k]] + ci, B = b.n_bytes;
            my_gpos = (uint32_t)g;
            cap = (int)(cn - ci);
            maxlen = cn - ci < (uint32_t)SUB_LMAX ? (int)(cn - ci) : SUB_LMAX;
            if ((int64_t)g >= win_lo && (int64_t)g + 8 <= win_hi) {        // staged with the tile's window: no trip to HBM
                const LdsAcc wt{nullptr, win_txt};
                const int q = (int)((int64_t)g - win_lo);
                w0 = wt.load32(q); w1 = wt.load32(q + 4);
            } else if (g + 8 <= B) { __builtin_memcpy(&w0, b.text + g, 4); __builtin_memcpy(&w1, b.text + g + 4, 4); }
            else for (int q = 0; q < 8; q++) if (g + q < B) (q < 4 ? w0 : w1) |= (uint32_t)b.text[g + q] << (8 * (q & 3));
            bid = T.byte_id[w0 & 0xFFu];
        }
        RowHead rh = row_head(T, own, w0, maxlen);           // which token lengths exist at all behind these bytes
#if defined(SPL_TAIL_CUT)
        if (SPL_TAIL_CUT >= 2) { rh.lm = 0; maxlen = maxlen < 2 ? maxlen : 2; }
#endif
        lm = rh.lm;
        uint32_t* const row = slab + tid * SUB_W;
        int ml = 1;
        {
            // all six lengths and the p8 bucket in ONE round trip (row_fill: one entry per probe); up to round 3 two batches
            // of buckets -- with the lengths 5 / 6 swapped between them for rows that start a three-byte character
            P8Bucket e8{0u, 0u};
            const bool want8 = maxlen >= 2 && cap > SUB_LMAX && (lm & 0x80u);
            if (want8) e8 = T.p8_tab[hash_p8(w0, w1) & T.p8_mask];
            uint32_t r[7];
            row_fill(T, rh, w0, w1, maxlen, r);
            if (maxlen >= 2) {
#pragma unroll
                for (int k = 0; k < SUB_W; k++) {
                    row[k] = r[k];
                    ml = (r[k] != SPL_NO_RANK && maxlen >= k + 2) ? k + 2 : ml;
                }
            }
            sid[tid] = bid;
            if (want8) {
                const int l8 = (int)p8_match(e8.a, e8.b, p8_tag(w0, w1));
                if (l8) ml = (l8 == 255 || l8 > cap) ? cap : l8;
            }
        }
        TT(1);                                               // rows filled (two dependent round trips: row head, entries)
        {
            uint32_t cover = wave_scan_max(own ? (uint32_t)(tid + ml - 1) : 0u);
            if (lane == 63) s_wsum4[wv] = cover;
            __syncthreads();
            for (int k = 0; k < wv; k++) cover = s_wsum4[k] > cover ? s_wsum4[k] : cover;
            const unsigned long long hb = __ballot(own && cover == (uint32_t)tid);
            if (lane == 0) { hard[2 * wv] = (uint32_t)hb; hard[2 * wv + 1] = (uint32_t)(hb >> 32); }
        }
        __syncthreads();
        TT(2);                                               // boundaries
        // a cut chunk: only what lies before the last boundary among the rows is complete; the rest
        // goes back on the list as a chunk of its own (nothing spans that boundary)
        uint32_t rows = total;
        if (cut) {
            int last = -1;
            for (int w = SEG_ROWS / 32 - 1; w >= 0 && last < 0; w--) if (hard[w]) last = 32 * w + 31 - __clz((int)hard[w]);
            rows = (uint32_t)(last + 1);
            if (last < 0 && tid == 0) ctl[2] = 1u;           // no boundary at all: left to the node-list loops
        }
        // ---- every row that starts a segment: up to 8 bytes are merged by the row's own lane (all spans
        //      are in the table), longer ones go to a group of 16 lanes, a wavefront, or back on the list
#ifndef SPL_TAIL_CUT
#define SPL_TAIL_CUT 0           /* timing experiments only (tokens missing): 1 no segment merges, 2 no table probes either */
#endif
        if (SPL_TAIL_CUT < 1 && (uint32_t)tid < rows && (tid == 0 || ((hard[(tid - 1) >> 5] >> ((tid - 1) & 31)) & 1u))) {
            const uint32_t h0 = hbits(tid);                  // the first boundary at or after the start ends the segment
            if (h0 & 0xFFu) {
                const int len = __ffs((int)h0);
                const uint32_t gpos = my_gpos;                         // (= first_byte_of(tid), without the search for the row's chunk)
                const uint32_t* const cells = slab + tid * SUB_W;      // node x of the segment: cells + x * SUB_W
                uint32_t alive = (1u << len) - 1u;
                for (;;) {                                   // bpe.rs:118-190 on at most 8 nodes in a bit mask
                    uint32_t best = SPL_NO_RANK, kill = 0, m = alive;
                    int x = __ffs((int)m) - 1;
                    m &= m - 1u;
                    while (m) {
                        const int y = __ffs((int)m) - 1;
                        const uint32_t m2 = m & (m - 1u);
                        const int e2 = m2 ? __ffs((int)m2) - 1 : len;
                        const uint32_t r = cells[x * SUB_W + (e2 - x - 2)];
                        if (r < best) { best = r; kill = 1u << y; }
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
                    emit_g(gpos + (uint32_t)x, e2 - x == 1 ? sid[tid + x] : cells[x * SUB_W + (e2 - x - 2)]);
                }
            } else if (h0 & 0xFFFFu) {
                mseg[atomicAdd(&ctl[6], 1u)] = (uint32_t)tid | (uint32_t)__ffs((int)h0) << 16;
            } else {
                const uint32_t h1 = hbits(tid + 32);
                const uint32_t l2 = h0 ? (uint32_t)__ffs((int)h0) : h1 ? 32u + (uint32_t)__ffs((int)h1) : 65u;
                if (l2 <= 64u) lseg[atomicAdd(&ctl[1], 1u)] = (uint32_t)tid | l2 << 16;
                else {
                    int q = tid + 64;
                    uint32_t hq;
                    while ((hq = hbits(q)) == 0) q += 32;    // (the last row of a chunk is a boundary)
                    const uint32_t l3 = (uint32_t)(q - tid) + (uint32_t)__ffs((int)hq);
                    const uint32_t qi = nl + (l3 <= 64u * XNPL ? 0u : atomicAdd(&ctl[5], 1u));
                    if (l3 <= 64u * XNPL) xseg[atomicAdd(&ctl[7], 1u)] = (uint32_t)tid | l3 << 16;   // a wavefront, several nodes per lane
                    else if (qi < (uint32_t)DIRECT_LQCAP) {  // longer still: a chunk of its own for the loops below
                        s_lq[2 * qi] = my_gpos;
                        s_lq[2 * qi + 1] = l3;
                        atomicOr(&ctl[3], 1u << qi);         // (not to be packed again)
                    } else {                                 // no room: the whole chunk stays on the list
                        atomicOr(&ctl[2], 1u << chunk_of(tid));
                    }
                }
            }
        }
        __syncthreads();
        TT(3);                                               // segments of up to 8 bytes, classification of the rest
        // ---- segments of 9..16 bytes: a group of 16 lanes each ------------------------------------------
        {
            const int gi = tid >> 4, gl = tid & 15;
            const uint32_t nmid = ctl[6];
            for (uint32_t q0 = 0; q0 < nmid; q0 += NT / 16) {
                const uint32_t q = q0 + (uint32_t)gi;
                const int s0 = q < nmid ? (int)(mseg[q] & 0xFFFFu) : 0, len = q < nmid ? (int)(mseg[q] >> 16) : 0;
                const uint32_t gpos = len ? first_byte_of(s0) : 0u;
                const bool gown = gl < len;
                group16_merge(T, slab + (gown ? s0 + gl : 0) * SUB_W, gown ? sid[s0 + gl] : SPL_DEAD, len, FAR_UNBOUNDED,
                              [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
            }
        }
        TT(4);                                               // 9..16 (thread 0's wavefront)
        // ---- segments of 17..64 bytes: one wavefront each ------------------------------------------
        for (uint32_t q = (uint32_t)wv; q < ctl[1]; q += NT / 64) {
            const int s0 = (int)(lseg[q] & 0xFFFFu), len = (int)(lseg[q] >> 16);
            const uint32_t gpos = first_byte_of(s0);
            const bool lown = lane < len;
            const uint32_t* const lrow = slab + (lown ? s0 + lane : 0) * SUB_W;
            wave64_merge(T, lrow, len >= 64 ? ~0ull : ((1ull << len) - 1ull), len, lane + 1 < len ? lrow[0] : SPL_NO_RANK,
                         lown ? sid[s0 + lane] : SPL_DEAD, FAR_UNBOUNDED,
                         [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
        }
        for (uint32_t q = (uint32_t)wv; q < ctl[7]; q += NT / 64) {          // 65 .. 64 XNPL bytes
            const int s0 = (int)(xseg[q] & 0xFFFFu), len = (int)(xseg[q] >> 16);
            const uint32_t gpos = first_byte_of(s0);
            wave_tab_merge<XNPL>(T, len, slab + s0 * SUB_W, [&](int i) { return sid[s0 + i]; },
                                 [&](int i, uint32_t id) { emit_g(gpos + (uint32_t)i, id); });
        }
        __syncthreads();
        TT(5);                                               // 17..64, 65.. and the wait for the other wavefronts
        TT_COUNT();
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
    }
    __syncthreads();
#undef TT
#undef TT_COUNT
    const uint32_t nl2 = nl + ctl[5];                       // the list grew by the segments set aside
    if (SPL_TAIL_SKIP_EMPTY && !ctl[8]) return 0u;          // (as of the last, empty pass: nothing of two bytes or more is left)
    return nl2 < (uint32_t)DIRECT_LQCAP ? nl2 : (uint32_t)DIRECT_LQCAP;
}

}  // namespace spl
