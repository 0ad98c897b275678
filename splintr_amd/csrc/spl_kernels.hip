// spl_kernels.hip -- gfx950 kernels of the batch encode path (DESIGN.md 4 has the full table).
//
//   k_pretok<tile, halo>
//                   one workgroup per tile: stage the window in LDS, classify code points, class
//                   bit masks and context-free sync points (spl_scan_masks.h), one scanner chain
//                   per lane (spl_scan.h), whole-chunk vocabulary probe (spl_lookup.h), and
//                   byte_pair_encode (src/core/bpe.rs:67-197) for the tile's misses with tabulated
//                   pair ranks -- reference: Tokenizer::encode, src/core/tokenizer.rs:729-808.
//                   DIRECT (tile-owned and queue mode): the tile also finishes its long chunks
//                   (bpe_tail_segments, node-list loops) and the chain that outgrew its window, and
//                   leaves a self-contained record
//   k_tile_out      tile-owned mode: tile records -> dense ids[] and per-document offsets (CSR)
//   k_mark_docs / k_special_scan
//                   text-start bitmap from the document offsets; special-token literals
//   k_deferred_wave, k_bpe_segments, k_bpe_long
//                   queue mode: chains and chunks that went to the global queues
//   k_range_count / k_range_out (queue mode)
//                   token bitmaps -> ranks -> CSR
// (The multi-pass pipeline of rounds 1-3 -- k_bpe_lanes64, k_count, k_scan, k_compact_docs and the k_pretok
//  instantiations without tile records -- was removed in round 4: no BASELINE configuration reached it.)
//   k_gatherv_pack / k_gatherv_unpack, k_decode
//                   slabs around the RCCL all-gather; id -> bytes gather
//
// Token bookkeeping: a token is identified by the byte position where it starts.  In the multi-pass
// pipeline producers set a bit in `tbits` and store the id at `stage[pos]`; the final order is the
// bitmap order, so ranks are popcount prefix sums and no kernel needs to know how many tokens
// another produced.  Tile-owned mode keeps the same bookkeeping per tile in LDS.
#include <hip/hip_runtime.h>

#include "spl_common.h"
#include "spl_lookup.h"
#include "spl_scan.h"
#include "spl_scan_masks.h"
#include "spl_scan_starts.h"
#include "spl_scan_words.h"

#define SPL_DBG_WG (b.dbg_wg == 0xFFFFFFFFu ? gridDim.x / 2 : b.dbg_wg)

#ifndef SPL_NO_SLOWPATH
#define SPL_NO_SLOWPATH 0      /* 1: timing experiment only (wrong ids for keys that overflowed their bucket): a full bucket never sends a probe on to the next one */
#endif

namespace spl {

constexpr int LH = 32;                   // left halo (previous character's class)
constexpr int WPAD = 16;                 // real bytes staged past the window (straddling chars, load32)
constexpr int NT = 256;
constexpr int RANK_BLK = 1024;           // positions per rank block (32 bitmap words)
// tile geometries (tile bytes, right halo): chains may run past the tile into the halo
#ifndef SPL_TILE_SMALL
#define SPL_TILE_SMALL 768, 224          /* window 1024 B: one 4-byte word per lane */
#endif
// Tile-owned mode: the same 1024-byte window with more of it owned.  The halo only has to hold the chunk that
// straddles the tile's end and the next sync point (anything longer is finished from a moving window), and every
// byte of halo is classified and masked twice: batches that fill the GPU several times over gain 3-10 % from
// 864 + 128 (English / code most); a batch of about 1 MB -- every tile resident at once, the step ends with
// k_tile_out, whose work grows with the tile -- is best at 800 + 192 (profiles/r02_tile_geometry.txt).
#ifndef SPL_TILE_DIRECT_A
#define SPL_TILE_DIRECT_A 800, 192       /* batches up to SPL_DIRECT_A_MAX_BYTES */
#endif
#ifndef SPL_TILE_DIRECT_B
#define SPL_TILE_DIRECT_B 864, 128
#endif
#ifndef SPL_DIRECT_A_MAX_BYTES
#define SPL_DIRECT_A_MAX_BYTES (1280u * 1024u)
#endif
#ifndef SPL_DIRECT_MAX_MB
#define SPL_DIRECT_MAX_MB 256
#endif
#ifndef SPL_QUEUE_MAX_MB
#define SPL_QUEUE_MAX_MB 2047         /* 0: queue mode off (larger batches then run the multi-pass pipeline) */
#endif
constexpr uint64_t SPL_QUEUE_MAX_BYTES = (uint64_t)SPL_QUEUE_MAX_MB << 20;
constexpr uint64_t SPL_DIRECT_MAX_BYTES = (uint64_t)SPL_DIRECT_MAX_MB << 20;   // batches up to this size: small tiles, tile-owned mode

// What a tile of the tile-owned mode leaves behind for k_tile_out.
struct TileDesc {
    uint32_t slot;             // first entry of the tile's window tokens in tile_ids[]
    uint32_t c_win;            // tokens that start inside the window (ids in tile_ids[])
    uint32_t c_ovf;            // tokens that start beyond it (ids in stage[], bits in tbits[])
    uint32_t ovf_hi;           // end (exclusive) of the byte range those occupy; 0 if none
    uint32_t d_first, d_cnt;   // documents that start in the tile: off_out[d] holds the LOCAL rank
    uint32_t ovf_lo;           // first byte beyond the window
    uint32_t c_own;            // queue mode: window tokens that start inside the tile's own byte range
};

constexpr int TILE_BITS_W = 36;   // >= window words + 1 of the small tile (34)

struct Batch {
    const uint8_t* text;
    uint32_t n_bytes;
    const uint64_t* doc_off;
    uint32_t n_docs;
    uint32_t* tstart;      // bitmap: a text starts at this byte
    uint32_t* skip;        // bitmap: byte belongs to a special-token literal (nullptr: none)
    const uint8_t* sp_lits; // special literals: n_special records of SP_REC bytes (general sets: SPG_REC + blob)
    uint32_t n_special;
    uint32_t* spcand;      // general literal sets: bitmap of the positions where some literal ENDS (k_special_ends)
    uint32_t* tbits;       // bitmap: a token starts at this byte
    uint32_t* stage;       // id of the token starting at this byte
    uint32_t* rank_scr;    // per-byte scratch for oversize chunks (bpe_block_rounds)
    uint32_t* aux;         // two more words per byte for the same
    uint32_t* qcount;      // [0] q64 [2] qlong [3] qdefer (global queues), [6] [7] work cursors of k_bpe_long
    uint2* q64;            // large batches: 17..64-byte misses, appended one workgroup at a time
    uint2* qlong; uint32_t* qdefer;
    uint32_t qcap64, qcaplong, qcapdefer;
    unsigned long long* dbg;   // optional phase cycle stamps of one k_pretok workgroup
    uint32_t stop_phase;       // profiling only: k_pretok returns at this phase boundary (0 = never)
    uint32_t dbg_wg;           // profiling only: the workgroup whose stamps are recorded (0xFFFFFFFF: the middle one)
    uint32_t* blk_base;    // exclusive token count per RANK_BLK block (+1 entry: total)
    uint32_t n_blk;
    uint32_t* ids_out; uint64_t ids_cap; uint64_t* off_out;
    // tile-owned mode (k_pretok<.., DIRECT> + k_tile_out)
    TileDesc* tdesc;           // one record per tile
    uint32_t* tile_ids;        // the tiles' window tokens, one fixed slot of tslot words per tile
    uint32_t* tctl;            // [16 + par * tgroups ...] token sums per 64 tiles, two parities ([0..15] spare)
    uint32_t tgroups;          // capacity of one parity's group-sum array
    uint32_t tslot;            // words per tile in tile_ids[] (window size + 1)
    // queue mode (tile-owned tiles + global queues for what is long, batches beyond the two-launch
    // limit): the tile's window token bitmap goes to tile_bits[] (TILE_BITS_W words per tile)
    uint32_t* tile_bits;
    uint32_t* tcnt;            // tokens per tile RANGE (k_range_count)
    // optional second copy of the result, laid out as a ragged all-gather slab (k_gatherv_pack's
    // format): k_tile_out writes it in the same pass, the separate pack launch goes away
    uint32_t* slab; uint32_t slab_cap, slab_max_docs;
    uint32_t tpar;             // parity of this call
    uint64_t* off_out2;        // tile-owned mode, optional: k_tile_out stores the final offsets here as well (pinned host memory)
    // EXTERNAL chunk boundaries (split patterns the scanner does not implement: the host splitter, spl_regex.h):
    // bit p of ext_starts: a chunk -- or a stretch of dropped bytes -- starts at byte p; bit p of ext_gaps: byte p is
    // dropped (no match covers it).  Tile-owned mode only; the scanner phases are skipped.
    const uint32_t* ext_starts; const uint32_t* ext_gaps;
};

// LDS hand-over between the lanes of ONE wavefront (no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// threadIdx.x for the helpers k_pretok inlines: opaque, so that what is derived from it (lane and group indices, row
// addresses, compare masks) is computed where it is used.  Left to common-subexpression elimination those values were kept
// alive across the 46 000-instruction kernel: with this, k_pretok<800,192> has NO spilled VGPR (round 3: 20, 84 B of scratch
// per lane) -- found while building the persistent-workgroup experiment (profiles/r04_persistent_workgroups.txt), whose loop
// made LLVM hoist all of it.
__device__ __forceinline__ uint32_t tidx() {
    uint32_t x = threadIdx.x;
    asm volatile("" : "+v"(x));
    return x;
}

// ------------------------------------------------------------------------------------------
__global__ void k_mark_docs(Batch b) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= b.n_docs) return;
    const uint64_t p = b.doc_off[d];
    if (p < b.n_bytes) atomicOr(&b.tstart[p >> 5], 1u << (p & 31));
}

// Special-token literals (reference src/core/tokenizer.rs:842-874: Aho-Corasick, Standard match
// kind, non-overlapping find_iter).  spl_add_special only admits literal sets in which no
// occurrence can overlap another (no literal contains another, no proper suffix of one is a
// prefix of another), so every occurrence is a match and positions are independent: one lane per
// byte compares the literals that start with that byte.  A match inside one text
//   * becomes a token at its first byte (id = the literal's id),
//   * is masked out of the text (skip bits; its first byte reads as end-of-text from the left),
//   * makes the byte after it a text start.
// Record layout (SP_REC = 40 bytes): u8 len | u8[3] pad | u32 id | u8 bytes[32].
// The buffer starts with a 32-byte header: the set of the literals' first bytes, so that all but the
// candidate positions leave after one bit test.  The byte after a match starts a text: its bit is
// set right here (if another literal starts there its skip bit wins in every reader, and occurrences
// never overlap, so that bit can never fall strictly inside a literal someone else is checking).
constexpr int SP_REC = 40;
constexpr int SP_HDR = 32;
constexpr int SP_MAXLEN = 32;
__global__ void k_special_scan(Batch b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_bytes) return;
    const uint32_t c0 = b.text[p];
    if (!((reinterpret_cast<const uint32_t*>(b.sp_lits)[c0 >> 5] >> (c0 & 31)) & 1u)) return;
    for (uint32_t k = 0; k < b.n_special; k++) {
        const uint8_t* rec = b.sp_lits + SP_HDR + (size_t)k * SP_REC;
        if (rec[8] != c0) continue;
        const uint32_t len = rec[0];
        if (p + len > b.n_bytes) continue;
        bool ok = true;
        for (uint32_t i = 1; i < len && ok; i++) ok = b.text[p + i] == rec[8 + i];
        // the occurrence must lie inside one document
        for (uint32_t i = 1; i < len && ok; i++) ok = !((b.tstart[(p + i) >> 5] >> ((p + i) & 31)) & 1u);
        if (!ok) continue;
        uint32_t id;
        memcpy(&id, rec + 4, 4);
        b.stage[p] = id;
        atomicOr(&b.tbits[p >> 5], 1u << (p & 31));
        for (uint32_t i = 0; i < len; i++) atomicOr(&b.skip[(p + i) >> 5], 1u << ((p + i) & 31));
        if (p + len < b.n_bytes) atomicOr(&b.tstart[(p + len) >> 5], 1u << ((p + len) & 31));
        return;
    }
}

// GENERAL literal sets (occurrences may overlap: one literal contains another, a suffix of one is a prefix
// of another, literals of up to 255 bytes).  The reference's matcher is Aho-Corasick with MatchKind::Standard
// driven by a non-overlapping find_iter (src/core/tokenizer.rs:429-434, 849-869): from the end of the
// previous match it reports the occurrence that ENDS first, the longest one on a tie, and goes on behind it.
// Two launches:
//   k_special_ends    one lane per byte: does ANY literal end here, inside one document?  -> spcand bitmap
//   k_special_select  one lane per document: walks the document's candidate ends in order and keeps the
//                     longest literal that ends there and starts at or behind the previous match's end --
//                     exactly the automaton restarted at that point -- then marks token, span and the
//                     text start behind it as k_special_scan does.  Candidates are sparse, so the walk is
//                     mostly skipping zero words.
// Table layout: 32-byte header = set of the literals' LAST bytes; n records of SPG_REC bytes
// {u32 len, u32 id, u32 blob offset, u32 last byte}; then the literal bytes.
constexpr int SPG_REC = 16;
__device__ __forceinline__ bool spg_match(const Batch& b, const uint8_t* blob, uint32_t k, uint32_t e, uint32_t min_start, uint32_t& len_out,
                                          uint32_t& id_out) {
    const uint32_t* rec = reinterpret_cast<const uint32_t*>(b.sp_lits + SP_HDR + (size_t)k * SPG_REC);
    const uint32_t len = rec[0];
    if (len > e || e - len < min_start) return false;
    const uint8_t* lit = blob + rec[2];
    const uint8_t* t = b.text + (e - len);
    for (uint32_t i = 0; i < len; i++) if (t[i] != lit[i]) return false;
    len_out = len; id_out = rec[1];
    return true;
}
__global__ void k_special_ends(Batch b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_bytes) return;
    const uint32_t c = b.text[p];
    if (!((reinterpret_cast<const uint32_t*>(b.sp_lits)[c >> 5] >> (c & 31)) & 1u)) return;
    const uint8_t* blob = b.sp_lits + SP_HDR + (size_t)b.n_special * SPG_REC;
    for (uint32_t k = 0; k < b.n_special; k++) {
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(b.sp_lits + SP_HDR + (size_t)k * SPG_REC);
        if (rec[3] != c) continue;
        uint32_t len, id;
        if (!spg_match(b, blob, k, p + 1u, 0u, len, id)) continue;
        bool ok = true;                                   // the occurrence must lie inside one document
        for (uint32_t i = p + 2u - len; i <= p && ok; i++) ok = !((b.tstart[i >> 5] >> (i & 31)) & 1u);
        if (!ok) continue;
        atomicOr(&b.spcand[p >> 5], 1u << (p & 31));
        return;
    }
}
__global__ void k_special_select(Batch b) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= b.n_docs) return;
    const uint32_t lo = (uint32_t)b.doc_off[d], hi = (uint32_t)b.doc_off[d + 1];
    if (hi <= lo) return;
    const uint8_t* blob = b.sp_lits + SP_HDR + (size_t)b.n_special * SPG_REC;
    uint32_t last = lo;                                   // end of the previous match: the automaton restarts here
    for (uint32_t w = lo >> 5; w <= (hi - 1u) >> 5; w++) {
        uint32_t word = b.spcand[w];
        if (w == (lo >> 5)) word &= ~0u << (lo & 31);
        if (w == ((hi - 1u) >> 5) && ((hi & 31u) != 0)) word &= (1u << (hi & 31u)) - 1u;
        while (word) {
            const uint32_t p = w * 32u + (uint32_t)(__ffs((int)word) - 1);
            word &= word - 1u;
            const uint32_t c = b.text[p];
            uint32_t best_len = 0, best_id = 0;
            for (uint32_t k = 0; k < b.n_special; k++) {
                const uint32_t* rec = reinterpret_cast<const uint32_t*>(b.sp_lits + SP_HDR + (size_t)k * SPG_REC);
                if (rec[3] != c || rec[0] <= best_len) continue;
                uint32_t len, id;
                if (spg_match(b, blob, k, p + 1u, last, len, id)) { best_len = len; best_id = id; }
            }
            if (!best_len) continue;
            const uint32_t st = p + 1u - best_len, e = p + 1u;
            b.stage[st] = best_id;
            atomicOr(&b.tbits[st >> 5], 1u << (st & 31));
            for (uint32_t i = st; i < e; i++) atomicOr(&b.skip[i >> 5], 1u << (i & 31));
            if (e < b.n_bytes) atomicOr(&b.tstart[e >> 5], 1u << (e & 31));
            last = e;
        }
    }
}

// ------------------------------------------------------------------------------------------
struct LdsAcc {
    const uint8_t* rec_;
    const uint8_t* txt_;
    __device__ __forceinline__ uint32_t rec(int q) const { return rec_[q]; }
    __device__ __forceinline__ uint32_t txt(int q) const { return txt_[q]; }
    __device__ __forceinline__ uint32_t load32(int p) const {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(txt_) + (p >> 2);
        return __builtin_amdgcn_alignbyte(w[1], w[0], p & 3);
    }
};

// v_writelane_b32: lane `lane` of `old` takes the wave-uniform value src.  (This compiler has no builtin for the
// intrinsic; through inline asm the hazard between a v_cmp that writes the SGPR and the read here went unhandled.)
extern "C" __device__ int spl_writelane(int src, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
#define write_lane(v, s, lane_) spl_writelane((int)(s), (lane_), (v))

// Window-wide bit vector for spl_scan_starts.h: one 32-bit word per lane of ONE wavefront (all 64 lanes
// active; lanes past the window hold zero words).  Shifts take the neighbour lane's word by DPP.
struct WaveBV {
    uint32_t x;
    __device__ __forceinline__ WaveBV operator&(const WaveBV& o) const { return WaveBV{x & o.x}; }
    __device__ __forceinline__ WaveBV operator|(const WaveBV& o) const { return WaveBV{x | o.x}; }
    __device__ __forceinline__ WaveBV operator~() const { return WaveBV{~x}; }
    __device__ __forceinline__ WaveBV shl1() const {       // bit i <- bit i - 1
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xF, 0xF, true);   // wave_shr:1
        return WaveBV{(x << 1) | (prev >> 31)};
    }
    __device__ __forceinline__ WaveBV shr1() const {       // bit i <- bit i + 1
        const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xF, 0xF, true);   // wave_shl:1
        return WaveBV{(x >> 1) | (next << 31)};
    }
    __device__ __forceinline__ bool any() const { return __any(x != 0u); }
};

// LdsAcc plus the window's class bitmasks (spl_scan_masks.h)
struct MaskLdsAcc {
    const uint8_t* rec_;
    const uint8_t* txt_;
    const uint32_t* mk_;      // [MK_COUNT][nbw]
    int nbw_, w_;
    bool eot_;
    __device__ __forceinline__ uint32_t rec(int q) const { return rec_[q]; }
    __device__ __forceinline__ uint32_t txt(int q) const { return txt_[q]; }
    __device__ __forceinline__ uint32_t mw(int which, int w) const { return mk_[which * nbw_ + w]; }
    __device__ __forceinline__ int wbits() const { return w_; }
    __device__ __forceinline__ bool end_is_eot() const { return eot_; }
};

// The long-chunk queue is filled from both ends: chunks the 16-lane groups of k_bpe_long take (up to
// 128 bytes) from the front, larger ones from the back -- each phase of k_bpe_long then walks only
// its own items (walking all of them cost one same-address atomic per item and wavefront phase).
// Chunks do not overlap and have at least two bytes, so the two ends never meet (capacity n_bytes/2).
constexpr int LONG_SMALL_NMAX = 128;
__device__ __forceinline__ void push_long(const Batch& b, uint32_t pos, uint32_t len) {
    if (len <= (uint32_t)LONG_SMALL_NMAX) {
        const uint32_t i = atomicAdd(&b.qcount[2], 1u);
        if (i < b.qcaplong) b.qlong[i] = make_uint2(pos, len);
    } else {
        const uint32_t i = atomicAdd(&b.qcount[4], 1u);
        if (i < b.qcaplong) b.qlong[b.qcaplong - 1u - i] = make_uint2(pos, len);
    }
}

// byte_pair_encode (reference src/core/bpe.rs:67-197) by a GROUP OF 16 LANES holding up to
// 16*NPL nodes in registers: node i (the token that starts at byte i of the chunk) lives in lane
// i % 16, slot i / 16.  Four chunks per wavefront advance in lock step.  Per merge:
//   * key = (rank << 8 | node index), minimum over the lane's slots, then a DPP min-reduction
//     inside the 16-lane row -> the leftmost minimum (bpe.rs:121-138);
//   * right neighbour / the one after / left neighbour from the group's alive bitmap, which every
//     lane of the group keeps and updates identically (no ballots);
//   * the winner takes the merged id (= the pair's rank), its right neighbour dies, and the two
//     affected pairs are re-ranked by the two lanes that own them in one predicated pair-table
//     probe, so both loads are in flight together (bpe.rs:160-166).
// No LDS arrays, no scratch.  `byte_at(i)` supplies chunk bytes, `emit(i, id)` takes survivors.
// all-reduce(min) inside each 16-lane row: quad xor 1, quad xor 2, half-row mirror, row mirror.
// The compiler turns update_dpp + min into v_mov_b32_dpp + v_min_u32 (two instructions and a wait
// state per step); v_min_u32_dpp does a step in one.  (s_nop 1: a VALU result needs two wait
// states before a DPP read; hazards inside inline asm are not the compiler's business.)
__device__ __forceinline__ uint32_t row16_min(uint32_t x) {
#ifndef SPL_NO_DPP_ASM
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
                 : "+v"(x));
    return x;
#else
    auto step = [](uint32_t v, uint32_t y) { return y < v ? y : v; };
    x = step(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    x = step(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    x = step(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false));   // row_half_mirror
    x = step(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false));   // row_mirror
    return x;
#endif
}
#ifndef SPL_PAIR_SHORT
#define SPL_PAIR_SHORT 1         /* 1: two chunks of up to 8 bytes share a 16-lane group, each in a half (a tile with 17..32 short misses
                                    then needs ONE pull per group more often: its shortest misses are the ones beyond the sixteenth) */
#endif
// The same with the group's width chosen per 16-lane row at run time: 8 lanes (two chunks of up to 8 bytes share a row,
// each in a half) or 16.  Three steps reduce inside the halves; the fourth joins them where the row is one group.
__device__ __forceinline__ uint32_t row_min_sub(uint32_t x, bool whole_row) {
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
                 : "+v"(x));
    uint32_t y = x;
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(y));
    return whole_row ? y : x;
}
// all-reduce(min) over a group of GW = 16 or 32 lanes (32: the two rows of a half exchanged by ds_swizzle)
template <int GW> __device__ __forceinline__ uint32_t group_min(uint32_t x, int sub = GW) {
    if (GW == 16 && SPL_PAIR_SHORT) return row_min_sub(x, sub == 16);     // (one instruction stream for both widths: rows of a wavefront differ)
    x = row16_min(x);
    if (GW == 32) {
#ifndef SPL_NO_PERMLANE_SWAP
        // gfx950: v_permlane16_swap_b32 exchanges the odd rows of one operand with the even rows of the other -- with both
        // operands holding x, one result has every row pair's even row twice, the other its odd row twice: one VALU
        // instruction where ds_swizzle (lane ^ 16) went through the LDS crossbar, in every round of a 17..32-byte word's merge
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        const uint32_t a = r[0], c = r[1];
        x = a < c ? a : c;
#else
        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F);   // lane ^ 16
        x = y < x ? y : x;
#endif
    }
    return x;
}

// Inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts and row broadcasts
// (six full-rate instructions, no LDS permutes and no per-lane address registers to keep alive).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return x;
}

// The same for the running maximum (values are non-negative: shifted-in zeros are neutral).
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
    auto mx = [](uint32_t a, int b) { return a > (uint32_t)b ? a : (uint32_t)b; };
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
    x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
    return x;
}

// 8 x 8 NIBBLE transpose across each group of 8 neighbouring lanes (spl_scan_words.h): in, nibble j of lane l; out,
// nibble l of lane j.  Three butterfly stages (lane ^ 4 / ^ 2 / ^ 1 with 16 / 8 / 4 bits): the partner's word by DPP, rotated
// so that the nibbles to take line up (v_alignbit), merged under a per-lane mask (v_bfi).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t nib_transpose8(uint32_t v) {
    const uint32_t l = tidx() & 7u;
    {   // stride 4: lanes with bit 2 clear keep nibbles 0-3 and take the partner's 0-3 as their 4-7; the others the mirror image
        uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);     // row_shl:4 -> banks 0, 2 (lane + 4)
        t = (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x114, 0xF, 0xA, false);               // row_shr:4 -> banks 1, 3 (lane - 4)
        const uint32_t km = (l & 4u) ? 0xFFFF0000u : 0x0000FFFFu;
        const uint32_t y = __builtin_amdgcn_alignbit(t, t, 16);
        v = (v & km) | (y & ~km);
    }
    {   // stride 2
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
        const uint32_t km = (l & 2u) ? 0xFF00FF00u : 0x00FF00FFu;
        const uint32_t y = __builtin_amdgcn_alignbit(t, t, (l & 2u) ? 8u : 24u);
        v = (v & km) | (y & ~km);
    }
    {   // stride 1
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
        const uint32_t km = (l & 1u) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
        const uint32_t y = __builtin_amdgcn_alignbit(t, t, (l & 1u) ? 4u : 28u);
        v = (v & km) | (y & ~km);
    }
    return v;
}

// Alive bitmaps are arrays of 32-bit words (64-bit shifts and bit scans are multi-instruction
// and slow on the vector ALU; v_ffbl_b32 / v_ffbh_u32 / 32-bit shifts are single full-rate ops).
template <int NW> __device__ __forceinline__ int next_set_bit(const uint32_t (&a)[NW], int from) {
    int res = -1;                                     // lowest set bit with index >= from
#pragma unroll
    for (int w = NW - 1; w >= 0; w--) {
        uint32_t x = a[w];
        const int lo = from - 32 * w;
        if (lo >= 32) x = 0;
        else if (lo > 0) x &= ~((1u << lo) - 1u);
        if (x) res = 32 * w + __ffs((int)x) - 1;
    }
    return res;
}
template <int NW> __device__ __forceinline__ int prev_set_bit(const uint32_t (&a)[NW], int before) {
    int res = -1;                                     // highest set bit with index < before
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t x = a[w];
        const int hi = before - 32 * w;
        if (hi <= 0) x = 0;
        else if (hi < 32) x &= (1u << hi) - 1u;
        if (x) res = 32 * w + 31 - __clz((int)x);
    }
    return res;
}

#ifdef SPL_MERGE_TIMING
__device__ unsigned long long g_mt[8];
#define MT_T(v) const long long v = clock64()
#define MT_ACC(i, a, b_) do { if (tidx() == 0 && blockIdx.x == gridDim.x / 2) g_mt[i] += (unsigned long long)((b_) - (a)); } while (0)
#else
#define MT_T(v)
#define MT_ACC(i, a, b_)
#endif
template <int NPL, class ByteAt, class Emit>
__device__ __forceinline__ void bpe_group16(const DeviceTables& T, int n, ByteAt byte_at, Emit emit) {
    constexpr int NW = (16 * NPL + 31) / 32;
    MT_T(t_init0);
    const int lane = tidx() & 63;
    const int gl = lane & 15;
    const int gbase = lane - gl;
    uint32_t id[NPL], rk[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int i = gl + 16 * k;
        id[k] = i < n ? T.byte_id[byte_at(i)] : SPL_DEAD;
    }
#pragma unroll
    for (int k = 0; k < NPL; k++) {                    // initial ranks (bpe.rs:114-116)
        const uint32_t same_slot = __shfl(id[k], gbase + ((gl + 1) & 15));
        const uint32_t next_slot = __shfl(k + 1 < NPL ? id[k + 1 < NPL ? k + 1 : k] : (uint32_t)SPL_DEAD, gbase);
        const uint32_t idn = gl < 15 ? same_slot : next_slot;
        rk[k] = (gl + 16 * k + 1 < n) ? pair_rank(T, id[k], idn) : SPL_NO_RANK;
    }
    uint32_t alive[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int c = n - 32 * w;
        alive[w] = c >= 32 ? ~0u : c > 0 ? (1u << c) - 1u : 0u;
    }
    MT_T(t_init1);
    MT_ACC(0, t_init0, t_init1);
    for (;;) {
        MT_T(t0);
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NPL; k++) {
            const uint32_t c = rk[k] == SPL_NO_RANK ? 0xFFFFFFFFu : ((rk[k] << 8) | (uint32_t)(gl + 16 * k));
            key = c < key ? c : key;
        }
        const uint32_t m = row16_min(key);
        const bool active = m != 0xFFFFFFFFu;
        if (!__any(active)) break;
        MT_T(t1);
        MT_ACC(1, t0, t1);
        const int mi = (int)(m & 255u);
        const uint32_t mn = m >> 8;
        // neighbours (group-uniform; meaningless but harmless when the group is idle)
        const int j = active ? next_set_bit<NW>(alive, mi + 1) : 0;
        const int j2 = active ? next_set_bit<NW>(alive, j + 1) : -1;
        const int h = active ? prev_set_bit<NW>(alive, mi) : -1;
        uint32_t sel_j2 = id[0], sel_h = id[0];
#pragma unroll
        for (int k = 1; k < NPL; k++) {
            sel_j2 = (j2 >> 4) == k ? id[k] : sel_j2;
            sel_h = (h >> 4) == k ? id[k] : sel_h;       // own slot: only meaningful in the lane that owns h
        }
        MT_T(t2);
        MT_ACC(2, t1, t2);
        const uint32_t id_j2 = __shfl(sel_j2, gbase + (j2 & 15));
        MT_T(t3);
        MT_ACC(3, t2, t3);
        // the owner of mi re-ranks (mi, j2), the owner of h re-ranks (h, mi): one predicated probe,
        // two loads in flight.  Only when both nodes sit in the same lane (NPL > 1) does that lane
        // probe a second time.
        const int la = mi & 15, lh = h & 15;
        uint32_t res = SPL_NO_RANK, res2 = SPL_NO_RANK;
        if (active) {
            if (gl == la) { if (j2 >= 0) res = pair_rank(T, mn, id_j2); }
            else if (h >= 0 && gl == lh) res = pair_rank(T, sel_h, mn);
            if (NPL > 1 && h >= 0 && la == lh && gl == la) res2 = pair_rank(T, sel_h, mn);
#ifdef SPL_MERGE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            { MT_T(t4); MT_ACC(4, t3, t4); if (tidx() == 0 && blockIdx.x == gridDim.x / 2) g_mt[6] += 1; }
#endif
#pragma unroll
            for (int w = 0; w < NW; w++)
                if ((j >> 5) == w) alive[w] &= ~(1u << (j & 31));
#pragma unroll
            for (int k = 0; k < NPL; k++) {
                const int i = gl + 16 * k;
                if (i == mi) { id[k] = mn; rk[k] = res; }          // res = NO_RANK when there is no right neighbour
                else if (i == j) rk[k] = SPL_NO_RANK;
                else if (i == h) rk[k] = (NPL > 1 && la == lh) ? res2 : res;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int i = gl + 16 * k;
        if (i < n && ((alive[(gl + 16 * k) >> 5] >> (i & 31)) & 1u) && id[k] != SPL_NO_RANK) emit(i, id[k]);
    }
}

// Whole-chunk probe of the tile kernel for keys of up to 12 bytes: the three length classes live in
// three tables, and a wavefront's lanes hold a mix of them.  All lanes first issue their bucket
// loads (two quads always, a third / fourth by class -- predicated loads, no wait in between),
// then compare by class; the wavefront pays ONE memory round trip instead of one per class.
#ifndef SPL_ROW_FILTER
#define SPL_ROW_FILTER 1
#endif
struct alignas(8) Ent2 { uint32_t x, y; };              // one tiny-table entry (dwordx2)
struct alignas(4) Ent3 { uint32_t x, y, z; };           // one t8-table entry (dwordx3, packed at 12-byte stride)
__device__ __forceinline__ uint32_t probe_short_mixed(const DeviceTables& T, uint32_t k0, uint32_t k1, uint32_t k2,
                                                      uint32_t n) {
    const bool tiny = n <= (uint32_t)SPL_TINY_MAX, t8 = !tiny && n <= (uint32_t)SPL_T8_MAX;
    // ONE round trip for what depends on the text alone: the key's two-byte prefix -- which token lengths exist behind it at
    // all (no probe for the others), the salts of its tiny / short hashes, the two-byte token's id -- and the filter entry
    // of its first four bytes (lengths 4..8 and "longer" that exist behind THOSE, and the salt of its t8 hash)
    const PfxEnt pe = T.pfx[k0 & 0xFFFFu];
    const uint32_t f4 = T.filt4[hash_f4(k0) >> T.filt4_shift];
    const uint32_t lm = pe.lm;
    if (n >= 2u && !((lm >> (n <= (uint32_t)SPL_T8_MAX ? n - 2u : 7u)) & 1u)) return SPL_NO_RANK;
    if (n == 2u) return pe.id2;                  // the prefix entry carries the two-byte token's id: no table to read
    if (SPL_ROW_FILTER && n >= 4u && !((f4 >> (n <= (uint32_t)SPL_T8_MAX ? n - 4u : 5u)) & 1u)) return SPL_NO_RANK;
    if (tiny || t8) {
        // one entry, one compare (the builder gave every key a slot of its own)
        const uint32_t h = hash_t8(k0, tiny ? 0u : k1, n, tiny ? lm >> 16 : f4 >> SPL_F4_MASK_BITS);     // (== hash_tiny for a tiny key)
        const uint32_t* e = tiny ? T.tiny_tab + (size_t)(h & T.tiny_mask) * SPL_TINY_WORDS : T.t8_tab + (size_t)(h & T.t8_mask) * SPL_T8_WORDS;
        const Ent3 q = *reinterpret_cast<const Ent3*>(e);          // (a tiny entry and the first word of the next one: the tables are padded)
        const uint32_t idw = tiny ? q.y : q.z;
        const bool hit = (q.x == k0) & (tiny | (q.y == k1)) & ((idw >> 24) == n);
        return hit ? (idw & SPL_ID_MASK) : SPL_NO_RANK;
    }
    return probe_short12(T, k0, k1, k2, n, (lm >> 8) & 0xFFu);
}
template <class TX>
__device__ __forceinline__ uint32_t probe_chunk_tile(const DeviceTables& T, const TX& tx, int p, int n) {
    if (n <= SPL_SHORT_MAX) {
        const uint32_t k0 = mask_tail(tx.load32(p), n);
        const uint32_t k1 = n > 4 ? mask_tail(tx.load32(p + 4), n - 4) : 0u;
        const uint32_t k2 = n > 8 ? mask_tail(tx.load32(p + 8), n - 8) : 0u;
        return probe_short_mixed(T, k0, k1, k2, (uint32_t)n);
    }
    if ((uint32_t)n > T.max_key_len) return SPL_NO_RANK;
    return probe_long(T, tx, p, n);
}

// Short chunks (<= 16 bytes), one node per lane, with the pair ranks TABULATED up front.  The
// reference ranks a pair by looking up the concatenated bytes (bpe.rs:99-111): the rank of (node
// starting at i, its right neighbour ending at e) is the id of the token text[i, e).  The lane that
// owns start i probes the short-key table for text[i, i+len), len = 2..8, in three batches whose
// bucket loads are all in flight together, and keeps the ids in its own LDS row.  The merge loop
// then needs no memory round trip per merge (one LDS read of the lane's own row); only spans
// longer than 8 bytes fall back to the pair table.
#ifndef SPL_SUB_LMAX
#define SPL_SUB_LMAX 8
#endif
constexpr int SUB_LMAX = SPL_SUB_LMAX;
constexpr int SUB_W = SUB_LMAX - 1;          // table width: lengths 2..8

// split probes of the tiny table (keys of 2..4 bytes) and of the t8 table (5..8 bytes)
// (`on` false: the key is known to miss -- the lane loads the table's spare bucket instead, one
//  cache line for all such lanes, and the finish step finds nothing there)
#ifdef SPL_FAKE_FILL      /* timing experiment only (wrong ids): every tabulation probe reads the spare bucket, i.e. always hits */
#define SPL_FILL_ON(on) false
#else
#define SPL_FILL_ON(on) (on)
#endif
__device__ __forceinline__ void tiny_issue_if(const DeviceTables& T, bool on, uint32_t k0, uint32_t n, uint32_t salt, Ent2& q) {
    const uint32_t slot = SPL_FILL_ON(on) ? hash_tiny(k0, n, salt) & T.tiny_mask : T.tiny_free;
    q = *reinterpret_cast<const Ent2*>(T.tiny_tab + (size_t)slot * SPL_TINY_WORDS);
}
__device__ __forceinline__ void t8_issue_if(const DeviceTables& T, bool on, uint32_t k0, uint32_t k1, uint32_t n, uint32_t salt, Ent3& q) {
    const uint32_t slot = SPL_FILL_ON(on) ? hash_t8(k0, k1, n, salt) & T.t8_mask : T.t8_free;
    q = *reinterpret_cast<const Ent3*>(T.t8_tab + (size_t)slot * SPL_T8_WORDS);
}
__device__ __forceinline__ uint32_t tiny_finish(uint32_t k0, uint32_t n, const Ent2& q) {
    return ((q.x == k0) & ((q.y >> 24) == n)) ? (q.y & SPL_ID_MASK) : SPL_NO_RANK;
}
__device__ __forceinline__ uint32_t t8_finish(uint32_t k0, uint32_t k1, uint32_t n, const Ent3& q) {
    return ((q.x == k0) & (q.y == k1) & ((q.z >> 24) == n)) ? (q.z & SPL_ID_MASK) : SPL_NO_RANK;
}

// The merge loop of one 16-lane group over `n` <= 16 nodes whose substring ids are tabulated: lane
// gl owns node gl, `row` is ITS table row, `id` its byte's id.  Survivors go to emit(gl, id).
// far_max (per lane): the longest token of more than SUB_LMAX bytes that can start at this lane's byte
// (p8 table: an upper bound; 0 = none) -- longer spans rank SPL_NO_RANK without a trip to the pair table.
constexpr int FAR_UNBOUNDED = 1 << 20;
// sub (GW == 16 only): 8 if the row holds TWO chunks of up to 8 bytes, one per half, else 16 -- uniform per 16-lane row.
template <int GW, class Emit>
__device__ __forceinline__ void group_merge(const DeviceTables& T, const uint32_t* row, uint32_t id, int n, int far_max, Emit emit, int sub = GW) {
    static_assert(GW == 16 || GW == 32, "groups of 16 or 32 lanes");
    const int lane = tidx() & 63;
    const int gl = lane & (sub - 1);
    const int gbase = lane - gl;
    const bool own = gl < n;
    uint32_t rk = (gl + 1 < n) ? row[0] : SPL_NO_RANK;          // initial ranks (bpe.rs:114-116)
    uint32_t alive = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);   // group-uniform, kept by every lane
    for (;;) {
        const uint32_t key = rk == SPL_NO_RANK ? 0xFFFFFFFFu : ((rk << 8) | (uint32_t)gl);
        const uint32_t m = group_min<GW>(key, sub);
        const bool active = m != 0xFFFFFFFFu;
        if (!__any(active)) break;
        const int mi = (int)(m & 255u);
        const uint32_t mn = m >> 8;
        const uint32_t above = active ? alive & ~((2u << mi) - 1u) : 1u;
        const int j = __ffs((int)above) - 1;
        const uint32_t above2 = above & (above - 1u);
        const int j2 = above2 ? __ffs((int)above2) - 1 : -1;
        const uint32_t above3 = above2 & (above2 - 1u);
        const int e_r = above3 ? __ffs((int)above3) - 1 : n;    // end of the pair (mi, j2)
        const int e_mi = j2 >= 0 ? j2 : n;                       // end of the merged node
        const uint32_t below = active ? alive & ((1u << mi) - 1u) : 0u;
        const int h = below ? 31 - __clz((int)below) : -1;
        const int len_r = e_r - mi, len_h = e_mi - h;
        // Branch-free update: every lane reads the one cell of its own row it could need (the owner
        // of mi the cell of the pair (mi, j2), everybody else -- of whom only the owner of h matters
        // -- the cell of (h, mi)); selects pick the three lanes that change.  Only spans longer than
        // the table (rare) take the branch to the pair table.
        const bool is_mi = gl == mi, is_h = gl == h;
        const int len = is_mi ? len_r : len_h;
        const bool far = active && len > SUB_LMAX && len <= far_max && ((is_mi && j2 >= 0) || is_h);
        const int cell = len - 2 < 0 ? 0 : len - 2 > SUB_W - 1 ? SUB_W - 1 : len - 2;
        uint32_t nr = len > SUB_LMAX ? SPL_NO_RANK : row[cell];
        if (__any(far)) {
            const uint32_t id_j2 = __shfl(id, gbase + (j2 & (sub - 1)));     // only long spans need neighbour ids
            if (far) nr = is_mi ? pair_rank(T, mn, id_j2) : pair_rank(T, id, mn);
        }
        nr = (is_mi && j2 < 0) ? SPL_NO_RANK : nr;
        rk = (active && (is_mi || is_h)) ? nr : (active && gl == j) ? SPL_NO_RANK : rk;
        id = (active && is_mi) ? mn : id;
        alive = active ? alive & ~(1u << j) : alive;
    }
    if (own && ((alive >> gl) & 1u) && id != SPL_NO_RANK) emit(gl, id);
}


// group_merge for groups in which NO token of more than SUB_LMAX bytes can start anywhere (far_max == 0 in every
// lane of the wavefront: nearly every pull): no pair-table branch, no ids carried through the rounds (a
// survivor's id is a cell of its own row), an idle group made harmless by the choice of its "winner"
// instead of by a predicate on every update -- about a fifth fewer instructions per round, in the loop
// that is 40 % of the tile kernel's instructions.
#ifndef SPL_MERGE_NEAR
#define SPL_MERGE_NEAR 1
#endif
template <int GW, class Emit>
__device__ __forceinline__ void group_merge_near(const uint32_t* row, uint32_t id, int n, Emit emit, int sub = GW) {
    const int gl = (tidx() & 63) & (sub - 1);
    uint32_t rk = (gl + 1 < n) ? row[0] : SPL_NO_RANK;
    uint32_t alive = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    for (;;) {
        const uint32_t m = group_min<GW>((rk << 8) | (uint32_t)gl, sub);     // (SPL_NO_RANK << 8 is beyond every real key)
        const bool active = m < 0xFFFFFF00u;
        if (!__any(active)) break;
        // an idle group "merges" at index 31: nothing lies above it, nobody owns it, bit 31 of alive goes (GW = 16)
        const int mi = active ? (int)(m & 255u) : 31;
        const uint32_t above = alive & (~1u << mi);
        const int j = __ffs((int)above) - 1;                             // -1: shifts below count mod 32
        const uint32_t above2 = above & (above - 1u);
        const uint32_t above3 = above2 & (above2 - 1u);
        const int e_mi = above2 ? __ffs((int)above2) - 1 : n;           // end of the merged node
        const int e_r = above3 ? __ffs((int)above3) - 1 : n;            // end of the pair it forms with the next one
        const uint32_t below = alive & ~(~0u << mi);
        const int h = 31 - __clz((int)below);                            // (-1 if there is none: __clz(0) == 32)
        const bool is_mi = (GW == 16 || active) && gl == mi, is_h = active && gl == h;   // (32 lanes: index 31 is a real node)
        const int len = is_mi ? e_r - mi : e_mi - h;
        const int cell = len - 2 < 0 ? 0 : len - 2 > SUB_W - 1 ? SUB_W - 1 : len - 2;
        uint32_t nr = row[cell];
        nr = (len > SUB_LMAX || (is_mi && !above2)) ? SPL_NO_RANK : nr;
        rk = (is_mi || is_h) ? nr : (gl == j) ? SPL_NO_RANK : rk;
        alive &= ~(((GW == 16 || active) ? 1u : 0u) << (j & 31));
    }
    if (gl < n && ((alive >> gl) & 1u)) {
        const uint32_t above = alive & (~1u << gl);
        const int len = (above ? __ffs((int)above) - 1 : n) - gl;
        const uint32_t tok = len == 1 ? id : row[len - 2];
        if (tok != SPL_NO_RANK) emit(gl, tok);
    }
}

template <class Emit>
__device__ __forceinline__ void group16_merge(const DeviceTables& T, const uint32_t* row, uint32_t id, int n, int far_max, Emit emit) {
    group_merge<16>(T, row, id, n, far_max, emit);
}

#ifdef SPL_DEBUG_STAMPS
#define SPL_WT(i) do { if (wt && (tidx() & 63) == 0) wt[i] = clock64(); } while (0)
#else
#define SPL_WT(i) do { } while (0)
#endif
// What a table row starts from, in ONE round trip (both loads depend on the text alone): the prefix entry of the
// lane's first two bytes -- length mask, salt, the id of the two-byte token (no probe for length 2) -- and the
// four-byte-prefix filter, which takes the lengths 4..8 (and "longer") that no token with these four bytes has
// out of the mask: their probes go to the spare bucket like those of the lengths the two-byte prefix rules out.
struct RowHead { uint32_t lm, id2, tsalt, fsalt; };     // lm: the length mask (low byte); tsalt / fsalt: salts of the tiny / t8 hashes
__device__ __forceinline__ RowHead row_head(const DeviceTables& T, bool own, uint32_t w0, int maxlen) {
    RowHead h{0u, SPL_NO_RANK, 0u, 0u};
    if (own) {
        const PfxEnt pe = T.pfx[w0 & 0xFFFFu];
        const uint32_t f = maxlen >= 4 ? (uint32_t)T.filt4[hash_f4(w0) >> T.filt4_shift] : 0u;
        const uint32_t f4 = SPL_ROW_FILTER ? f & 0x3Fu : (maxlen >= 4 ? 0x3Fu : 0u);
        h.lm = pe.lm & 0xFFu & (0x03u | (f4 << 2));
        h.id2 = pe.id2;
        h.tsalt = pe.lm >> 16;
        h.fsalt = f >> SPL_F4_MASK_BITS;
    }
    return h;
}
// The ids of text[pos, pos + len), len = 2..8, of one table row: ALL six probes in flight together -- one entry each
// (round 4; up to round 3 two batches of buckets, a dependent round trip apart, for want of registers).  r[len - 2];
// lengths the masks rule out, or beyond maxlen, read the table's empty slot (one cache line for all such lanes) and
// give SPL_NO_RANK.  maxlen < 2: nothing is loaded.
__device__ __forceinline__ void row_fill(const DeviceTables& T, const RowHead& rh, uint32_t w0, uint32_t w1, int maxlen, uint32_t (&r)[7]) {
#pragma unroll
    for (int k = 0; k < 7; k++) r[k] = SPL_NO_RANK;
    // (ONE predicate for the six probes: with one per length the compiler waits after every single probe instead of
    //  keeping all the loads in flight together)
    if (maxlen >= 2) {
        const uint32_t lm = rh.lm, k3 = w0 & 0xFFFFFFu, h5 = w1 & 0xFFu, h6 = w1 & 0xFFFFu, h7 = w1 & 0xFFFFFFu;
        Ent2 q3, q4;
        Ent3 q5, q6, q7, q8;
        tiny_issue_if(T, (lm & 2u) != 0 && maxlen >= 3, k3, 3u, rh.tsalt, q3);
        tiny_issue_if(T, (lm & 4u) != 0 && maxlen >= 4, w0, 4u, rh.tsalt, q4);
        t8_issue_if(T, (lm & 8u) != 0 && maxlen >= 5, w0, h5, 5u, rh.fsalt, q5);
        t8_issue_if(T, (lm & 0x10u) != 0 && maxlen >= 6, w0, h6, 6u, rh.fsalt, q6);
        t8_issue_if(T, (lm & 0x20u) != 0 && maxlen >= 7, w0, h7, 7u, rh.fsalt, q7);
        t8_issue_if(T, (lm & 0x40u) != 0 && maxlen >= 8, w0, w1, 8u, rh.fsalt, q8);
        r[0] = rh.id2;
        r[1] = tiny_finish(k3, 3u, q3);
        r[2] = tiny_finish(w0, 4u, q4);
        r[3] = t8_finish(w0, h5, 5u, q5);
        r[4] = t8_finish(w0, h6, 6u, q6);
        r[5] = t8_finish(w0, h7, 7u, q7);
        r[6] = t8_finish(w0, w1, 8u, q8);
    }
}
// Tabulation of ONE table row: the lane probes the ids of text[pos, pos + len), len = 2 .. min(rem, 8) -- six entries,
// all in flight together (row_fill) -- into `row`; returns the id of its byte and, in far_max, the longest token of more
// than 8 bytes that can start there (p8 bound; its load rides in the same round trip).  `own` false: idle lane.
__device__ __forceinline__ uint32_t tab_row(const DeviceTables& T, const LdsAcc& tx, bool own, int pos, int rem, uint32_t* row,
                                            int& far_max, long long* wt = nullptr) {
    (void)wt;
    const int maxlen = own ? (rem < SUB_LMAX ? rem : SUB_LMAX) : 0;
    const uint32_t w0 = own ? tx.load32(pos) : 0u;
    const uint32_t w1 = own ? tx.load32(pos + 4) : 0u;
    const uint32_t id = own ? T.byte_id[w0 & 0xFFu] : SPL_DEAD;
    // which token lengths exist at all behind the lane's first two / four bytes: the other probes go to the empty slot
    const RowHead rh = row_head(T, own, w0, maxlen);
    SPL_WT(1);
    far_max = 0;
    // spans of more than 8 bytes (the last merges of a chunk of 9..16 bytes): can a token that long start at this byte at
    // all?  Almost never -- and then its rank is known without the pair table, whose round trip every lane of the wavefront
    // would wait for, merge round after merge round.
    P8Bucket e8{0u, 0u};
    const bool want8 = maxlen >= 2 && rem > SUB_LMAX && (rh.lm & 0x80u);
    if (want8) e8 = T.p8_tab[hash_p8(w0, w1) & T.p8_mask];
    uint32_t r[7];
    row_fill(T, rh, w0, w1, maxlen, r);
    if (want8) {
        const int l8 = (int)p8_match(e8.a, e8.b, p8_tag(w0, w1));
        far_max = l8 == 255 ? FAR_UNBOUNDED : l8;
    }
    if (maxlen >= 2) {
#pragma unroll
        for (int k = 0; k < SUB_W; k++) row[k] = r[k];
    }
    SPL_WT(3);
    return id;
}

// width (GW == 16): 8 if this 16-lane row holds TWO chunks of up to 8 bytes (p, n: per half), else 16
template <int GW, class Emit>
__device__ __forceinline__ void bpe_group_tab(const DeviceTables& T, const LdsAcc& tx, int p, int n, uint32_t* sub,
                                              Emit emit, long long* wt = nullptr, int width = GW) {
    (void)wt;
    SPL_WT(0);
    const int gl = (tidx() & 63) & (width - 1);
    uint32_t* row = sub + ((tidx() & 63) & (GW - 1)) * SUB_W;
    int far_max;
    const uint32_t id = tab_row(T, tx, gl < n, p + gl, n - gl, row, far_max, wt);
    if (SPL_MERGE_NEAR && !__any(far_max > 0)) group_merge_near<GW>(row, id, n, emit, width);
    else group_merge<GW>(T, row, id, n, far_max, emit, width);
    SPL_WT(4);
}
template <class Emit>
__device__ __forceinline__ void bpe_group16_tab(const DeviceTables& T, const LdsAcc& tx, int p, int n, uint32_t* sub,
                                                Emit emit, long long* wt = nullptr, int width = 16) {
    bpe_group_tab<16>(T, tx, p, n, sub, emit, wt, width);
}

// The merge loop of one WAVEFRONT over the nodes in `alive` (lanes of a range that ends at `end`),
// with tabulated substring ids: `row` is the lane's own table row, `rk` its pair's rank, `idv` its id.
// (far_max: as in group16_merge)
template <class Emit>
__device__ __forceinline__ void wave64_merge(const DeviceTables& T, const uint32_t* row, unsigned long long alive, int end,
                                             uint32_t rk, uint32_t idv, int far_max, Emit emit) {
    const int lane = tidx() & 63;
    for (;;) {
        const uint32_t key = rk == SPL_NO_RANK ? 0xFFFFFFFFu : ((rk << 6) | (uint32_t)lane);
        uint32_t m = row16_min(key);
        const uint32_t r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
        const uint32_t r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
        const uint32_t a = r0 < r1 ? r0 : r1, c = r2 < r3 ? r2 : r3;
        m = a < c ? a : c;                                  // wave-uniform
        if (m == 0xFFFFFFFFu) break;
        const int mi = (int)(m & 63u);
        const uint32_t mn = m >> 6;
        const unsigned long long above = alive & ~((2ull << mi) - 1ull);
        const int j = __builtin_ctzll(above);
        const unsigned long long above2 = above & (above - 1ull);
        const int j2 = above2 ? __builtin_ctzll(above2) : -1;
        const unsigned long long above3 = above2 & (above2 - 1ull);
        const int e_r = above3 ? __builtin_ctzll(above3) : end;
        const int e_mi = j2 >= 0 ? j2 : end;
        const unsigned long long below = alive & ((1ull << mi) - 1ull);
        const int h = below ? 63 - __builtin_clzll(below) : -1;
        const int len_r = e_r - mi, len_h = e_mi - h;
        const uint32_t id_j2 = (j2 >= 0 && len_r > SUB_LMAX) ? __builtin_amdgcn_readlane(idv, j2) : 0u;
        if (lane == mi) {
            idv = mn;
            rk = j2 < 0 ? SPL_NO_RANK : len_r <= SUB_LMAX ? row[len_r - 2] : len_r <= far_max ? pair_rank(T, mn, id_j2) : SPL_NO_RANK;
        } else if (lane == h) {
            rk = len_h <= SUB_LMAX ? row[len_h - 2] : len_h <= far_max ? pair_rank(T, idv, mn) : SPL_NO_RANK;
        } else if (lane == j) {
            rk = SPL_NO_RANK;
        }
        alive &= ~(1ull << j);
    }
    if (((alive >> lane) & 1ull) && idv != SPL_NO_RANK) emit(lane, idv);
}

// One chunk of 17..64 bytes per WAVEFRONT with tabulated pair ranks (see bpe_group16_tab): everything per-chunk is wave-uniform (the minimum, the
// alive bitmap, the neighbour indices) and lives in scalar registers; one chunk of 17..64 bytes per
// wavefront, lane i owns node i and the ids of text[i, i+len), len = 2..8, in its LDS row.
template <class Emit>
__device__ __forceinline__ void bpe_wave64_tab(const DeviceTables& T, const LdsAcc& tx, int p, int n, uint32_t* sub,
                                               Emit emit) {
    const int lane = tidx() & 63;
    const bool own = lane < n;
    const int maxlen = own ? (n - lane < SUB_LMAX ? n - lane : SUB_LMAX) : 0;
    const uint32_t w0 = own ? tx.load32(p + lane) : 0u;
    const uint32_t w1 = own ? tx.load32(p + lane + 4) : 0u;
    uint32_t id = own ? T.byte_id[w0 & 0xFFu] : SPL_DEAD;
    uint32_t* row = sub + lane * SUB_W;
    // which token lengths exist at all behind the lane's first two bytes: the other probes go to the
    // spare bucket (28 % fewer table lines for English text, 85 % for CJK)
    const RowHead rh = row_head(T, own, w0, maxlen);
    const uint32_t lm = rh.lm;
    {
        uint32_t r[7];
        row_fill(T, rh, w0, w1, maxlen, r);
        if (maxlen >= 2) {
#pragma unroll
            for (int k = 0; k < SUB_W; k++) row[k] = r[k];
        }
    }
    const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    // Independent segments.  A merge never crosses a byte boundary that no token spans, so the
    // stretches between such boundaries merge independently of each other -- and a chunk of CJK
    // text is mostly such boundaries (few tokens span two characters).  Lane i knows the longest
    // token starting at byte i (its table row; beyond 8 bytes the bound of the p8 table); the
    // running maximum of "last byte covered" says which boundaries nothing spans.  Segments of up
    // to 16 bytes then go through the 16-lane loop four at a time, on the rows already filled:
    // a few short loops side by side instead of one loop over every merge of the chunk.
    unsigned long long starts = 1ull;
#ifndef SPL_SEG_ASCII
#define SPL_SEG_ASCII 0          /* 1: look for independent segments in ASCII chunks too (A/B) */
#endif
    // the longest token of more than 8 bytes that can start at this lane's byte (p8 table: an upper bound)
    int l8 = 0;
    if (own && n - lane > SUB_LMAX && (lm & 0x80u)) {
        const P8Bucket e8 = T.p8_tab[hash_p8(w0, w1) & T.p8_mask];
        l8 = (int)p8_match(e8.a, e8.b, p8_tag(w0, w1));
    }
    const int far_max = l8 == 255 ? FAR_UNBOUNDED : l8;
    if (SPL_SEG_ASCII || __any(own && (w0 & 0x80u))) {
        int ml = 1;
#pragma unroll
        for (int k = 0; k < SUB_W; k++) ml = (k + 2 <= maxlen && row[k] != SPL_NO_RANK) ? k + 2 : ml;
        if (own && n - lane > SUB_LMAX) {
            const int cap = n - lane;
            ml = l8 == 0 ? ml : (l8 == 255 || l8 > cap) ? cap : l8;
        }
        const uint32_t cover = wave_scan_max(own ? (uint32_t)(lane + ml - 1) : 0u);
        starts = ((__ballot(own && cover == (uint32_t)lane) << 1) | 1ull) & all;
    }
    wave_lds_sync();                                           // rows are read across lanes from here on
    const int gl = lane & 15, g = lane >> 4;
    unsigned long long rem = starts, longsegs = 0;             // longsegs: starts of segments beyond 16 bytes
    if (starts == 1ull) { rem = 0; longsegs = 1ull; }          // (the usual case: one segment, the whole chunk)
    while (rem) {
        int gs = 0, glen = 0;                                  // this 16-lane group's segment
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (rem) {
                const int sk = __builtin_ctzll(rem);
                rem &= rem - 1ull;
                const int ek = rem ? __builtin_ctzll(rem) : n;
                if (ek - sk > 16) longsegs |= 1ull << sk;
                else if (g == k) { gs = sk; glen = ek - sk; }
            }
        }
        const uint32_t gid = __shfl(id, gs + gl);
        const int gfar = __shfl(far_max, gs + gl);
        group16_merge(T, sub + (gl < glen ? gs + gl : 0) * SUB_W, gl < glen ? gid : SPL_DEAD, glen, gfar,
                      [&](int i, uint32_t tid_) { emit(gs + i, tid_); });
    }
    while (longsegs) {
        const int sk = __builtin_ctzll(longsegs);
        longsegs &= longsegs - 1ull;
        const unsigned long long later = starts & ~((2ull << sk) - 1ull);
        const int ek = later ? __builtin_ctzll(later) : n;
        const unsigned long long seg = (ek >= 64 ? ~0ull : ((1ull << ek) - 1ull)) & ~((1ull << sk) - 1ull);
        wave64_merge(T, row, seg, ek, (lane >= sk && lane + 1 < ek) ? row[0] : SPL_NO_RANK, id, far_max, emit);
    }
}


// ------------------------------------------------------------------------------------------
// Global-memory accessor: class records computed on the fly (slow path, rare).
struct GlobalAcc {
    const DeviceTables* T;
    const Batch* b;
    __device__ uint32_t txt(int64_t q) const { return q < (int64_t)b->n_bytes ? b->text[q] : 0u; }
    __device__ uint32_t txt(int q) const { return txt((int64_t)(uint32_t)q); }
    __device__ uint32_t load32(int p) const {
        const int64_t q = (uint32_t)p;
        return txt(q) | (txt(q + 1) << 8) | (txt(q + 2) << 16) | (txt(q + 3) << 24);
    }
    __device__ uint32_t rec(int qi) const {
        const int64_t q = (uint32_t)qi;
        const int64_t B = b->n_bytes;
        if (q >= B) return C_EOT | CB_TSTART | CB_SYNC;
        if (b->skip && ((b->skip[q >> 5] >> (q & 31)) & 1u)) return C_EOT | CB_TSTART;
        const uint32_t* const tsb = b->tstart;
        uint32_t r = byte_record(*T, *this, [&](int i) { return ((tsb[(uint32_t)i >> 5] >> (i & 31)) & 1u) != 0; },
                                 [&](uint32_t c) { return cp_class(*T, c); }, qi, 0, (int)B);
        if ((b->tstart[q >> 5] >> (q & 31)) & 1u) r |= CB_TSTART | CB_SYNC;
        return r;
    }
};

// GlobalAcc for the single-pass kernel, which has no text-start bitmap in HBM: a chain that is
// continued beyond the window stops at the first text start after its own start, so that ONE
// position (found once by a search of doc_off) stands in for the bitmap.
struct DirectAcc {
    const DeviceTables* T;
    const Batch* b;
    uint32_t next_ts;          // first text start after the chain's start (n_bytes if none)
    uint32_t lo;               // the chain's start (a character start: nothing before it matters)
    __device__ __forceinline__ uint32_t txt(int64_t q) const { return q < (int64_t)b->n_bytes ? b->text[q] : 0u; }
    __device__ __forceinline__ uint32_t txt(int q) const { return txt((int64_t)(uint32_t)q); }
    __device__ __forceinline__ uint32_t load32(int p) const {
        const int64_t q = (uint32_t)p;
        return txt(q) | (txt(q + 1) << 8) | (txt(q + 2) << 16) | (txt(q + 3) << 24);
    }
    __device__ __forceinline__ uint32_t rec(int qi) const {
        const int64_t q = (uint32_t)qi;
        const int64_t B = b->n_bytes;
        if (q >= B) return C_EOT | CB_TSTART | CB_SYNC;
        if (b->skip && ((b->skip[q >> 5] >> (q & 31)) & 1u)) return C_EOT | CB_TSTART;          // inside a special literal
        const uint32_t* const tsb = b->tstart;
        const uint32_t nts = next_ts;
        uint32_t r = byte_record(*T, *this,
                                 [&](int i) { return (uint32_t)i == nts || (tsb && ((tsb[(uint32_t)i >> 5] >> (i & 31)) & 1u)); },
                                 [&](uint32_t c) { return cp_class(*T, c); }, qi, (int)lo, (int)B);
        if ((uint32_t)q == next_ts) r |= CB_TSTART | CB_SYNC;
        if (b->tstart && ((b->tstart[q >> 5] >> (q & 31)) & 1u)) r |= CB_TSTART | CB_SYNC;    // behind a special literal
        return r;
    }
};

__device__ __forceinline__ void emit_token(const Batch& b, uint32_t pos, uint32_t id) {
    b.stage[pos] = id;
    atomicOr(&b.tbits[pos >> 5], 1u << (pos & 31));
}

// Chains that outgrew a tile window, ONE WAVEFRONT per chain: there are few such chains (tens per
// 40 MB) but each is long, and a lane that walks it byte by byte from HBM pays a memory round trip
// per character.  Here the
// 64 lanes stage a window of DEFER_WIN bytes and its class records in LDS (classified in parallel,
// as k_pretok does), lane 0 runs the scanner over LDS, and the window is moved along the chain.
// A single chunk longer than the window falls back to the byte-wise walk.
constexpr int DEFER_WIN = 2048;
constexpr int DEFER_BACK = 4;                 // bytes staged before the start (previous character's class)
struct WinAcc {
    const uint8_t* rec_;
    const uint8_t* txt_;
    int n_;                                   // staged records; beyond: window end
    __device__ __forceinline__ uint32_t rec(int q) const { return q < n_ ? (uint32_t)rec_[q] : (uint32_t)C_WEND; }
    __device__ __forceinline__ uint32_t txt(int q) const { return txt_[q]; }
    __device__ __forceinline__ uint32_t load32(int p) const {
        return (uint32_t)txt_[p] | ((uint32_t)txt_[p + 1] << 8) | ((uint32_t)txt_[p + 2] << 16) | ((uint32_t)txt_[p + 3] << 24);
    }
};
__global__ __launch_bounds__(64) void k_deferred_wave(DeviceTables T, Batch b) {
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[DEFER_WIN + 32];
    __shared__ uint8_t s_rec[DEFER_WIN + 32];
    __shared__ uint8_t s_ascii[128];
    const int lane = tidx();
    const uint32_t nq = min(b.qcount[3], b.qcapdefer);
    const int64_t B = b.n_bytes;
    for (int k = lane; k < 128; k += 64) s_ascii[k] = T.ucls_stage2[((uint32_t)T.ucls_stage1[0] << T.ucls_shift) + k];
    for (uint32_t it = blockIdx.x; it < nq; it += gridDim.x) {
        const uint32_t pent = b.qdefer[it];
        int64_t p = pent & 0x7FFFFFFFu;                         // wave-uniform: start of the next chunk
        bool first_chunk = !(pent >> 31);                       // (bit 31: a chunk starts there only if it is no sync point)
        for (;;) {                                              // one window per pass
            if (p >= B) break;
            const int64_t base = p >= DEFER_BACK ? p - DEFER_BACK : 0;
            const int q0 = (int)(p - base);
            const int nst = (int)((B - base) < (int64_t)(DEFER_WIN + 16) ? (B - base) : (int64_t)(DEFER_WIN + 16));   // staged text bytes
            const int nrec = nst < DEFER_WIN ? nst + 1 : DEFER_WIN;     // records (one past the text = end of text)
            wave_lds_sync();
            for (int i = lane; i < DEFER_WIN + 32; i += 64) s_txt[i] = i < nst ? b.text[base + i] : (uint8_t)0;
            wave_lds_sync();
            for (int i = lane; i < nrec; i += 64) {
                const int64_t g = base + i;
                uint32_t r;
                if (g >= B) r = C_EOT | CB_TSTART | CB_SYNC;
                else if (b.skip && ((b.skip[g >> 5] >> (g & 31)) & 1u)) r = C_EOT | CB_TSTART;
                else {
                    // (window index i = global position base + i; look-back stops at the window's first byte:
                    //  DEFER_BACK bytes precede the chain's start, which is a character start anyway)
                    const WinAcc tx{s_rec, s_txt, 0};
                    r = byte_record(T, tx, [&](int k) { const int64_t gg = base + k; return ((b.tstart[gg >> 5] >> (gg & 31)) & 1u) != 0; },
                                    [&](uint32_t c) { return (uint32_t)s_ascii[c]; }, i, 0, nst);
                    if ((b.tstart[g >> 5] >> (g & 31)) & 1u) r |= CB_TSTART | CB_SYNC;
                }
                s_rec[i] = (uint8_t)r;
            }
            wave_lds_sync();
            // lane 0 walks the chain inside the window; state back to the wavefront through LDS-free
            // broadcasts: next position, and whether the chain is finished
            int64_t np = p;
            int done = 0, fallback = 0;
            if (lane == 0) {
                const WinAcc acc{s_rec, s_txt, nrec};
                int q = q0;
                bool fc = first_chunk;
                for (;;) {
                    if (!fc) {                                          // does a chunk start here at all?
                        const uint32_t r = acc.rec(q);
                        if (r == (uint32_t)C_WEND) { np = base + q; break; }       // need the next window to tell
                        if (r & (CB_SYNC | CB_TSTART)) { done = 1; break; }
                        int j = q - 1;
                        while (j > 0 && (acc.rec(j) & CB_CLASS) == C_CONT && j > q - 4) j--;
                        const uint32_t prev = acc.rec(j) & CB_CLASS;
                        if (prev < C_EOT && is_sync((int)T.pattern, prev, r & CB_CLASS)) { done = 1; break; }
                    }
                    const int e = match_end(acc, q, (int)T.pattern);
                    if (e == SPL_DEFER) {
                        if (q == q0) fallback = 1;                      // longer than a whole window
                        np = base + q;
                        break;
                    }
                    fc = false;
                    const uint32_t gp = (uint32_t)(base + q), n = (uint32_t)(e - q);
                    const uint32_t id = probe_chunk(T, acc, q, (int)n);
                    if (id != SPL_NO_RANK) emit_token(b, gp, id);
                    else if (n > 1) push_long(b, gp, n);
                    q = e;
                    np = base + q;
                    if (np >= B) { done = 1; break; }
                }
                if (fallback) {                                         // one chunk, byte-wise from HBM
                    const GlobalAcc ga{&T, &b};
                    const uint32_t gp = (uint32_t)np;
                    const int e = match_end(ga, (int)gp, (int)T.pattern);
                    const uint32_t n = (uint32_t)e - gp;
                    const uint32_t id = probe_chunk(T, ga, (int)gp, (int)n);
                    if (id != SPL_NO_RANK) emit_token(b, gp, id);
                    else if (n > 1) push_long(b, gp, n);
                    np = (int64_t)(uint32_t)e;
                    if (np >= B) done = 1;
                }
            }
            const uint32_t np_lo = __builtin_amdgcn_readfirstlane((uint32_t)np);
            done = __builtin_amdgcn_readfirstlane(done);
            // (a window that made no progress can only be the "need the next window" case right at its
            //  start, which cannot happen: DEFER_BACK + 1 records are always staged before the end)
            first_chunk = (int64_t)np_lo == p ? first_chunk : false;
            p = (int64_t)np_lo;
            if (done) break;
        }
    }
}

// ------------------------------------------------------------------------------------------
// byte_pair_encode, ONE LANE PER CHUNK (17..64 bytes), for large batches: node arrays interleaved
// in LDS (node-major, lane-minor: conflict-free when lanes touch the same node index), merge loop
// = bpe_serial (spl_lookup.h).  Slow per chunk, but every lane carries its own chain of dependent
// pair-table probes, so a CU keeps hundreds of them in flight.
constexpr int GROUP_NMAX = 128;       // 16 lanes x 8 register slots
constexpr int WAVE_NMAX = 512;
constexpr uint32_t NIL16 = 0xFFFFu;


template <class Emit>
__device__ __forceinline__ void bpe_wave(const DeviceTables& T, const Batch& b, uint32_t pos, int n, uint32_t* s_id,
                                         uint32_t* s_rk, uint16_t* s_nx, uint16_t* s_pv, Emit emit) {
    const int lane = tidx() & 63;
    for (int i = lane; i < n; i += 64) {
        s_id[i] = T.byte_id[b.text[pos + i]];
        s_nx[i] = (uint16_t)(i + 1 < n ? i + 1 : (int)NIL16);
        s_pv[i] = (uint16_t)(i > 0 ? i - 1 : (int)NIL16);
    }
    wave_lds_sync();
    for (int i = lane; i < n; i += 64) s_rk[i] = (i + 1 < n) ? pair_rank(T, s_id[i], s_id[i + 1]) : SPL_NO_RANK;
    wave_lds_sync();
    for (;;) {
        uint32_t key = 0xFFFFFFFFu;
        for (int i = lane; i < n; i += 64) {
            const uint32_t r = s_rk[i];
            const uint32_t k = r == SPL_NO_RANK ? 0xFFFFFFFFu : ((r << 9) | (uint32_t)i);
            key = k < key ? k : key;
        }
        uint32_t m = row16_min(key);
        const uint32_t r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
        const uint32_t r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
        const uint32_t a = r0 < r1 ? r0 : r1, c = r2 < r3 ? r2 : r3;
        m = a < c ? a : c;
        if (m == 0xFFFFFFFFu) break;
        const uint32_t mi = m & 511u, mn = m >> 9;
        const uint32_t j = s_nx[mi];                       // uniform addresses: LDS broadcasts
        const uint32_t j2 = s_nx[j];
        const uint32_t h = s_pv[mi];
        const uint32_t id_j2 = j2 != NIL16 ? s_id[j2] : 0u;
        const uint32_t id_h = h != NIL16 ? s_id[h] : 0u;
        wave_lds_sync();
        if (lane == 0) {
            s_id[mi] = mn;
            s_id[j] = SPL_DEAD;
            s_rk[j] = SPL_NO_RANK;
            s_nx[mi] = (uint16_t)j2;
            if (j2 != NIL16) s_pv[j2] = (uint16_t)mi;
        } else if (lane == 1) {
            s_rk[mi] = j2 != NIL16 ? pair_rank(T, mn, id_j2) : SPL_NO_RANK;
        } else if (lane == 2) {
            if (h != NIL16) s_rk[h] = pair_rank(T, id_h, mn);
        }
        wave_lds_sync();
    }
    for (int i = lane; i < n; i += 64) {
        const uint32_t id = s_id[i];
        if (id != SPL_DEAD && id != SPL_NO_RANK) emit(pos + (uint32_t)i, id);
    }
    wave_lds_sync();
}

// bpe_block_lds: chunks of up to BLOCK_LDS_NMAX bytes by the WHOLE workgroup with the node list
// in LDS (the layout of bpe_wave, capacity `cap` nodes): every thread scans its nodes for the
// minimum, a workgroup min-reduction picks the leftmost one, one thread relinks while two others
// (in other wavefronts) re-rank the two affected pairs.  Three barriers and one memory round trip
// per merge.
constexpr int BLOCK_LDS_NMAX = 2048;      // index bits in the reduction key
template <class Emit>
__device__ __forceinline__ void bpe_block_lds(const DeviceTables& T, const Batch& b, uint32_t pos, int n, uint32_t* s_id,
                                              uint32_t* s_rk, uint16_t* s_nx, uint16_t* s_pv, uint32_t* s_red4, Emit emit) {
    const int tid = tidx();
    for (int i = tid; i < n; i += NT) {
        s_id[i] = T.byte_id[b.text[pos + i]];
        s_nx[i] = (uint16_t)(i + 1 < n ? i + 1 : (int)NIL16);
        s_pv[i] = (uint16_t)(i > 0 ? i - 1 : (int)NIL16);
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) s_rk[i] = (i + 1 < n) ? pair_rank(T, s_id[i], s_id[i + 1]) : SPL_NO_RANK;
    __syncthreads();
    for (;;) {
        uint32_t key = 0xFFFFFFFFu;
        for (int i = tid; i < n; i += NT) {
            const uint32_t r = s_rk[i];
            const uint32_t k = r == SPL_NO_RANK ? 0xFFFFFFFFu : ((r << 11) | (uint32_t)i);
            key = k < key ? k : key;
        }
        uint32_t m = row16_min(key);
        const uint32_t r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
        const uint32_t r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
        const uint32_t a = r0 < r1 ? r0 : r1, c = r2 < r3 ? r2 : r3;
        m = a < c ? a : c;
        if ((tid & 63) == 0) s_red4[tid >> 6] = m;
        __syncthreads();
        m = s_red4[0];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) m = s_red4[w] < m ? s_red4[w] : m;
        if (m == 0xFFFFFFFFu) break;
        const uint32_t mi = m & 2047u, mn = m >> 11;
        const uint32_t j = s_nx[mi];
        const uint32_t j2 = s_nx[j];
        const uint32_t h = s_pv[mi];
        const uint32_t id_j2 = j2 != NIL16 ? s_id[j2] : 0u;
        const uint32_t id_h = h != NIL16 ? s_id[h] : 0u;
        __syncthreads();
        if (tid == 0) {
            s_id[mi] = mn;
            s_id[j] = SPL_DEAD;
            s_rk[j] = SPL_NO_RANK;
            s_nx[mi] = (uint16_t)j2;
            if (j2 != NIL16) s_pv[j2] = (uint16_t)mi;
        } else if (tid == 64) {
            s_rk[mi] = j2 != NIL16 ? pair_rank(T, mn, id_j2) : SPL_NO_RANK;
        } else if (tid == 128) {
            if (h != NIL16) s_rk[h] = pair_rank(T, id_h, mn);
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += NT) {
        const uint32_t id = s_id[i];
        if (id != SPL_DEAD && id != SPL_NO_RANK) emit(pos + (uint32_t)i, id);
    }
    __syncthreads();
}

// bpe_block_rounds: chunks beyond the LDS capacities (pathological single-class runs of any
// length): one workgroup per chunk, nodes in HBM scratch in their original slots (ids in stage[],
// pair ranks in rank_scr[], merged-away slots are tomb-stones).  The reference's loop
// (src/core/bpe.rs:118-190) takes the leftmost pair of minimal rank, one merge at a time; here one
// ROUND takes EVERY pair of the minimal rank m at once -- in a run of consecutive pairs of rank m
// the 1st, 3rd, ... (what leftmost-first leaves of such a run) -- which is the same sequence of
// merges as long as no merge creates a pair of rank <= m.  That is checked, not assumed: each
// selected merge looks up the two pairs it creates (left: with the final left neighbour; right:
// with the still unmerged right neighbour, the state the sequential order passes through), the
// leftmost merge whose new pair ranks <= m ends the round, and only the merges up to it are
// committed.  64 KB of one character takes ~15 rounds instead of ~60 000 merges.
// Four coalesced passes over the slots per round, each wavefront on a contiguous quarter:
//   A  minimum rank m                        C  neighbours + new ranks of the selected -> aux[]
//   B  selection by parity inside runs       D  commit (writes only what aux[] says)
// Selection marks live in rank_scr (RK_SEL bit); a selected pair (a, b) owns aux[2a], aux[2a+1],
// aux[2b], aux[2b+1], so pass D needs no neighbour search while ids and ranks change under it.
constexpr uint32_t RK_DEAD = 0xFFFFFFFEu;     // rank slot of a merged-away node
constexpr uint32_t RK_SEL = 0x40000000u;      // rank slot: selected for this round
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long mask) {      // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ void bpe_rounds_core(const uint8_t* text, const uint32_t* byte_id, const uint64_t* pair_tab,
                                                uint32_t pair_mask, uint32_t* ids, uint32_t* rks, uint32_t* aux, const int n,
                                                uint32_t* s_red4, int& n_out, const uint32_t*& pos_out) {
    const int tid = tidx(), lane = tid & 63, wv = tid >> 6;
    DeviceTables T{};
    T.pair_tab = pair_tab;
    T.pair_mask = pair_mask;
#pragma nounroll
    for (int i = tid; i < n; i += NT) ids[i] = byte_id[text[i]];
    __syncthreads();
#pragma nounroll
    for (int i = tid; i < n; i += NT) rks[i] = (i + 1 < n) ? pair_rank(T, ids[i], ids[i + 1]) : SPL_NO_RANK;
    __syncthreads();
    // The slots are compacted whenever half of them are tomb-stones (every pass of a round walks all
    // slots): ncur slots in use, posbuf[i] = original offset of slot i's node once that differs from i.
    // aux: [0, n) scratch of passes C / D and of the compaction, [n, 2n) two position arrays in turn.
    int ncur = n, flip = 0;
    const uint32_t* posbuf = nullptr;
    for (;;) {
        const int groups = (ncur + 63) >> 6, gw = (groups + NT / 64 - 1) / (NT / 64);
        const int g0 = wv * gw, g1 = g0 + gw < groups ? g0 + gw : groups;
        // A: the minimal rank (and how many slots are alive)
        uint32_t key = SPL_NO_RANK, wcnt = 0;
#pragma unroll 4
        for (int g = g0; g < g1; g++) {
            const int i = g * 64 + lane;
            const uint32_t r = i < ncur ? rks[i] : RK_DEAD;
            key = r < key ? r : key;
            wcnt += (uint32_t)__popcll(__ballot(r != RK_DEAD));
        }
        uint32_t m = row16_min(key);
        {
            const uint32_t r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
            const uint32_t r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
            const uint32_t x = r0 < r1 ? r0 : r1, y = r2 < r3 ? r2 : r3;
            m = x < y ? x : y;
        }
        if (lane == 0) s_red4[wv] = m;
        __syncthreads();
        m = s_red4[0];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) m = s_red4[w] < m ? s_red4[w] : m;
        if (m >= RK_DEAD) break;
        if (ncur > 4096) {
            __syncthreads();
            if (lane == 0) s_red4[wv] = wcnt;
            __syncthreads();
            uint32_t total = 0, base = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; w++) { total += s_red4[w]; base += w < wv ? s_red4[w] : 0u; }
            __syncthreads();
            if (2 * total <= (uint32_t)ncur) {
                uint32_t* const t_id = aux;
                uint32_t* const t_rk = aux + total;
                uint32_t* const npos = aux + n + (flip ? (n + 1) / 2 : 0);
                for (int g = g0; g < g1; g++) {
                    const int i = g * 64 + lane;
                    const uint32_t r = i < ncur ? rks[i] : RK_DEAD;
                    const unsigned long long al = __ballot(r != RK_DEAD);
                    if (r != RK_DEAD) {
                        const uint32_t d = base + mbcnt64(al);
                        t_id[d] = ids[i];
                        t_rk[d] = r;
                        npos[d] = posbuf ? posbuf[i] : (uint32_t)i;
                    }
                    base += (uint32_t)__popcll(al);
                }
                __syncthreads();
                for (uint32_t k = (uint32_t)tid; k < total; k += NT) { ids[k] = t_id[k]; rks[k] = t_rk[k]; }
                __syncthreads();
                posbuf = npos;
                flip ^= 1;
                ncur = (int)total;
                continue;                                    // (the next turn finds the same minimum among fewer slots)
            }
        }
        const uint32_t msel = m | RK_SEL;
        // B: selection.  carry = alive nodes of rank m immediately before the group (its parity counts)
        uint32_t carry = 0;
        for (int g = g0 - 1; g >= 0 && g0 < g1; g--) {            // the run entering this quarter
            const uint32_t r = rks[g * 64 + lane];                // (other wavefronts may be marking: m or msel)
            const unsigned long long alive = __ballot(r != RK_DEAD), eqm = __ballot((r & ~RK_SEL) == m);
            const unsigned long long noneq = alive & ~eqm;
            if (noneq == 0) { carry += (uint32_t)__popcll(alive); continue; }
            const int hb = 63 - __builtin_clzll(noneq);
            carry += (uint32_t)__popcll((alive >> hb) >> 1);
            break;
        }
        for (int g = g0; g < g1; g++) {
            const int i = g * 64 + lane;
            const uint32_t r = i < ncur ? rks[i] : RK_DEAD;
            const unsigned long long alive = __ballot(r != RK_DEAD), eqm = __ballot(r == m);
            const unsigned long long noneq = alive & ~eqm;
            uint32_t off = mbcnt64(alive);                       // alive nodes below this lane in the group
            if (mbcnt64(noneq) == 0) off += carry;               // the run comes in from the previous group
            else {
                int l2 = lane;
                asm volatile("" : "+v"(l2));                     // (keeps the lane mask out of long-lived registers)
                const unsigned long long below = (1ull << l2) - 1ull;
                off -= (uint32_t)__popcll(alive & ((2ull << (63 - __builtin_clzll(noneq & below))) - 1ull));
            }
            if (r == m && !(off & 1u)) rks[i] = msel;
            if (noneq == 0) carry += (uint32_t)__popcll(alive);
            else carry = (uint32_t)__popcll((alive >> (63 - __builtin_clzll(noneq))) >> 1);
        }
        __syncthreads();
        // C: neighbours and new ranks of every selected merge; F = leftmost one that ends the round
        uint32_t fail = 0xFFFFFFFFu;
        for (int g = g0; g < g1; g++) {
            const int i = g * 64 + lane;
            const uint32_t r = i < ncur ? rks[i] : RK_DEAD;
            if (r == msel) {
                uint32_t bb = (uint32_t)i + 1;
                while (rks[bb] == RK_DEAD) bb++;                 // exists: slot i has a rank
                uint32_t idl = SPL_NO_RANK, idc = SPL_NO_RANK;   // ids left and right of the new token (none: no pair)
                {
                    int h = i - 1;
                    while (h >= 0 && rks[h] == RK_DEAD) h--;
                    uint32_t leftw = 0xFFFFFFFFu;                // whose slot holds the left pair's rank
                    if (h >= 0) {
                        int hh = h - 1;
                        while (hh >= 0 && rks[hh] == RK_DEAD) hh--;
                        if (hh >= 0 && rks[hh] == msel) idl = m;                 // h merges into hh first
                        else { leftw = (uint32_t)h; idl = ids[h]; }
                    }
                    aux[2 * bb] = leftw;
                    uint32_t c = bb + 1;
                    while (c < (uint32_t)ncur && rks[c] == RK_DEAD) c++;
                    uint32_t csel = 0xFFFFFFFFu;
                    if (c < (uint32_t)ncur) { idc = ids[c]; csel = rks[c] == msel ? c : csel; }
                    aux[2 * bb + 1] = csel;
                }
#pragma nounroll
                for (int side = 0; side < 2; side++) {           // (one lookup site: registers)
                    const uint32_t q = pair_rank(T, side ? m : idl, side ? idc : m);
                    aux[2 * i + side] = q;
                    if (q <= m) fail = (uint32_t)i < fail ? (uint32_t)i : fail;
                }
            }
        }
        fail = row16_min(fail);
        {
            const uint32_t r0 = __builtin_amdgcn_readlane(fail, 0), r1 = __builtin_amdgcn_readlane(fail, 16);
            const uint32_t r2 = __builtin_amdgcn_readlane(fail, 32), r3 = __builtin_amdgcn_readlane(fail, 48);
            const uint32_t x = r0 < r1 ? r0 : r1, y = r2 < r3 ? r2 : r3;
            fail = x < y ? x : y;
        }
        if (lane == 0) s_red4[wv] = fail;
        __syncthreads();
        fail = s_red4[0];
#pragma unroll
        for (int w = 1; w < NT / 64; w++) fail = s_red4[w] < fail ? s_red4[w] : fail;
        // D: commit the merges up to `fail`; the others lose their mark
        for (int g = g0; g < g1; g++) {
            const int i = g * 64 + lane;
            const uint32_t r = i < ncur ? rks[i] : RK_DEAD;
            if (r == msel) {
                if ((uint32_t)i > fail) { rks[i] = m; continue; }
                uint32_t bb = (uint32_t)i + 1;
                while (rks[bb] == RK_DEAD) bb++;                 // only this lane ever writes slot bb
                const uint32_t ql = aux[2 * i], qr = aux[2 * i + 1], leftw = aux[2 * bb], csel = aux[2 * bb + 1];
                uint32_t nr = qr;
                if (csel != 0xFFFFFFFFu && csel <= fail) nr = aux[2 * (size_t)csel];   // the right neighbour merges too
                ids[i] = m;
                ids[bb] = SPL_DEAD;
                rks[bb] = RK_DEAD;
                rks[i] = nr;
                if (leftw != 0xFFFFFFFFu) rks[leftw] = ql;
            }
        }
        __syncthreads();
    }
    n_out = ncur;
    pos_out = posbuf;
}
template <class Emit>
__device__ __forceinline__ void bpe_block_rounds(const DeviceTables& T, const Batch& b, uint32_t pos, int n, uint32_t* s_red4,
                                                 Emit emit) {
    uint32_t* ids = b.stage + pos;
    uint32_t* aux = b.aux + 2 * (size_t)pos;
    int nc = n;
    const uint32_t* posbuf = nullptr;
    bpe_rounds_core(b.text + pos, T.byte_id, T.pair_tab, T.pair_mask, ids, b.rank_scr + pos, aux, n, s_red4, nc, posbuf);
    const uint32_t* from = ids;
    if (posbuf) {                                         // compacted: the ids leave stage[] before tokens are written there
        for (int i = tidx(); i < nc; i += NT) aux[i] = ids[i];
        __syncthreads();
        from = aux;
    }
    for (int i = tidx(); i < nc; i += NT) {          // survivors become tokens
        const uint32_t id = from[i];
        if (id != SPL_DEAD && id != SPL_NO_RANK) emit(pos + (posbuf ? posbuf[i] : (uint32_t)i), id);
    }
    __syncthreads();
}

// Longer chunks (up to 64 * NPL bytes) by ONE wavefront with tabulated pair ranks: node i lives in
// lane i % 64, slot i / 64, and row i of the wavefront's LDS table holds the ids of
// text[i, i+len), len = 2..8 (see bpe_group16_tab).  The alive bitmap is wave-uniform (scalar
// registers); a merge costs one min-reduction and LDS reads of the two affected rows -- no memory
// round trip unless a merged token is longer than 8 bytes.  `word_at(q)` returns the 4 text bytes
// at chunk offset q (little endian; bytes past the chunk may be anything).  `sub` holds 64 * NPL
// rows of SUB_W words followed by 64 * NPL words for the initial ids.
// The table is filled ONE probe per pass of a plain loop over (slot, length): any batching of the
// probe code inside a loop makes the register allocator need 150-220 VGPRs.
template <int NW> __device__ __forceinline__ int next_set64(const unsigned long long (&a)[NW], int from) {
    int res = -1;                                     // lowest set bit with index >= from
#pragma unroll
    for (int w = NW - 1; w >= 0; w--) {
        unsigned long long x = a[w];
        const int lo = from - 64 * w;
        if (lo >= 64) x = 0;
        else if (lo > 0) x &= ~((1ull << lo) - 1ull);
        if (x) res = 64 * w + __builtin_ctzll(x);
    }
    return res;
}
template <int NW> __device__ __forceinline__ int prev_set64(const unsigned long long (&a)[NW], int before) {
    int res = -1;                                     // highest set bit with index < before
#pragma unroll
    for (int w = 0; w < NW; w++) {
        unsigned long long x = a[w];
        const int hi = before - 64 * w;
        if (hi <= 0) x = 0;
        else if (hi < 64) x &= (1ull << hi) - 1ull;
        if (x) res = 64 * w + 63 - __builtin_clzll(x);
    }
    return res;
}
template <int NPL, class IdAt, class Emit>
__device__ __forceinline__ void wave_tab_merge(const DeviceTables& T, int n, const uint32_t* sub, IdAt id_at, Emit emit);
template <int NPL, class WordAt, class Emit>
__device__ __forceinline__ void bpe_wave_tab(const DeviceTables& T, int n, uint32_t* sub, WordAt word_at, Emit emit) {
    const int lane = tidx() & 63;
    const int slots = (n + 63) >> 6;
#pragma nounroll
    for (int job = 0; job < slots * SUB_W; job++) {
        const int k = job / SUB_W, len = 2 + job % SUB_W;
        const int i = lane + 64 * k;
        if (i + len <= n) {
            const uint32_t k0 = mask_tail(word_at(i), len);
            const uint32_t k1 = len > 4 ? mask_tail(word_at(i + 4), len - 4) : 0u;
            sub[i * SUB_W + len - 2] = probe_short(T, k0, k1, 0u, (uint32_t)len);
        }
    }
    wave_tab_merge<NPL>(T, n, sub, [&](int i) { return T.byte_id[word_at(i) & 0xFFu]; }, emit);
}
// The merge loop of bpe_wave_tab over a filled table: `sub` is row 0, id_at(i) the id of byte i.
template <int NPL, class IdAt, class Emit>
__device__ __forceinline__ void wave_tab_merge(const DeviceTables& T, int n, const uint32_t* sub, IdAt id_at, Emit emit) {
    const int lane = tidx() & 63;
    uint32_t id[NPL], rk[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int i = lane + 64 * k;
        id[k] = i < n ? id_at(i) : SPL_DEAD;
        rk[k] = (i + 1 < n) ? sub[i * SUB_W] : SPL_NO_RANK;
    }
    unsigned long long alive[NPL];
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int c = n - 64 * k;
        alive[k] = c >= 64 ? ~0ull : c > 0 ? (1ull << c) - 1ull : 0ull;
    }
    for (;;) {
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NPL; k++) {
            const uint32_t c = rk[k] == SPL_NO_RANK ? 0xFFFFFFFFu : ((rk[k] << 8) | (uint32_t)(lane + 64 * k));
            key = c < key ? c : key;
        }
        uint32_t m = row16_min(key);
        const uint32_t r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16);
        const uint32_t r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
        const uint32_t a = r0 < r1 ? r0 : r1, c = r2 < r3 ? r2 : r3;
        m = a < c ? a : c;                                  // wave-uniform
        if (m == 0xFFFFFFFFu) break;
        const int mi = (int)(m & 255u);
        const uint32_t mn = m >> 8;
        const int j = next_set64<NPL>(alive, mi + 1);
        const int j2 = next_set64<NPL>(alive, j + 1);
        const int j3 = j2 >= 0 ? next_set64<NPL>(alive, j2 + 1) : -1;
        const int h = prev_set64<NPL>(alive, mi);
        const int e_r = j3 >= 0 ? j3 : n;                   // end of the pair (mi, j2)
        const int e_mi = j2 >= 0 ? j2 : n;                  // end of the merged node
        const int len_r = e_r - mi, len_h = e_mi - h;
        uint32_t id_j2 = 0;
        if (j2 >= 0 && len_r > SUB_LMAX) {
            uint32_t sel = id[0];
#pragma unroll
            for (int k = 1; k < NPL; k++) sel = (j2 >> 6) == k ? id[k] : sel;
            id_j2 = __builtin_amdgcn_readlane(sel, j2 & 63);
        }
#pragma unroll
        for (int k = 0; k < NPL; k++) {
            const int i = lane + 64 * k;
            if (i == mi) {
                id[k] = mn;
                rk[k] = j2 < 0 ? SPL_NO_RANK : len_r <= SUB_LMAX ? sub[i * SUB_W + len_r - 2] : pair_rank(T, mn, id_j2);
            } else if (i == h) {
                rk[k] = len_h <= SUB_LMAX ? sub[i * SUB_W + len_h - 2] : pair_rank(T, id[k], mn);
            } else if (i == j) {
                rk[k] = SPL_NO_RANK;
            }
        }
#pragma unroll
        for (int k = 0; k < NPL; k++)
            if ((j >> 6) == k) alive[k] &= ~(1ull << (j & 63));
    }
#pragma unroll
    for (int k = 0; k < NPL; k++) {
        const int i = lane + 64 * k;
        if (i < n && ((alive[k] >> lane) & 1ull) && id[k] != SPL_NO_RANK) emit(i, id[k]);
    }
}

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

// Workgroup -> tile.  Workgroups go to the eight XCDs round robin; with SPL_XCD_MAP each XCD works a CONTIGUOUS eighth
// of the tiles (a bijection for any grid size), so that neighbouring tiles share their halo lines -- and k_tile_out
// finds a tile's ids -- in that XCD's own L2.
#ifndef SPL_XCD_MAP
#define SPL_XCD_MAP 1            /* 0: workgroup i works tile i (A/B) */
#endif
__device__ __forceinline__ uint32_t xcd_tile() {
    if (!SPL_XCD_MAP) return blockIdx.x;
    const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3, q = gridDim.x >> 3, r = gridDim.x & 7u;
    return x * q + (x < r ? x : r) + j;
}
// e_flags of k_pretok: which optional inputs exist, and the split pattern
constexpr uint32_t PRETOK_E_TSTART = 1u, PRETOK_E_SKIP = 2u, PRETOK_E_GAPS = 4u, PRETOK_E_EXT = 8u;
inline uint32_t pretok_flags(const DeviceTables& T, const Batch& b) {
    return (b.tstart ? PRETOK_E_TSTART : 0u) | (b.skip ? PRETOK_E_SKIP : 0u) | (b.ext_gaps ? PRETOK_E_GAPS : 0u) |
           (b.ext_starts ? PRETOK_E_EXT : 0u) | (T.pattern << 4);
}
// the kernel-argument segment of k_pretok as the ABI lays it out (every argument at its natural alignment, in order)
struct PretokKernargs {
    const uint8_t* e_text; const uint64_t* e_doc_off; uint32_t e_n_bytes, e_n_docs; unsigned long long* e_dbg;
    const uint32_t* e_akind; uint32_t e_flags; DeviceTables T; Batch b;
};
#define PRETOK_EARLY(T, b) (b).text, (b).doc_off, (b).n_bytes, (b).n_docs, (b).dbg, (T).akind, pretok_flags(T, b)
template <int TB_, int RH_>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(SPL_PRETOK_WAVES)))
void k_pretok(const uint8_t* e_text, const uint64_t* e_doc_off, uint32_t e_n_bytes, uint32_t e_n_docs, unsigned long long* e_dbg,
              const uint32_t* e_akind, uint32_t e_flags, DeviceTables T_ka, Batch b_ka) {
    // The e_* arguments repeat what the first phase needs (text, offsets, sizes, which optional bitmaps exist, the
    // pattern, the ASCII kind table) as LEADING SCALARS -- the first line of the argument segment -- so that the text
    // and offset loads go out before the two structs are touched: 0.5 KB that five thousand wavefronts ask the same
    // few L2 lines for at the same moment (profiles/r03_launch_probes.txt).  The structs themselves are read through
    // the kernel-argument segment pointer, laundered BEHIND the first text loads (T and b below): left to itself the
    // compiler hoists all their loads to the kernel's first instructions, waits for them there and parks the values
    // in VGPR lanes (153 spilled SGPRs, 74 this way).  Built with -mllvm -amdgpu-kernarg-preload-count=16 the
    // scalars would arrive in SGPRs with the wavefront; measured, that is no faster (the wave launch waits instead).
    using G = TileGeom<TB_, RH_>;
    constexpr bool DIRECT = true;                        // (every launch leaves tile records since round 4: tile-owned and queue mode; the
                                                         //  instantiations without them belonged to the multi-pass pipeline)
#ifdef SPL_FIXED_PATTERN
    constexpr int KPAT = SPL_FIXED_PATTERN;              // (A/B: the kernel specialised for one split pattern)
#else
    const int KPAT = (int)((e_flags >> 4) & 3u);
#endif
    constexpr int Wv = G::Wv;
    __shared__ __attribute__((aligned(16))) uint32_t s_txt32[G::NW32];
    __shared__ __attribute__((aligned(16))) union {
        PretokScanLds<TB_, RH_> a;
        PretokTailLds t;
    } s_u;
    uint32_t* const s_rec32 = s_u.a.rec32;
    uint32_t* const s_mk = s_u.a.mk;
    auto& s_sub = s_u.a.sub;
    uint32_t* const s_miss = s_u.a.miss;
    __shared__ uint32_t s_ts[G::NBW + 1];                // text-start bits of the window
    __shared__ uint32_t s_sk[G::NBW + 1];                // special-literal bits of the window
    __shared__ uint32_t s_cbits[G::NBW + 1];
    __shared__ uint32_t s_kill[G::NBW + 1], s_add[G::NBW + 1];   // o200k contraction suffixes: starts to drop / to add
    __shared__ uint32_t s_tbits[G::NBW + 1];
    static_assert(!DIRECT || (Wv + 2) / 2 >= SG_WORDS, "bpe_tail_segments' scratch must fit s_cpos");
    __shared__ __attribute__((aligned(16))) uint16_t s_cpos[Wv + 2];   // (the single-pass tail borrows it: bpe_tail_segments)
    __shared__ uint8_t s_ascii[128];
    __shared__ __attribute__((aligned(8))) KindEnt s_aent[128];   // ASCII byte -> kind nibbles | class (spl_scan_words.h)
    __shared__ __attribute__((aligned(8))) KindEnt s_kent[16];    // class -> kind nibbles
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_total;
    __shared__ uint32_t s_nch;                           // chunks on the probe list (small windows)
    __shared__ uint32_t s_fast;                          // the tile's starts came from the bit-vector computation
    __shared__ uint32_t s_scnt[17];                      // counting sort of the short misses by length
    __shared__ uint32_t s_nq[4];                         // miss counts [0] (<= 16 B) [1] (17..64 B), work cursors [2] [3]
    // single-pass state
    __shared__ uint32_t s_ids[DIRECT ? Wv : 1];          // id of the token that starts at this window index
    __shared__ uint32_t s_wpre[DIRECT ? G::NBW + 2 : 1]; // exclusive token counts of the window's bitmap words
    __shared__ uint32_t s_lq[DIRECT ? 2 * DIRECT_LQCAP : 1];   // (global position, length): the tail's working list
    constexpr bool TILE_LIST = DIRECT && SPL_TILE_MISS_LIST;
    __shared__ uint32_t s_tmiss[TILE_LIST ? G::C16 : 1]; // tile-owned: EVERY miss of the tile, p | n << 16 (outside the union:
                                                         // the tail's slab overlays the scanner's arrays)
    __shared__ uint32_t s_dq[12];                        // [0] long-list fill [1] deferred count [2],[3] deferred starts
                                                         // [4] end of the overflow range [5] chain cursor [6] chain done
    __shared__ unsigned long long s_red[NT / 64];
    uint8_t* const s_txt = reinterpret_cast<uint8_t*>(s_txt32);
    uint8_t* const s_rec = reinterpret_cast<uint8_t*>(s_rec32);
    // probe list of a small window: p | n << 16 per chunk, in the (then still unused) substring table
    constexpr bool LIST_CHUNKS = Wv <= (NT / 16) * 16 * SUB_W;
    uint32_t* const s_chunk = &s_sub[0][0];
    // Phase stamps, per-workgroup records and the phase cut-off are compiled in only with
    // -DSPL_DEBUG_STAMPS (tools/ab_build.sh): their live values cost the product kernel registers.
#if defined(SPL_DEBUG_STAMPS) && defined(SPL_STAMP_ALL)
    // every workgroup's wall clock at the phase boundaries (tools/dev/gpu_phase_walls.py): eight words per workgroup
    // in the per-workgroup record area -- start, stamps 1 2 3 4 6 7, end
#define SPL_STAMP(i) do { if (e_dbg && threadIdx.x == 0 && SPL_REC_BLK < SPL_DEBUG_BLOCKS / 2 && (i) >= 1 && (i) <= 7 && (i) != 5) \
                              e_dbg[16 + 8 * SPL_REC_BLK + ((i) < 5 ? (i) : (i) - 1)] = (unsigned long long)wall_clock64(); } while (0)
#elif defined(SPL_DEBUG_STAMPS)
#define SPL_STAMP(i) do { if (e_dbg && blockIdx.x == SPL_DBG_WG && threadIdx.x == 0) e_dbg[i] = clock64(); \
                          if ((i) >= 1 && (i) <= 7 && b.stop_phase == (uint32_t)(i)) return; } while (0)
#else
#define SPL_STAMP(i) do { } while (0)
#endif

#ifndef SPL_ROTATE_WAVES
#define SPL_ROTATE_WAVES 0
#endif
    // (A/B) logical wavefront index rotated by the workgroup index: phases that only fill the low
    // wavefronts (chains, probe list, per-word scans) then load different SIMDs in different workgroups
    const int tid = SPL_ROTATE_WAVES ? (int)((threadIdx.x + ((blockIdx.x & 3u) << 6)) & (NT - 1)) : (int)threadIdx.x;
#ifdef SPL_PASSES      /* timing experiment only (group sums wrong): every workgroup works its tile SPL_PASSES times -- the later
                          passes find the kernel's code in the instruction cache (tools/dev/gpu_phase_walls.py) */
    for (int spl_pass = 0; spl_pass < SPL_PASSES; spl_pass++) {
    __syncthreads();
#define SPL_REC_BLK (blockIdx.x + 1024u * (uint32_t)spl_pass)
#else
#define SPL_REC_BLK blockIdx.x
#endif
    if (DIRECT) __builtin_amdgcn_s_setprio(SPL_WORK_PRIO);
    const uint32_t tile_ix = xcd_tile();
    const int64_t t0 = (int64_t)tile_ix * TB_;
    const int64_t w0 = t0 - LH;                       // global position of window index 0
    const int64_t B = e_n_bytes;
    // profiling: span of this kernel on the constant-rate wall clock (start of workgroup 0, max end
    // over all workgroups) -- what a kernel trace reports, without host-side event overhead
    if (e_dbg && tid == 0 && blockIdx.x == 0) e_dbg[14] = (unsigned long long)wall_clock64();   // dispatched first
#ifdef SPL_DEBUG_STAMPS
    const unsigned long long blk_t0 = e_dbg ? (unsigned long long)wall_clock64() : 0ull;
    unsigned long long blk_w1 = 0, blk_w2 = 0;
#endif

    // ---- stage text (coalesced 16 B per lane): the loads go out before anything else ------------
    auto text16 = [&](int v) {
        const int64_t g = w0 + (int64_t)v * 16;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (g >= 0 && g + 16 <= B) x = *reinterpret_cast<const uint4*>(e_text + g);
        else if (g >= 0 && g < B) {
            uint32_t tmp[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++)
                if (g + k < B) tmp[k >> 2] |= (uint32_t)e_text[g + k] << (8 * (k & 3));
            x = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
        return x;
    };
    constexpr bool ONE_ROUND = (Wv + WPAD) / 16 <= NT;           // small windows: at most one 16-byte load per lane
    uint4 x_first = make_uint4(0, 0, 0, 0);
    if (ONE_ROUND && tid < (Wv + WPAD) / 16) x_first = text16(tid);
    // the two argument structs, from here on (see the head of the kernel)
#ifndef SPL_LATE_KERNARGS
#define SPL_LATE_KERNARGS 1
#endif
    typedef const PretokKernargs __attribute__((address_space(4))) KernargsK;
    KernargsK* ka = (KernargsK*)__builtin_amdgcn_kernarg_segment_ptr();
    if (SPL_LATE_KERNARGS) asm volatile("" : "+s"(ka) : : "memory");
    const DeviceTables& T = SPL_LATE_KERNARGS ? *(const DeviceTables*)&ka->T : T_ka;
    const Batch& b = SPL_LATE_KERNARGS ? *(const Batch*)&ka->b : b_ka;
#ifndef SPL_KERNARG_PREFETCH
#define SPL_KERNARG_PREFETCH 0   /* (A/B) 1: one lane per 64-byte line of the argument segment touches it with a vector load right behind the text loads */
#endif
    uint32_t ka_pf = 0;
    if (SPL_KERNARG_PREFETCH && tid < (int)((sizeof(PretokKernargs) + 63) / 64))
        ka_pf = *reinterpret_cast<const volatile uint32_t*>((uintptr_t)ka + 64u * (uint32_t)tid);
#ifdef SPL_DEBUG_STAMPS
    if (e_dbg && tid == 0 && blockIdx.x == SPL_DBG_WG) e_dbg[11] = (unsigned long long)wall_clock64();
    if (e_dbg && tid == 0 && blockIdx.x == gridDim.x - 1) e_dbg[13] = (unsigned long long)wall_clock64();
#endif
    if (ONE_ROUND) {
        if (tid < (Wv + WPAD) / 16) *reinterpret_cast<uint4*>(s_txt32 + tid * 4) = x_first;
    } else {
        for (int v = tid; v < (Wv + WPAD) / 16; v += NT) *reinterpret_cast<uint4*>(s_txt32 + v * 4) = text16(v);
    }
    if (tid < G::NBW + 1) {
        const int64_t wi = (w0 >> 5) + tid;           // w0 is a multiple of 32
        const bool in = wi >= 0 && wi * 32 < B;
        // (tile-owned mode has these bitmaps only for SPL_WITH_SPECIAL: document starts come from
        //  the search below, the bitmap adds the text starts behind special literals)
        s_ts[tid] = (in && (!DIRECT || (e_flags & PRETOK_E_TSTART))) ? b.tstart[wi] : 0u;
        s_sk[tid] = ((in && (e_flags & PRETOK_E_SKIP)) ? b.skip[wi] : 0u) | ((DIRECT && in && (e_flags & PRETOK_E_GAPS)) ? b.ext_gaps[wi] : 0u);
        s_cbits[tid] = 0;
        s_kill[tid] = 0; s_add[tid] = 0;
        s_tbits[tid] = 0;
    }
    if (tid < 128) {                                      // the ASCII kind table (spl_scan_words.h), built once per handle on the host: 1 KB
        const uint2 e = reinterpret_cast<const uint2*>(e_akind)[tid];
        s_aent[tid] = KindEnt{e.x, e.y};
        s_ascii[tid] = (uint8_t)(e.y >> 28);              // (the byte's class code rides in the top nibble)
    } else if (tid < 144) s_kent[tid - 128] = kind_entry((uint32_t)tid - 128u);
    if (tid < 4) s_nq[tid] = 0;
    if (tid < 17) s_scnt[tid] = 0;                    // (the counting sort of the merge phase: zeroed here, one barrier less there)
    if (tid < 12) s_dq[tid] = 0;
    if (DIRECT) {                                            // (length 0: no entry)
        int t_early = tid;                                   // an index of its own: shared with the tail's uses of
        asm volatile("" : "+v"(t_early));                    // s_lq[2 * tid], it would be kept -- spilled -- until then
        if (t_early < DIRECT_LQCAP) s_lq[2 * t_early + 1] = 0;
    }
    if (tid == 0) { s_nch = 0; s_fast = 0; }
    // single pass: the window's text starts straight from doc_off.  NT-ary search for the first
    // document that starts at or after the window (two rounds up to 65 536 documents), then the
    // documents of the window set their bits.
    uint32_t dw = 0;                                   // first document with doc_off >= max(w0, 0)
    if (DIRECT) {
        uint32_t lo = 0, hi = e_n_docs;
        const uint64_t target = w0 > 0 ? (uint64_t)w0 : 0ull;
        uint64_t p_held = ~0ull;                        // doc_off[d_held] from the first round, if it settled the search
        uint32_t d_held = 0xFFFFFFFFu, d_held_end = 0;
        if (target != 0 && hi > (uint32_t)NT) {
            // first round by interpolation: with documents of similar size the answer lies within NT
            // entries of target * n_docs / n_bytes, and ONE round of loads finds it; otherwise this
            // round only narrows [lo, hi] for the search below.  (The guess in float: it only has to be
            // near, and a 64-bit division costs a wavefront more than a hundred instructions.)
            const float gf = (float)target * ((float)hi * __builtin_amdgcn_rcpf((float)B));
            const uint32_t g = gf >= (float)hi ? hi : (uint32_t)gf;
            const uint32_t glo = g > (uint32_t)(NT / 2) ? g - NT / 2 : 0u;
            const uint32_t ghi = glo + NT < hi ? glo + NT : hi;
            const uint32_t idx = glo + (uint32_t)tid;
            const uint64_t p1 = idx < ghi ? e_doc_off[idx] : ~0ull;
            const bool below = idx < ghi && p1 < target;
            const uint32_t c = (uint32_t)__syncthreads_count(below);
            if (c == 0) hi = glo;                           // entry glo (if any) is not below the target
            else if (c == ghi - glo) lo = ghi;              // every probed entry is
            else { lo = hi = glo + c; p_held = p1; d_held = idx; d_held_end = ghi; }   // found: the entries behind it are already here
        }
        while (target != 0 && lo < hi) {
            const uint32_t span = hi - lo, st = (span + NT - 1) / NT;
            const uint64_t idx = (uint64_t)lo + (uint64_t)tid * st;
            const bool below = idx < hi && e_doc_off[idx] < target;
            const uint32_t c = (uint32_t)__syncthreads_count(below);
            if (c == 0) { hi = lo; break; }
            const uint64_t nhi = (uint64_t)lo + (uint64_t)c * st;
            lo = lo + (c - 1) * st + 1;                // element lo + (c-1)*st is below the target
            hi = nhi < hi ? (uint32_t)nhi : hi;        // element lo + c*st (if any) is not
        }
        dw = lo;
        __syncthreads();                               // s_ts zeroed by all before any bit is set
        const uint64_t lim = (uint64_t)(w0 + (int64_t)(G::NBW + 1) * 32);
        uint32_t base = dw;
        if (d_held_end > dw) {                         // the window's documents from the first round's loads
            const bool in = d_held >= dw && d_held < d_held_end && p_held < lim && p_held < (uint64_t)B;
            if (in) { const uint32_t i = (uint32_t)(p_held - (uint64_t)w0); atomicOr(&s_ts[i >> 5], 1u << (i & 31)); }
            // more only if the last entry fetched is still inside the window
            base = __syncthreads_or(d_held == d_held_end - 1u && in) ? d_held_end : 0xFFFFFFFFu;
        }
        for (; base != 0xFFFFFFFFu; base += NT) {
            const uint64_t d = (uint64_t)base + tid;
            uint64_t p = ~0ull;
            if (d < e_n_docs) p = e_doc_off[d];
            const bool in = p < lim && p < (uint64_t)B;
            if (in) { const uint32_t i = (uint32_t)(p - (uint64_t)w0); atomicOr(&s_ts[i >> 5], 1u << (i & 31)); }
            if (!__syncthreads_or(tid == NT - 1 && in)) break;
        }
    }
    SPL_STAMP(0);
    __syncthreads();
    if (SPL_KERNARG_PREFETCH) asm volatile("" : : "v"(ka_pf));
    SPL_STAMP(1);

    const int iB = (B - w0 < (int64_t)Wv) ? (int)(B - w0) : Wv;   // first index past the text
    const int iT = (B - w0 < (int64_t)(Wv + WPAD)) ? (int)(B - w0) : Wv + WPAD;   // staged text end
    constexpr int NBW1 = G::NBW + 1;
    const bool ext = DIRECT && (e_flags & PRETOK_E_EXT) != 0u;     // chunk boundaries come from the host splitter
    if (ext) {
        // The tile owns the chunks that START in its own range [LH, LH + TB): their starts (and the terminator of
        // the last one: the first start at or behind the tile's end, a document start, or the end of the corpus)
        // are the window's bits of the external bitmap -- the "fast starts" path takes them from s_cbits as it
        // takes the bit-vector starts.  A last chunk whose end lies beyond the window is finished by the tail
        // from global memory (one deferred start, as a chain that outgrows the window).
        if (tid < 64) {
            const int ln = tid;
            const bool in = ln < G::NBW;
            uint32_t ew = 0;
            if (in) {
                const int64_t wi = (w0 >> 5) + ln;
                if (wi >= 0 && wi * 32 < B) ew = b.ext_starts[wi];
                ew |= s_ts[ln];
                if (B - w0 <= (int64_t)Wv && (iB >> 5) == ln) ew |= 1u << (iB & 31);      // the corpus ends inside the window
                if ((iB >> 5) == ln && (iB & 31) != 31) ew &= (2u << (iB & 31)) - 1u;      // nothing behind its end
                if ((iB >> 5) < ln) ew = 0;
            }
            auto range_word = [&](int from, int to) -> uint32_t {             // bits [from, to) of this lane's word
                const int lo = from - ln * 32, hi = to - ln * 32;
                if (hi <= 0 || lo >= 32) return 0u;
                uint32_t w = ~0u;
                if (lo > 0) w &= ~0u << lo;
                if (hi < 32) w &= (1u << hi) - 1u;
                return w;
            };
            auto first_in = [&](int from, int to) -> int {
                const uint32_t word = ew & range_word(from, to);
                const unsigned long long bl = __ballot(word != 0u);
                if (!bl) return -1;
                const int l0 = __ffsll((long long)bl) - 1;
                return l0 * 32 + __ffs((int)__builtin_amdgcn_readlane(word, l0)) - 1;
            };
            auto last_in = [&](int from, int to) -> int {
                const uint32_t word = ew & range_word(from, to);
                const unsigned long long bl = __ballot(word != 0u);
                if (!bl) return -1;
                const int l0 = 63 - __builtin_clzll(bl);
                return l0 * 32 + 31 - __clz((int)__builtin_amdgcn_readlane(word, l0));
            };
            const int fs = first_in(LH, LH + TB_);
            if (fs >= 0) {
                const int ls = last_in(LH, LH + TB_);
                const int fe = first_in(LH + TB_, Wv + 1);
                const uint32_t bits = ew & range_word(fs, (fe >= 0 ? fe : ls) + 1);
                if (in && bits) s_cbits[ln] = bits;
                if (fe < 0 && ln == 0) { s_dq[1] = 1u; s_dq[2] = (uint32_t)(w0 + ls); }   // the chunk at ls outgrows the window
            }
            if (ln == 0) s_fast = 3u;
        }
        SPL_STAMP(2);
        __syncthreads();
    } else {
    // ---- classify + class bitmasks in ONE pass, four bytes per lane (spl_scan_words.h) -------------------
    // Each lane turns its word into the four class records and into two words of kind NIBBLES (bit k of nibble j: byte k is
    // of kind j); ASCII words -- nearly all of English / code -- through a 128-entry LDS table.  Eight neighbouring lanes
    // then transpose their nibbles (three DPP exchanges per word) and lane 8w + j holds mask word w of kind j.  Up to round
    // 3 this was two steps with a barrier between them -- records first, then one byte per lane and a dozen ballots per
    // 64-byte row -- and a fifth of the kernel's vector instructions (profiles/r04_phase_instruction_mix.txt).
    // (records past the window are all "window end": written directly)
    for (int wi = Wv / 4 + tid; wi < G::NW32; wi += NT) s_rec32[wi] = (uint32_t)C_WEND * 0x01010101u;
    if (tid < MK_COUNT) s_mk[tid * NBW1 + G::NBW - 1] = 0;   // the word of position W (never a real byte)
    if (tid < MK_COUNT) s_mk[tid * NBW1 + G::NBW] = 0;
    for (int wbase = 0; wbase < Wv / 4; wbase += NT) {       // (uniform trip count: every lane takes part in the exchanges)
        const int wi = wbase + tid;
        WordKinds wk{0u, 0u, 0u};
        if (wi < Wv / 4) {
            const int i0 = wi * 4;
            const uint32_t tw = s_txt32[wi];
            const uint32_t ts4 = (s_ts[i0 >> 5] >> (i0 & 31)) & 0xFu;
            const uint32_t sk4 = (s_sk[i0 >> 5] >> (i0 & 31)) & 0xFu;
            if (!(tw & 0x80808080u) && sk4 == 0u && i0 + 3 < iB && w0 + i0 >= 0) {
                const KindEnt e[4] = {s_aent[tw & 0xFFu], s_aent[(tw >> 8) & 0xFFu], s_aent[(tw >> 16) & 0xFFu], s_aent[tw >> 24]};
                wk = classify_word_ascii(e, ts4);
            } else {
                // A word with a byte beyond ASCII (or at an edge of the text): the neighbouring words and the text-start bits
                // of [i0 - 4, i0 + 12) go into registers once; the look-back / look-ahead (at most 3 bytes either way, plus
                // the decode) is arithmetic on them.  (Bytes before window index 0 do not exist for the look-back: that
                // only concerns the first bytes of the left halo, whose records nothing in the tile depends on.)
                uint32_t ts16;
                const int b0 = i0 - 4;                        // (a multiple of 4; negative only for the first word)
                if (b0 < 0) ts16 = s_ts[0] << 4;
                else {
                    const int sh = b0 & 31;
                    ts16 = s_ts[b0 >> 5] >> sh;
                    if (sh > 16) ts16 |= s_ts[(b0 >> 5) + 1] << (32 - sh);
                }
                const uint32_t wp = wi > 0 ? s_txt32[wi - 1] : 0u, wn = s_txt32[wi + 1];
                const int lo_i = w0 < 0 ? (int)-w0 : 0;
                // well-formed text away from every edge (an accented letter, a dash, CJK): the lean form; else the general one
                bool done = false;
                if (sk4 == 0u && i0 + 3 < iB && i0 >= lo_i)
                    done = classify_word_text(T, KPAT, wp, tw, wn, ts16, [&](uint32_t c) { return s_aent[c]; },
                                              [&](uint32_t c) { return s_kent[c]; }, i0, lo_i, iT, wk);
                if (!done)
                    wk = classify_word(T, KPAT, wp, tw, wn, ts16, [&](uint32_t c) { return s_kent[c]; },
                                       [&](uint32_t c) { return (uint32_t)s_ascii[c]; }, ts4, sk4, i0, iB, Wv, lo_i, iT);
            }
            s_rec32[wi] = wk.rec;
        }
        const uint32_t t0k = nib_transpose8(wk.v0), t1k = nib_transpose8(wk.v1);
        const uint32_t g8 = (uint32_t)tid & 7u;
        if (wi < Wv / 4) {
            s_mk[((V0_KINDS >> (4u * g8)) & 15u) * NBW1 + (wi >> 3)] = t0k;
            if (g8 < (uint32_t)V1_NKINDS) s_mk[((V1_KINDS >> (4u * g8)) & 15u) * NBW1 + (wi >> 3)] = t1k;
        }
    }
    __syncthreads();
    SPL_STAMP(2);
    // (the sync-point mask -- word operations on the kind masks, the rules of is_sync -- is made by the wavefronts that compute the starts)
    // ---- ALL match starts of the tile by bit-vector arithmetic (spl_scan_starts.h), every pattern ------
    // One mask word per lane.  The tile owns [fs, fe): fs = its first sync point, fe = the
    // first sync point or text start at or behind the tile's end.  Needs fe inside the window and no
    // disqualifying byte (MK_BAD) in the range; otherwise the chains below do the work as before.
    // Three wavefronts share the work (letters and numbers / "other" runs and contractions / whitespace);
    // each finds the range for itself and ORs its starts into s_cbits; the tile is "fast" if all three agree.
    static_assert(LIST_CHUNKS, "the probe list lives in the substring table; the chains must not share s_cbits with the start masks");
    if (SPL_MASK_STARTS && DIRECT && tid < 192) {
        const int part = tid >> 6, ln = tid & 63;           // lane ln owns mask word ln
        uint32_t fine = 0;
        {
            const bool in = ln < G::NBW;
            auto ld = [&](int k) { return in ? s_mk[k * NBW1 + ln] : 0u; };
            // the sync-point mask of this lane's word, from the kind words and their left neighbours' top bits
            const uint32_t ts = ld(MK_TS);
            uint32_t sy;
            {
                uint32_t kw[MK_COUNT], kp[MK_COUNT];
#pragma unroll
                for (int k = 0; k < MK_COUNT; k++) {
                    const bool used = k == MK_L || k == MK_N || k == MK_S || k == MK_NL || k == MK_O || k == MK_CS || k == MK_TS ||
                                      (KPAT != PAT_CL100K && (k == MK_M || k == MK_AP));
                    const bool shifted = k == MK_L || k == MK_N || k == MK_NL || k == MK_O || (KPAT != PAT_CL100K && k == MK_M);
                    kw[k] = used ? ld(k) : 0u;
                    kp[k] = shifted ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kw[k], 0x138, 0xF, 0xF, true) : 0u;   // wave_shr:1: lane - 1's word
                }
                sy = sync_word(KPAT, kw, kp);
                if (part == 0 && in) s_mk[MK_SY * NBW1 + ln] = sy;      // (the chains of a tile that does not qualify read it)
            }
            auto range_word = [&](int from, int to) -> uint32_t {             // bits [from, to) of this lane's word
                const int lo = from - ln * 32, hi = to - ln * 32;
                if (hi <= 0 || lo >= 32) return 0u;
                uint32_t w = ~0u;
                if (lo > 0) w &= ~0u << lo;
                if (hi < 32) w &= (1u << hi) - 1u;
                return w;
            };
            auto first_in = [&](uint32_t word, int from, int to) -> int {    // first set bit in [from, to), -1 if none
                word &= range_word(from, to);
                const unsigned long long bl = __ballot(word != 0u);
                if (!bl) return -1;
                const int l0 = __ffsll((long long)bl) - 1;
                return l0 * 32 + __ffs((int)__builtin_amdgcn_readlane(word, l0)) - 1;
            };
            const int fs = first_in(sy, LH, LH + TB_);
            const int fe = first_in(sy | ts, iB < LH + TB_ ? iB : LH + TB_, Wv + 1);   // (a text that ends in the tile: its end)
            if (fs < 0) fine = 1;                              // nothing owned
            else if (fe >= 0) {
                const uint32_t own = range_word(fs, fe);
                if (__any((ld(MK_BAD) & own) != 0u)) {
                } else if (KPAT != PAT_CL100K) {
                    // o200k family: letters + numbers / "other" runs and contraction suffixes / whitespace
                    const bool mistral = KPAT == PAT_MISTRAL_V3;
                    const O200kStartMasks<WaveBV> om{WaveBV{ld(MK_L)}, WaveBV{ld(MK_UP)}, WaveBV{ld(MK_LB)}, WaveBV{ld(MK_N)}, WaveBV{ld(MK_S)},
                                                     WaveBV{ld(MK_NL)}, WaveBV{ld(MK_O)}, WaveBV{ld(MK_AP)}, WaveBV{ld(MK_SP)},
                                                     WaveBV{ld(MK_SL)}, WaveBV{ld(MK_CS)}, WaveBV{ts}};
                    bool ok = true;
                    uint32_t bits;
                    if (part == 0) bits = o200k_starts_ln(om, mistral, ok, 16).x | range_word(fe, fe + 1);          // + the terminator
                    else if (part == 2) bits = o200k_starts_s(om, mistral, ok, 16).x;
                    else {
                        WaveBV CAND;
                        bits = o200k_starts_o(om, mistral, CAND, ok, 16).x;
                        uint32_t ca = mistral ? 0u : CAND.x & own;
                        const LdsAcc acc{s_rec, s_txt};
                        bool chain = false;
                        while (ca) {                           // the few apostrophes behind a letter
                            const int ap = ln * 32 + __ffs((int)ca) - 1;
                            ca &= ca - 1;
                            int e;
                            if (!o200k_contraction_at(acc, ap, e)) chain = true;
                            else if (e > 0) {                  // the suffix starts nothing, the byte behind it does
                                for (int q = ap; q < e; q++) atomicOr(&s_kill[q >> 5], 1u << (q & 31));
                                if (e < fe) atomicOr(&s_add[e >> 5], 1u << (e & 31));
                            }
                        }
                        if (__any(chain)) ok = false;
                    }
                    bits &= range_word(fs, fe + 1);
                    if (ok) {
                        fine = 1;
                        if (bits) atomicOr(&s_cbits[ln], bits);
                    }
                } else {
                    const Cl100kStartMasks<WaveBV> cm{WaveBV{ld(MK_L)}, WaveBV{ld(MK_N)}, WaveBV{ld(MK_S)}, WaveBV{ld(MK_NL)},
                                                      WaveBV{ld(MK_O)}, WaveBV{ld(MK_AP)}, WaveBV{ld(MK_SP)}, WaveBV{ld(MK_CS)}, WaveBV{ts}};
                    bool ok = true;
                    uint32_t bits;
                    if (part == 0) bits = cl100k_starts_ln(cm, ok, 16).x | ts | range_word(fe, fe + 1);   // + text starts, terminator
                    else if (part == 2) bits = cl100k_starts_s(cm, ok, 16).x;
                    else {
                        WaveBV CA;
                        bits = cl100k_starts_o(cm, CA).x;
                        uint32_t ca = CA.x & own;
                        const LdsAcc acc{s_rec, s_txt};
                        while (ca) {                           // the few apostrophes that start a match
                            const int ap = ln * 32 + __ffs((int)ca) - 1;
                            ca &= ca - 1;
                            const int e = contraction(acc, ap);
                            if (e > 0 && e < fe) atomicOr(&s_cbits[e >> 5], 1u << (e & 31));
                        }
                    }
                    bits &= range_word(fs, fe + 1);
                    if (ok) {
                        fine = 1;
                        if (bits) atomicOr(&s_cbits[ln], bits);
                    }
                }
            }
        }
        if (ln == 0 && fine) atomicAdd(&s_fast, 1u);
    }
    }   // !ext
    if (SPL_MASK_STARTS && DIRECT) __syncthreads();
    const bool fast_starts = SPL_MASK_STARTS && DIRECT && s_fast == 3u;
    SPL_STAMP(3);

    // ---- chains: each sync point inside the tile scans to the next sync point -------------------
    // The sync points are first enumerated (popcount scan of the sync mask restricted to the tile)
    // so that every lane runs ONE chain: lanes that own a word with several sync points would
    // otherwise serialise them while their neighbours idle.
    {
        uint32_t word = 0;
        if (fast_starts) {
            if (tid < G::NBW) word = (s_cbits[tid] & ~s_kill[tid]) | s_add[tid];   // the tile's starts and their terminator
        } else if (tid < G::NBW) {
            word = s_mk[MK_SY * NBW1 + tid];
            const int lo = LH - tid * 32, hi = LH + TB_ - tid * 32;       // tile range inside this word
            if (hi <= 0 || lo >= 32) word = 0;
            else {
                if (lo > 0) word &= ~0u << lo;
                if (hi < 32) word &= (1u << hi) - 1u;
            }
        }
        const uint32_t cnt = __popc(word);
        uint32_t x = wave_scan_incl(cnt);
        if ((tid & 63) == 63) s_wsum[tid >> 6] = x;
        __syncthreads();
        uint32_t base = x - cnt;
        for (int wv = 0; wv < (tid >> 6); wv++) base += s_wsum[wv];
        if (tid == NT - 1) s_total = base + cnt;
        while (word) {
            const int bit = __ffs(word) - 1;
            word &= word - 1;
            s_cpos[base++] = (uint16_t)(tid * 32 + bit);
        }
        __syncthreads();
    }
    {
        const MaskLdsAcc acc{s_rec, s_txt, s_mk, NBW1, Wv, (B - w0) <= (int64_t)Wv};
        const int nsync = fast_starts ? 0 : (int)s_total;
        // only the LAST chain of a tile can reach the window end, so at most one start is recorded
        auto push_defer = [&](uint32_t gpos) {
            if (DIRECT && !b.qcount) {
                const uint32_t qi = atomicAdd(&s_dq[1], 1u);
                if (qi < 2) s_dq[2 + qi] = gpos;
            } else {
                const uint32_t qi = atomicAdd(&b.qcount[3], 1u);
                if (qi < b.qcapdefer) b.qdefer[qi] = gpos;
            }
        };
        for (int k = tid; k < nsync; k += NT) {
            int p = s_cpos[k];
            for (;;) {
                const int e = match_end_m(acc, p, KPAT);
                if (e == SPL_DEFER) {                 // the match outgrows the window
                    push_defer((uint32_t)(w0 + p));
                    break;
                }
                // small windows: the chunk goes straight onto the probe list (order is irrelevant:
                // tokens are identified by their position) -- no marks, no second enumeration
                s_chunk[atomicAdd(&s_nch, 1u)] = (uint32_t)p | ((uint32_t)(e - p) << 16);
                p = e;
                if (p >= Wv) {                         // ended on the window edge, or up to WPAD bytes behind it (a straddling character)
                    // The chain goes on from p -- IF a chunk starts there: p may be a sync point, which the tile that
                    // holds it works itself (bit 31: "check first").  (It used to go on from the window's end
                    // whatever p was, as a certain chunk start: a chunk that ended behind the edge was then partly
                    // worked twice, and a sync point exactly on the edge got its chunk from both tiles.)
                    if (w0 + p < B) push_defer((uint32_t)(w0 + p) | 0x80000000u);
                    break;
                }
                if (((s_mk[MK_SY * NBW1 + (p >> 5)] | s_mk[MK_TS * NBW1 + (p >> 5)]) >> (p & 31)) & 1u) break;   // the next owner's start
            }
        }
    }
    __syncthreads();
    SPL_STAMP(4);

    SPL_STAMP(5);

    // ---- whole-chunk probe (start masks: the last marked position is only a terminator) ----------------
    {
        LdsAcc tx{s_rec, s_txt};
        const bool from_list = LIST_CHUNKS && !fast_starts;
        const int K = from_list ? (int)s_nch + 1 : (int)s_total;
        for (int k = tid; k + 1 < K; k += NT) {
            int p, n;
            if (from_list) {
                const uint32_t c = s_chunk[k];
                p = (int)(c & 0xFFFFu); n = (int)(c >> 16);
            } else {
                p = s_cpos[k];
                if (ext ? ((s_sk[p >> 5] >> (p & 31)) & 1u) != 0u : (s_rec[p] & CB_CLASS) >= C_EOT) continue;   // a special-literal span / dropped bytes
                n = (int)s_cpos[k + 1] - p;
            }
            const uint32_t id = probe_chunk_tile(T, tx, p, n);
            if (id != SPL_NO_RANK) {
                if (DIRECT) s_ids[p] = id;
                else b.stage[w0 + p] = id;
                atomicOr(&s_tbits[p >> 5], 1u << (p & 31));
            } else if (n > 1) {
                const uint32_t item = (uint32_t)p | ((uint32_t)n << 16);
                if (TILE_LIST) {
                    // tile-owned: every miss goes on ONE list and through the segment pass of the tail
                    // (bpe_tail_segments: all of them tabulated together, merged side by side); queue
                    // mode keeps chunks of more than 64 bytes for the global queue
                    if (n <= 64 || !b.qcount) s_tmiss[atomicAdd(&s_nq[0], 1u)] = item;
                    else push_long(b, (uint32_t)(w0 + p), (uint32_t)n);
                }
                // multi-pass: short and medium chunks are merged right here by this workgroup (list in
                // LDS); long ones go to the global queue for k_bpe_long
                else if (n <= 16) s_miss[atomicAdd(&s_nq[0], 1u)] = item;
                else if (n <= 64) {
                    // (round-1 routing) multi-byte text of the single-pass tile: to the back of the long list
                    bool sent = false;
                    if (DIRECT && !b.qcount && ((s_txt[p] | s_txt[p + 1]) & 0x80u)) {
                        const uint32_t m = atomicAdd(&s_dq[11], 1u);
                        if (m < (uint32_t)DIRECT_LQ_MEDIUM) {
                            s_lq[2 * (DIRECT_LQCAP - 1 - m)] = (uint32_t)(w0 + p);
                            s_lq[2 * (DIRECT_LQCAP - 1 - m) + 1] = (uint32_t)n;
                            sent = true;
                        }
                    }
                    if (!sent) s_miss[G::C16 + atomicAdd(&s_nq[1], 1u)] = item;
                } else if (DIRECT && !b.qcount) {        // at most Wv / 65 of them
                    const uint32_t qi = atomicAdd(&s_dq[0], 1u);
                    s_lq[2 * qi] = (uint32_t)(w0 + p);
                    s_lq[2 * qi + 1] = (uint32_t)n;
                } else push_long(b, (uint32_t)(w0 + p), (uint32_t)n);
            }
        }
    }
    __syncthreads();
    SPL_STAMP(6);

    // ---- merge loop for this tile's misses: wavefronts pull work until both lists are empty ------
    // (scanner phases run at high priority, the merge loops below them: a workgroup that is still
    // scanning is never starved by older workgroups that already merge; +3 % on the bench batch)
    if (DIRECT) __builtin_amdgcn_s_setprio(SPL_MERGE_PRIO);
    if (!TILE_LIST) {
        const uint32_t m16 = s_nq[0], m64 = s_nq[1];
        // Short misses sorted by length, longest first (counting sort into s_cpos, which is free
        // until the tile record): the four chunks a wavefront merges in lock step then have similar
        // lengths -- a round lasts as long as its longest chunk -- and the longest chains start first.
#ifndef SPL_SORT_SHORT
#define SPL_SORT_SHORT 1         /* 0: the short misses in list order (A/B) */
#endif
        constexpr bool SORT_SHORT = SPL_SORT_SHORT && Wv <= 1024;            // window index (10 bits) | n - 1 (4 bits) in 16 bits
        if (SORT_SHORT) {
            uint32_t my_item[(G::C16 + NT - 1) / NT], my_r[(G::C16 + NT - 1) / NT];
#pragma unroll
            for (int q = 0; q < (G::C16 + NT - 1) / NT; q++) {
                const uint32_t k = tid + q * NT;
                if (k < m16) { my_item[q] = s_miss[k]; my_r[q] = atomicAdd(&s_scnt[16 - (my_item[q] >> 16)], 1u); }
            }
            __syncthreads();
            if (tid < 64) {                                // exclusive prefix sums of the 17 counts: one wavefront scan
                const uint32_t c = tid < 17 ? s_scnt[tid] : 0u;
                const uint32_t x = wave_scan_incl(c);
                if (tid < 17) s_scnt[tid] = x - c;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (G::C16 + NT - 1) / NT; q++) {
                const uint32_t k = tid + q * NT;
                if (k < m16) {
                    const uint32_t n = my_item[q] >> 16;
                    s_cpos[s_scnt[16 - n] + my_r[q]] = (uint16_t)((my_item[q] & 0x3FFu) | ((n - 1) << 10));
                }
            }
            __syncthreads();
        }
        uint32_t* const stage_w0 = b.stage + w0;          // window index -> global position
        // (A chunk may reach up to WPAD bytes beyond the window -- a character that straddles its end --, so a token
        //  inside it may START there: tile-owned mode keeps ids only for window positions, such a token goes the way
        //  of the tail's tokens beyond the window.  It used to be written behind s_ids and counted as a window token:
        //  a garbage id, found by the randomized stress run, seed 22739.)
        auto put = [&](int q, uint32_t id) {
            if (DIRECT && q >= Wv) {
                const uint32_t g = (uint32_t)(w0 + q);
                __hip_atomic_store(&b.stage[g], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(&b.tbits[g >> 5], 1u << (g & 31));
                atomicMax(&s_dq[4], g + 1u);
                return;
            }
            if (DIRECT) s_ids[q] = id;
            else stage_w0[q] = id;
            atomicOr(&s_tbits[q >> 5], 1u << (q & 31));
        };
        const int lane = tid & 63;
        // Every 17..64-byte chunk gets a whole wavefront (or half of one): lowest latency per merge -- their chains are the critical
        // path -- and with the ranks tabulated also the faster form on tiles dense with such chunks (8 MB of the C3 mix 1.19 ms
        // against 1.25 ms for 16-lane groups with four nodes per lane, the form of rounds 1-2, removed in round 4).
#ifdef SPL_DEBUG_STAMPS
        const long long ws_t0 = clock64();
        uint32_t ws_nmed = 0, ws_nshort = 0;
        long long ws_wt[6] = {0, 0, 0, 0, 0, 0};
#endif
        for (;;) {
            // A wavefront takes TWO chunks per pull: if both have at most 32 bytes (of ASCII: no independent
            // segments to look for) each gets a half of the wavefront and they merge side by side -- a tile
            // with several long words (the slowest tiles of the bench batch are those) needs half the pulls.
            uint32_t it = 0;
            if (lane == 0) it = atomicAdd(&s_nq[3], SPL_MEDIUM_PAIRS ? 2u : 1u);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= m64) break;
#ifdef SPL_DEBUG_STAMPS
            ws_nmed++;
#endif
            if (DIRECT && SPL_MEDIUM_PRIO != SPL_MERGE_PRIO) __builtin_amdgcn_s_setprio(SPL_MEDIUM_PRIO);
            const uint32_t itemA = s_miss[G::C16 + it];
            const uint32_t itemB = (SPL_MEDIUM_PAIRS && it + 1u < m64) ? s_miss[G::C16 + it + 1u] : 0u;
            const int pA = (int)(itemA & 0xFFFFu), nA = (int)(itemA >> 16), pB = (int)(itemB & 0xFFFFu), nB = (int)(itemB >> 16);
            bool pair = SPL_MEDIUM_PAIRS && nA <= 32 && nB <= 32;
            if (pair) {
                const int half = lane >> 5, hl = lane & 31;
                const int p = half ? pB : pA, n = half ? nB : nA;
                if (__any(hl < n && (s_txt[p + hl] & 0x80u))) pair = false;
                else {
#if defined(SPL_DEBUG_STAMPS) && defined(SPL_STAMP_MEDIUM)
                    long long* const wtm = (e_dbg && blockIdx.x == SPL_DBG_WG && ws_nmed == 1) ? ws_wt : nullptr;
#else
                    long long* const wtm = nullptr;
#endif
                    bpe_group_tab<32>(T, LdsAcc{s_rec, s_txt}, p, n, s_sub[(tid >> 6) * 4 + half * 2],
                                      [&](int i, uint32_t id) {
                                          put(p + i, id);
                                      }, wtm);
                }
            }
            if (!pair) {
                bpe_wave64_tab(T, LdsAcc{s_rec, s_txt}, pA, nA, s_sub[(tid >> 6) * 4],
                               [&](int i, uint32_t id) {
                                   put(pA + i, id);
                               });
                if (nB) bpe_wave64_tab(T, LdsAcc{s_rec, s_txt}, pB, nB, s_sub[(tid >> 6) * 4],
                                       [&](int i, uint32_t id) {
                                           put(pB + i, id);
                                       });
            }
        }
        if (DIRECT && SPL_MEDIUM_PRIO != SPL_MERGE_PRIO) __builtin_amdgcn_s_setprio(SPL_MERGE_PRIO);
        SPL_STAMP(9);
#ifdef SPL_DEBUG_STAMPS
        const long long ws_t1 = clock64();
#endif
        // every 16-lane group pulls its own short chunks (one node per lane)
        // The sorted list holds the chunks of 9..16 bytes first (items [0, first8)), then those of up to 8.  A SLOT is one
        // 16-lane group's work of a pull: one chunk of the first kind, or two of the second, one per half of the group.
        const uint32_t first8 = (SPL_PAIR_SHORT && SORT_SHORT) ? s_scnt[8] : m16;
        const uint32_t nslots = first8 + (m16 - first8 + 1u) / 2u;
        for (;;) {
            uint32_t it = 0;
            if ((lane & 15) == 0) it = atomicAdd(&s_nq[2], 1u);
            it = __shfl(it, lane & ~15);
            const bool slot = it < nslots;
            if (!__any(slot)) break;
#ifdef SPL_DEBUG_STAMPS
            ws_nshort++;
#endif
            const bool paired = slot && it >= first8;
            const uint32_t k = paired ? first8 + 2u * (it - first8) + (uint32_t)((lane >> 3) & 1) : it;
            const bool has = slot && k < m16;
            uint32_t item = 0;
            if (has) {
                if (SORT_SHORT) { const uint32_t c = s_cpos[k]; item = (c & 0x3FFu) | (((c >> 10) + 1u) << 16); }
                else item = s_miss[k];
            }
            const int p = (int)(item & 0xFFFFu);
#if defined(SPL_DEBUG_STAMPS) && !defined(SPL_STAMP_MEDIUM)
            long long* const wtp = (e_dbg && blockIdx.x == SPL_DBG_WG && ws_nshort == 1) ? ws_wt : nullptr;
#else
            long long* const wtp = nullptr;
#endif
            bpe_group16_tab(T, LdsAcc{s_rec, s_txt}, p, has ? (int)(item >> 16) : 0, s_sub[tid >> 4],
                            [&](int i, uint32_t id) {
                                put(p + i, id);
                            }, wtp, paired ? 8 : 16);
        }
#ifdef SPL_DEBUG_STAMPS
        if (e_dbg && blockIdx.x == SPL_DBG_WG && (tid & 63) == 0) {
            unsigned long long* r2 = e_dbg + 16 + 4 * (SPL_DEBUG_BLOCKS - 16 + 2 * (tid >> 6));
#ifdef SPL_STAMP_MEDIUM
            for (int k = 0; k < 6; k++) r2[k] = (unsigned long long)(ws_wt[k] - ws_t0);     // the first MEDIUM pull, since the medium loop began
#else
            for (int k = 0; k < 6; k++) r2[k] = (unsigned long long)(ws_wt[k] - ws_t1);
#endif
        }
#endif
#ifdef SPL_DEBUG_STAMPS
        if (e_dbg && blockIdx.x == SPL_DBG_WG && (tid & 63) == 0) {      // per-wavefront record of the middle workgroup
            unsigned long long* r = e_dbg + 16 + 4 * (SPL_DEBUG_BLOCKS - 8 + (tid >> 6));
            r[0] = (unsigned long long)(ws_t1 - ws_t0);
            r[1] = (unsigned long long)(clock64() - ws_t1);
            r[2] = (unsigned long long)ws_nmed | ((unsigned long long)ws_nshort << 32);
            r[3] = (unsigned long long)m16 | ((unsigned long long)m64 << 32);
        }
#endif
    }
    SPL_STAMP(10);
    __syncthreads();
    SPL_STAMP(7);
#ifdef SPL_DEBUG_STAMPS
    if (e_dbg) blk_w1 = blk_w2 = (unsigned long long)wall_clock64();
#endif
    {
        const int lane = tid & 63, wv = tid >> 6;
        const uint32_t ovf_lo = (uint32_t)(w0 + Wv);       // tokens from here on live in HBM (stage[] / tbits[])
        auto emit_g = [&](uint32_t q, uint32_t id) {
            const int64_t i = (int64_t)q - w0;
            if (i < (int64_t)Wv) {
                s_ids[i] = id;
                atomicOr(&s_tbits[i >> 5], 1u << (i & 31));
            } else {
                __hip_atomic_store(&b.stage[q], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(&b.tbits[q >> 5], 1u << (q & 31));
                atomicMax(&s_dq[4], q + 1u);
            }
        };
        // ---- the tile's misses (and, rarely, the chain that outgrew the window) ---------------------
        // Every chunk the whole-chunk probe missed is merged here, up to DIRECT_LQCAP of them at a time:
        // bpe_tail_segments lays them end to end over the table rows, fills the rows with two batches of
        // probes for ALL of them together and merges their segments side by side -- one lane per
        // segment of up to 8 bytes, 16 lanes up to 16, a wavefront beyond -- where the per-chunk route
        // paid a fill and a lock-step loop per group of four chunks.
        const uint32_t n_tm = TILE_LIST ? s_nq[0] : 0u;
#ifndef SPL_SKIP_TAIL
#define SPL_SKIP_TAIL 0          /* timing experiment only (tokens missing): the tile-owned tail does nothing */
#endif
        if (!SPL_SKIP_TAIL && (n_tm | s_dq[0] | s_dq[1] | s_dq[11])) {         // workgroup-uniform
            uint32_t mcur = 0;
            for (;;) {
#ifndef SPL_TAIL_LISTFILL_ALWAYS
#define SPL_TAIL_LISTFILL_ALWAYS 0
#endif
                if (TILE_LIST || SPL_TAIL_LISTFILL_ALWAYS) { // (only the tile-miss-list build moves misses onto the list here: two barriers)
                    const uint32_t have = s_dq[0];           // entries the chain continuation left on the list
                    uint32_t m = n_tm - mcur;
                    if (m > (uint32_t)DIRECT_LQCAP - have) m = (uint32_t)DIRECT_LQCAP - have;
                    if ((uint32_t)tid < m) {
                        const uint32_t item = s_tmiss[mcur + tid];
                        s_lq[2 * (have + tid)] = (uint32_t)(w0 + (item & 0xFFFFu));
                        s_lq[2 * (have + tid) + 1] = item >> 16;
                    }
                    __syncthreads();
                    if (tid == 0) s_dq[0] = have + m;
                    mcur += m;
                    __syncthreads();
                }
                // (round-1 routing: medium chunks sit at the back of the list, unused entries have length 0)
                const uint32_t nl0 = (!TILE_LIST && (s_dq[11] || s_dq[0] > (uint32_t)DIRECT_LQCAP)) ? (uint32_t)DIRECT_LQCAP : s_dq[0];
                // (the same value in every lane, read from LDS behind a barrier: as a scalar, so that the branch below is one)
                const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)bpe_tail_segments<2>(
                    T, b, s_lq, nl0, s_u.t.slab[0], reinterpret_cast<uint32_t*>(s_cpos), s_wsum, s_txt, w0, w0 + iT, emit_g));
                // (nl: the list's length, finished entries -- length 0 -- included; 0 if no chunk is left at all)
                if (!SPL_TAIL_SKIP_EMPTY || nl) {
                for (uint32_t it = wv; it < nl; it += NT / 64) {         // one wavefront per chunk
                    const int n = (int)s_lq[2 * it + 1];
                    const uint32_t pos = s_lq[2 * it];
                    uint32_t* const slab = s_u.t.slab[wv];
                    if (n >= 2 && n <= DIRECT_TAB_NMAX) {                // tabulated: no round trip per merge
                        bpe_wave_tab<DIRECT_TAB_NMAX / 64>(T, n, slab,
                            [&](int q) {
                                const uint64_t g = (uint64_t)pos + (uint32_t)q;
                                uint32_t w = 0;
                                if (g + 4 <= (uint64_t)B) __builtin_memcpy(&w, e_text + g, 4);
                                else for (int k = 0; k < 4; k++) if (g + k < (uint64_t)B) w |= (uint32_t)e_text[g + k] << (8 * k);
                                return w;
                            },
                            [&](int i, uint32_t id) { emit_g(pos + (uint32_t)i, id); });
                        wave_lds_sync();
                    }
                }
                __syncthreads();
                // 129..256 bytes: the LDS node list, a quarter of the slab per wavefront; 257..512
                // bytes: the same with half of the slab, two wavefronts (ONE call site: a second
                // instance of the merge loop costs the kernel registers it does not have).  The
                // workgroup-wide fallback for what is longer costs tens of microseconds per merge.
#pragma nounroll
                for (int pass = 0; pass < 2; pass++) {
                    const int cap = pass ? WAVE_NMAX : DIRECT_WAVE_NMAX, lo = pass ? DIRECT_WAVE_NMAX : DIRECT_TAB_NMAX;
                    const uint32_t nwav = pass ? 2u : 4u;
                    if ((uint32_t)wv < nwav) {
                        uint32_t* const slab = s_u.t.slab[pass ? 2 * wv : wv];
                        uint32_t seen = 0;
                        for (uint32_t it = 0; it < nl; it++) {
                            const int n = (int)s_lq[2 * it + 1];
                            if (n <= lo || n > cap) continue;
                            if ((seen++ % nwav) != (uint32_t)wv) continue;
                            bpe_wave(T, b, s_lq[2 * it], n, slab, slab + cap, reinterpret_cast<uint16_t*>(slab + 2 * cap),
                                     reinterpret_cast<uint16_t*>(slab + 2 * cap) + cap, emit_g);
                        }
                    }
                    __syncthreads();
                }
                for (uint32_t it = 0; it < nl; it++) {                   // oversize: the whole workgroup
                    const int n = (int)s_lq[2 * it + 1];
                    if (n > WAVE_NMAX) bpe_block_rounds(T, b, s_lq[2 * it], n, s_wsum, emit_g);
                }
                __syncthreads();
                }
                // continue the chain(s) that ran beyond the window: the workgroup stages the next DIRECT_WIN
                // bytes and their class records in LDS (in parallel), thread 0 walks the chain there --
                // whole-chunk hits become tokens at once, misses of ANY length refill the list for the
                // loops above.  (s_dq[7]: 0 no chain open, 1 the next chunk is the chain's first, 2 a
                // chunk only starts here if this is no sync point; + 4 / + 8 see below.)
                if (tid == 0) { s_dq[0] = 0; s_dq[11] = 0; }
                for (;;) {
                    __syncthreads();
                    const uint32_t ndc = s_dq[1] < 2u ? s_dq[1] : 2u;
                    if (s_dq[6] >= ndc || s_dq[0] >= (uint32_t)DIRECT_LQCAP) break;
                    if (ext) {
                        // external boundaries: the ONE chunk that starts at the deferred position ends at the next start
                        // bit, the next document, or the end of the corpus -- nothing to scan for
                        if (tid == 0) {
                            const uint32_t pc = s_dq[2 + s_dq[6]] & 0x7FFFFFFFu;
                            uint32_t lo = 0, hi = e_n_docs;             // first document that starts behind pc
                            while (lo < hi) {
                                const uint32_t mid = lo + (hi - lo) / 2;
                                if (e_doc_off[mid] <= (uint64_t)pc) lo = mid + 1; else hi = mid;
                            }
                            const uint32_t lim = lo < e_n_docs ? (uint32_t)e_doc_off[lo] : e_n_bytes;
                            uint32_t e = lim;
                            for (uint32_t w = (pc + 1u) >> 5; w * 32u < lim; w++) {
                                uint32_t word = b.ext_starts[w];
                                if (w == ((pc + 1u) >> 5)) word &= ~0u << ((pc + 1u) & 31u);
                                if (word) { const uint32_t q = w * 32u + (uint32_t)(__ffs((int)word) - 1); if (q < lim) e = q; break; }
                            }
                            const uint32_t n = e - pc;
                            // (a stretch of DROPPED bytes that outgrows the window -- a gap of a pattern that does not tile the
                            //  text, or a special literal's span -- is deferred like a chunk, but there is nothing to encode:
                            //  tools/dev/gpu_custom_stress.py found its bytes tokenised, 46 of 2 883 batches)
                            const bool dropped = b.ext_gaps && ((b.ext_gaps[pc >> 5] >> (pc & 31u)) & 1u) != 0u;
                            uint32_t fill = s_dq[0];
                            if (!dropped) {
                                const DirectAcc ga{&T, &b, lim, pc};
                                const uint32_t id = probe_chunk(T, ga, (int)pc, (int)n);
                                if (id != SPL_NO_RANK) emit_g(pc, id);
                                else if (n > 1) { s_lq[2 * fill] = pc; s_lq[2 * fill + 1] = n; fill++; }
                            }
                            s_dq[0] = fill;
                            s_dq[6] += 1;
                        }
                        continue;
                    }
                    if (tid == 0 && s_dq[7] == 0) {
                        const uint32_t pent = s_dq[2 + s_dq[6]], pc = pent & 0x7FFFFFFFu;   // (bit 31: only a chunk start if no sync point)
                        uint32_t lo = 0, hi = e_n_docs;         // first text start after the chain's start -- or AT it, if whether
                        while (lo < hi) {                        // a chunk of this chain starts there is still to be seen
                            const uint32_t mid = lo + (hi - lo) / 2;
                            if (e_doc_off[mid] + (uint64_t)(pent >> 31) <= (uint64_t)pc) lo = mid + 1; else hi = mid;
                        }
                        s_dq[5] = pc;
                        s_dq[8] = lo < e_n_docs ? (uint32_t)e_doc_off[lo] : e_n_bytes;
                        s_dq[7] = (pent >> 31) ? 2u : 1u;
                    }
                    __syncthreads();
                    const int64_t pc = s_dq[5];
                    const uint32_t next_ts = s_dq[8];
                    const uint32_t st = s_dq[7];                // 1 / 2 as above; + 4: splice a periodic run; + 8: walk it from HBM
                    uint8_t* const wtxt = reinterpret_cast<uint8_t*>(s_u.t.slab[0]);
                    uint8_t* const wrec = wtxt + DIRECT_WIN + 32;
                    const int64_t base = pc >= DEFER_BACK ? pc - DEFER_BACK : 0;
                    const int q0 = (int)(pc - base);
                    if (st & 8u) {                           // one chunk, byte-wise from HBM (no window could hold it)
                        if (tid == 0) {
                            uint32_t fill = s_dq[0];
                            const uint32_t np = (uint32_t)pc;
                            const DirectAcc ga{&T, &b, next_ts, np};
                            const int e = match_end(ga, (int)np, KPAT);
                            const uint32_t n = (uint32_t)e - np;
                            const uint32_t id = probe_chunk(T, ga, (int)np, (int)n);
                            if (id != SPL_NO_RANK) emit_g(np, id);
                            else if (n > 1) { s_lq[2 * fill] = np; s_lq[2 * fill + 1] = n; fill++; }
                            s_dq[0] = fill;
                            s_dq[5] = (uint32_t)e;
                            if ((uint32_t)e >= e_n_bytes) { s_dq[6] += 1; s_dq[7] = 0; }
                            else s_dq[7] = 2u;
                        }
                        continue;
                    }
                    // A chunk that no window holds is, in practice, one character repeated (64 KB of spaces):
                    // the text is periodic with the character's length P.  The window is then staged with the
                    // middle of that stretch cut out -- 16 characters of it stay on either side, what is cut
                    // is a whole number of characters from the inside of a run of identical ones, which no
                    // rule of the patterns can tell from a shorter run (no counted repeat is that long) -- and
                    // match_end's result is shifted by what was cut.  Found in parallel: 64 KB in 16 steps.
                    int split = 0x7FFFFFFF;                   // window index where the cut is
                    uint32_t removed = 0;
                    if (st & 4u) {
                        const int64_t lim = (int64_t)next_ts < B ? (int64_t)next_ts : B;
                        int tid_s = tid;                      // (as tid_late below: no 64-bit value derived from tid
                        asm volatile("" : "+v"(tid_s));      //  is kept from the kernel's start for this rare path)
                        int64_t g0 = pc + DIRECT_WIN / 2;
                        while (g0 > pc && (e_text[g0] & 0xC0u) == 0x80u) g0--;
                        const int P = (int)utf8_len(e_text[g0]);
                        if (tid == 0) { s_dq[9] = 0xFFFFFFFFu; s_dq[10] = 0; }
                        __syncthreads();
                        for (int64_t blk = g0;; blk += NT * 16) {       // first byte that differs from the one P further on
                            uint32_t bad = 0xFFFFFFFFu;
                            for (int k = 0; k < 16 && bad == 0xFFFFFFFFu; k++) {
                                const int64_t i = blk + tid_s * 16 + k;
                                if (i + P >= lim || e_text[i] != e_text[i + P]) bad = (uint32_t)i;
                            }
                            if (bad != 0xFFFFFFFFu) atomicMin(&s_dq[9], bad);
                            __syncthreads();
                            const bool found = s_dq[9] != 0xFFFFFFFFu;
                            __syncthreads();
                            if (found) break;
                        }
                        for (int64_t i = g0 - 1 - tid_s; i >= pc; i -= NT)  // and the last such byte before g0
                            if (i + P >= lim || e_text[i] != e_text[i + P]) { atomicMax(&s_dq[10], (uint32_t)(i - pc) + 1u); break; }
                        __syncthreads();
                        const int64_t e_per = (int64_t)s_dq[9] + P;      // the periodic text is [a_per, e_per)
                        const int64_t a_per = pc + (int64_t)s_dq[10];
                        const int64_t a_al = g0 - (g0 - a_per) / P * P;  // whole characters in phase with g0
                        const int64_t e_al = g0 + (e_per - g0) / P * P;
                        const int64_t head_end = a_al + 16 * P, tail_start = e_al - 16 * P;
                        if (tail_start <= head_end || head_end - base > DIRECT_WIN / 2 + 64 * 4) {
                            __syncthreads();
                            if (tid == 0) s_dq[7] = (st & 3u) | 8u;     // not periodic (enough): from HBM
                            continue;
                        }
                        removed = (uint32_t)(tail_start - head_end);
                        split = (int)(head_end - base);
                    }
                    const int64_t Bv = B - (int64_t)removed;           // length of the text as the window sees it
                    const int nst = (int)((Bv - base) < (int64_t)(DIRECT_WIN + 16) ? (Bv - base) : (int64_t)(DIRECT_WIN + 16));
                    const int nrec = nst < DIRECT_WIN ? nst + 1 : DIRECT_WIN;
                    for (int i = tid; i < DIRECT_WIN + 32; i += NT)
                        wtxt[i] = i < nst ? e_text[base + i + (i >= split ? (int64_t)removed : 0)] : (uint8_t)0;
                    __syncthreads();
                    for (int i = tid; i < nrec; i += NT) {
                        const int64_t g = base + i + (i >= split ? (int64_t)removed : 0);
                        uint32_t r;
                        if (g >= B) r = C_EOT | CB_TSTART | CB_SYNC;
                        else if (b.skip && ((b.skip[g >> 5] >> (g & 31)) & 1u)) r = C_EOT | CB_TSTART;
                        else {
                            // (the cut of a periodic run removes whole characters of a run of identical ones, so
                            //  the bytes on either side of it are what the look-back and the clamp would see anyway)
                            const WinAcc tx{wrec, wtxt, 0};
                            r = byte_record(T, tx,
                                            [&](int k) { const int64_t gg = base + k + (k >= split ? (int64_t)removed : 0);
                                                         return (uint32_t)gg == next_ts || (b.tstart && ((b.tstart[gg >> 5] >> (gg & 31)) & 1u)); },
                                            [&](uint32_t c) { return (uint32_t)s_ascii[c]; }, i, i >= q0 ? q0 : 0, nst);
                            if ((uint32_t)g == next_ts) r |= CB_TSTART | CB_SYNC;
                            if (b.tstart && ((b.tstart[g >> 5] >> (g & 31)) & 1u)) r |= CB_TSTART | CB_SYNC;
                        }
                        wrec[i] = (uint8_t)r;
                    }
                    __syncthreads();
                    if (tid == 0) {
                        const WinAcc acc{wrec, wtxt, nrec};
                        auto gpos = [&](int q) { return (uint32_t)(base + q + (q >= split ? (int64_t)removed : 0)); };
                        uint32_t fill = s_dq[0];
                        int q = q0;
                        bool fc = (st & 3u) == 1u, finished = false, whole = false, at_cut = false;
                        for (;;) {
                            if (!fc) {                               // does a chunk start here at all?
                                const uint32_t r = acc.rec(q);
                                if (r == (uint32_t)C_WEND) break;                  // the next window will tell
                                if (r & (CB_SYNC | CB_TSTART)) { finished = true; break; }
                                int j = q - 1;
                                while (j > 0 && (acc.rec(j) & CB_CLASS) == C_CONT && j > q - 4) j--;
                                const uint32_t prev = acc.rec(j) & CB_CLASS;
                                if (prev < C_EOT && is_sync(KPAT, prev, r & CB_CLASS)) { finished = true; break; }
                            }
                            if (fill >= (uint32_t)DIRECT_LQCAP) break;
                            const int e = match_end(acc, q, KPAT);
                            if (e == SPL_DEFER) { whole = q == q0; break; }       // (longer than a whole window: below)
                            fc = false;
                            const bool spans = q < split && e > split;             // the chunk the cut was made for
                            const uint32_t gp = gpos(q), n = (uint32_t)(e - q) + (spans ? removed : 0u);
                            const uint32_t id = spans ? SPL_NO_RANK : probe_chunk(T, acc, q, (int)n);   // (far beyond any token's length)
                            if (id != SPL_NO_RANK) emit_g(gp, id);
                            else if (n > 1) { s_lq[2 * fill] = gp; s_lq[2 * fill + 1] = n; fill++; }
                            q = e;
                            if ((int64_t)gpos(q) >= B) { finished = true; break; }
                            if (q == split) { at_cut = true; break; }              // (a chunk ended at the cut: plain windows from here)
                        }
                        s_dq[0] = fill;
                        if (whole) s_dq[7] = (fc ? 1u : 2u) | ((st & 4u) ? 8u : 4u);   // first the splice, then the walk from HBM
                        else {
                            s_dq[5] = at_cut ? (uint32_t)(base + q) : gpos(q);
                            if (finished) { s_dq[6] += 1; s_dq[7] = 0; }
                            else s_dq[7] = fc ? 1u : 2u;
                        }
                    }
                }
                __syncthreads();
                const uint32_t nd = s_dq[1] < 2u ? s_dq[1] : 2u;
                if (s_dq[0] == 0 && s_dq[6] >= nd && mcur >= n_tm) break;
            }
        }
        // SPL_WITH_SPECIAL: the literals that start in this tile are tokens of this tile (k_special_scan
        // left their ids in stage[] and marked their first bytes in tbits[], inside the skip spans)
        int tid_late = tid;                                  // (64-bit values derived from tid are rebuilt after
        asm volatile("" : "+v"(tid_late));                   //  the tail instead of living in registers across it)
        if (b.skip) {
            if (tid_late < G::NBW + 1) {
                const int64_t wi = (w0 >> 5) + tid_late;
                uint32_t sp = (wi >= 0 && wi * 32 < B) ? (b.tbits[wi] & s_sk[tid_late]) : 0u;
                const int lo = LH - tid_late * 32, hi = LH + TB_ - tid_late * 32;     // the tile's own range inside this word
                if (hi <= 0 || lo >= 32) sp = 0;
                else {
                    if (lo > 0) sp &= ~0u << lo;
                    if (hi < 32) sp &= (1u << hi) - 1u;
                }
                if (sp) atomicOr(&s_tbits[tid_late], sp);
                while (sp) {
                    const int bit = __ffs(sp) - 1;
                    sp &= sp - 1;
                    s_ids[tid_late * 32 + bit] = b.stage[w0 + tid_late * 32 + bit];
                }
            }
            __syncthreads();
        }
        // the first NT documents of the window are fetched now: their load overlaps the count below
        const bool last_tile = tile_ix == gridDim.x - 1;
        const uint64_t own_lo = (uint64_t)t0, own_hi = (uint64_t)(t0 + TB_);
        const uint64_t d_first = (uint64_t)dw + (uint32_t)tid_late;
        uint64_t p_first = ~0ull;
        if (d_first <= e_n_docs) p_first = e_doc_off[d_first];      // entry n_docs is the end of the corpus
        // ---- token count of the tile: window bitmap + overflow range --------------------------------
        uint32_t c_win;
        {
            uint32_t word = tid_late < G::NBW + 1 ? s_tbits[tid_late] : 0u;
            const uint32_t cnt = __popc(word);
            uint32_t x = wave_scan_incl(cnt);
            if (lane == 63) s_wsum[wv] = x;
            __syncthreads();
            uint32_t basew = x - cnt;
            for (int k = 0; k < wv; k++) basew += s_wsum[k];
            if (tid_late == NT - 1) s_total = basew + cnt;
            if (tid_late < G::NBW + 2) s_wpre[tid_late] = basew;
            while (word) {                                  // token positions in order
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                s_cpos[basew++] = (uint16_t)(tid_late * 32 + bit);
            }
            __syncthreads();
            c_win = s_total;
        }
        const uint32_t ovf_hi = s_dq[4];                     // exclusive; 0 if nothing went beyond the window
        const uint32_t wlo = ovf_lo >> 5, whi = ovf_hi > ovf_lo ? (ovf_hi + 31) >> 5 : wlo;
        uint32_t c_ovf = 0;
        if (whi > wlo) {
            uint32_t mine = 0;
            // (the range ends inside its last word: a special token's bit just behind it -- k_special_scan
            //  marks those in the same bitmap -- belongs to the tile that owns that byte)
            for (uint32_t w = wlo + tid_late; w < whi; w += NT) {
                uint32_t word = __hip_atomic_load(&b.tbits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w == whi - 1u && (ovf_hi & 31u)) word &= (1u << (ovf_hi & 31u)) - 1u;
                mine += __popc(word);
            }
            if (tid_late == 0) s_dq[9] = 0;
            __syncthreads();
            if (mine) atomicAdd(&s_dq[9], mine);
            __syncthreads();
            c_ovf = s_dq[9];
        }
        const unsigned long long total = (unsigned long long)c_win + c_ovf;
        // ---- the tile's record: packed window tokens, local document ranks, counts ------------------
        // (k_tile_out turns these into the final CSR once every tile's count is known; nothing here
        //  waits for another workgroup, so a tile that is slow -- long chunks, a chain that runs far
        //  beyond the window -- only delays itself)
        const bool queue_mode = b.qcount != nullptr;          // long chunks / chains went to the global queues
        if (tid_late == 0 && !queue_mode) atomicAdd(&b.tctl[16 + b.tpar * b.tgroups + (tile_ix >> 6)], (uint32_t)total);
        if (queue_mode && tid_late < TILE_BITS_W)
            b.tile_bits[(size_t)tile_ix * TILE_BITS_W + tid_late] = tid_late < G::NBW + 1 ? s_tbits[tid_late] : 0u;
#ifdef SPL_DEBUG_STAMPS
        if (e_dbg) blk_w2 = (unsigned long long)wall_clock64();
#endif
        const uint32_t slot = tile_ix * b.tslot;                // fixed slots: nothing to wait for
        for (uint32_t k = tid_late; k < c_win; k += NT) b.tile_ids[slot + k] = s_ids[s_cpos[k]];
        uint32_t d_lo = 0xFFFFFFFFu, d_n = 0;
        for (uint32_t db = dw;; db += NT) {
            const uint64_t d = (uint64_t)db + tid_late;
            uint64_t p = p_first;
            if (db != dw) { p = ~0ull; if (d <= e_n_docs) p = e_doc_off[d]; }
            const bool in = d <= e_n_docs && (p < own_hi || last_tile);
            const bool own = in && p >= own_lo;
            if (own && !queue_mode) {
                const uint32_t i = (uint32_t)(p - (uint64_t)w0);
                b.off_out[d] = (uint64_t)(s_wpre[i >> 5] + __popc(s_tbits[i >> 5] & ((1u << (i & 31)) - 1u)))
                               + ((last_tile && p >= (uint64_t)B) ? c_ovf : 0u);
            }
            // owned documents are consecutive: first index and count by ballots (s_wsum as mailboxes)
            const unsigned long long mo = __ballot(own);
            if (lane == 0) { s_red[wv] = mo; }
            __syncthreads();
            for (int k = 0; k < NT / 64; k++) {
                const unsigned long long mk = s_red[k];
                if (mk) {
                    if (d_lo == 0xFFFFFFFFu) d_lo = db + 64u * k + (uint32_t)(__ffsll((long long)mk) - 1);
                    d_n += (uint32_t)__popcll(mk);
                }
            }
            if (tid_late == NT - 1) s_dq[10] = in ? 1u : 0u;     // more documents beyond this batch of NT?
            __syncthreads();
            if (!s_dq[10]) break;
        }
        if (tid_late == 0) {
            TileDesc td;
            td.slot = slot; td.c_win = c_win; td.c_ovf = c_ovf; td.ovf_hi = whi > wlo ? ovf_hi : 0u;
            td.d_first = d_lo == 0xFFFFFFFFu ? 0u : d_lo; td.d_cnt = d_n; td.ovf_lo = ovf_lo;
            td.c_own = s_wpre[DIRECT ? (LH + TB_) >> 5 : 0];             // tile range ends on a word boundary
            b.tdesc[tile_ix] = td;
        }
    }
    SPL_STAMP(8);
#ifdef SPL_DEBUG_STAMPS
    if (e_dbg && tid == 0 && blockIdx.x == SPL_DBG_WG) e_dbg[12] = (unsigned long long)wall_clock64();
#ifdef SPL_STAMP_ALL
    if (e_dbg && tid == 0 && SPL_REC_BLK < SPL_DEBUG_BLOCKS / 2) {
        e_dbg[16 + 8 * SPL_REC_BLK] = blk_t0;
        e_dbg[16 + 8 * SPL_REC_BLK + 7] = (unsigned long long)wall_clock64();
    }
    if (false) {
#else
    if (e_dbg && tid == 0 && blockIdx.x < SPL_DEBUG_BLOCKS) {
#endif
        // wall-clock ticks: start, end of the merge phase, counts done, end
        unsigned long long* r = e_dbg + 16 + 4 * blockIdx.x;
        r[0] = blk_t0;
        r[1] = blk_w1;
        r[2] = blk_w2;
        r[3] = (unsigned long long)wall_clock64();
    }
#endif
    {
        int tid_end = tid;                                   // (as tid_late: nothing tid-derived kept for this)
        asm volatile("" : "+v"(tid_end));
        if (e_dbg && tid_end == 0) atomicMax(&e_dbg[15], (unsigned long long)wall_clock64());
    }
#ifdef SPL_PASSES
    }
#endif
#undef SPL_REC_BLK
#undef SPL_STAMP
}

// ------------------------------------------------------------------------------------------
// Tile-owned mode, second and last kernel: one workgroup per tile turns the tile records into the
// final CSR.  The number of tokens before a tile is the sum of the 64-tile group sums before its
// group (accumulated by k_pretok with one atomic per tile) plus the counts of the earlier tiles of
// its own group -- every workgroup computes its own base, there is no scan pass and nothing waits.
// Workgroup 0 also re-arms the next call: packing cursor and the OTHER parity's group sums to zero.
#ifndef SPL_TILE_OUT_NT
#define SPL_TILE_OUT_NT 128
#endif
constexpr int TOUT_NT = SPL_TILE_OUT_NT;               // threads of a k_tile_out workgroup (a tile has a few hundred tokens)
// What k_tile_out reads of the batch: two lines of argument segment instead of the six of a whole Batch (every wavefront of a launch
// waits for its freshly written arguments first: profiles/r03_launch_probes.txt)
struct TileOutArgs {
    uint32_t* tctl; const TileDesc* tdesc; const uint32_t* tile_ids; uint32_t* ids_out; uint64_t ids_cap; uint64_t* off_out; uint64_t* off_out2;
    uint32_t* slab; uint32_t* tbits; const uint32_t* stage; const uint32_t* skip;
    uint32_t tpar, tgroups, tslot, slab_cap, slab_max_docs, n_docs;
};
inline TileOutArgs tile_out_args(const Batch& b) {
    return TileOutArgs{b.tctl, b.tdesc, b.tile_ids, b.ids_out, b.ids_cap, b.off_out, b.off_out2, b.slab, b.tbits, b.stage, b.skip,
                       b.tpar, b.tgroups, b.tslot, b.slab_cap, b.slab_max_docs, b.n_docs};
}
__global__ __launch_bounds__(TOUT_NT) void k_tile_out(TileOutArgs b) {
    __shared__ unsigned long long s_part[TOUT_NT / 64];
    __shared__ uint32_t s_wsum[TOUT_NT / 64];
    const uint32_t t = xcd_tile();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t* gs = b.tctl + 16 + b.tpar * b.tgroups;
    const uint32_t g = t >> 6;
    // the tile's slot is fixed, so its first TOUT_NT tokens are fetched before the counts are known
    // (most tiles hold fewer): the copy below then depends on ONE round of loads, not two
    const uint32_t slot0 = t * b.tslot;
    const uint32_t first_id = b.tile_ids[slot0 + tid];
    unsigned long long mine = 0;
    for (uint32_t k = tid; k < g; k += TOUT_NT) mine += gs[k];
    {
        const uint32_t u = (g << 6) + (uint32_t)tid;
        if (tid < 64 && u < t) { const TileDesc q = b.tdesc[u]; mine += (unsigned long long)q.c_win + q.c_ovf; }
    }
    const TileDesc td = b.tdesc[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0) s_part[wv] = mine;
    __syncthreads();
    unsigned long long base = 0;
    for (int k = 0; k < TOUT_NT / 64; k++) base += s_part[k];
    const uint32_t s_ids_at = 3 + b.slab_max_docs, s_ids_cap = b.slab ? b.slab_cap - s_ids_at : 0u;
    for (uint32_t k = tid; k < td.c_win; k += TOUT_NT) {
        const unsigned long long r = base + k;
        const uint32_t id = k < (uint32_t)TOUT_NT ? first_id : b.tile_ids[td.slot + k];
        if (r < b.ids_cap) b.ids_out[r] = id;
        if (r < s_ids_cap) b.slab[s_ids_at + r] = id;
    }
    for (uint32_t k = tid; k < td.d_cnt; k += TOUT_NT) {
        const unsigned long long v = b.off_out[td.d_first + k] + base;
        b.off_out[td.d_first + k] = v;
        if (b.off_out2) b.off_out2[td.d_first + k] = v;
        if (b.slab && td.d_first + k <= b.slab_max_docs) b.slab[2 + td.d_first + k] = (uint32_t)v;
    }
    if (b.slab && t == gridDim.x - 1 && tid == 0) {        // header: T (the last tile ends the corpus), N
        b.slab[0] = (uint32_t)(base + td.c_win + td.c_ovf);
        b.slab[1] = b.n_docs;
    }
    if (td.ovf_hi > td.ovf_lo) {                             // rare: tokens that start beyond the window
        const uint32_t wlo = td.ovf_lo >> 5, whi = (td.ovf_hi + 31) >> 5;
        unsigned long long running = base + td.c_win;
        for (uint32_t wb = wlo; wb < whi; wb += TOUT_NT) {
            const uint32_t w = wb + tid;
            uint32_t word = w < whi ? b.tbits[w] : 0u;
            if (w == whi - 1u && (td.ovf_hi & 31u)) word &= (1u << (td.ovf_hi & 31u)) - 1u;     // (as in k_pretok's count)
            const uint32_t cnt = __popc(word);
            uint32_t x = wave_scan_incl(cnt);
            __syncthreads();
            if (lane == 63) s_wsum[wv] = x;
            __syncthreads();
            unsigned long long r = running + (x - cnt);
            uint32_t all = 0;
            for (int k = 0; k < TOUT_NT / 64; k++) { if (k < wv) r += s_wsum[k]; all += s_wsum[k]; }
            if (word && !b.skip) b.tbits[w] = 0u;           // clean after use: the bitmap is all-zero between calls
                                                            // (with special tokens it is cleared per call instead)
            while (word) {
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                if (r < b.ids_cap) b.ids_out[r] = b.stage[w * 32 + bit];
                if (r < s_ids_cap) b.slab[s_ids_at + r] = b.stage[w * 32 + bit];
                r++;
            }
            running += all;
        }
    }
    if (t == 0) {
        uint32_t* other = b.tctl + 16 + (b.tpar ^ 1u) * b.tgroups;
        for (uint32_t k = tid; k < b.tgroups; k += TOUT_NT) other[k] = 0u;
    }
}

// ------------------------------------------------------------------------------------------
// Queue mode (batches beyond the two-launch limit): k_pretok<.., DIRECT> works its tiles as in
// tile-owned mode but sends chunks of more than 64 bytes and chains that outgrow a window to the
// GLOBAL queues, where k_deferred_wave / k_bpe_segments / k_bpe_long balance them over the whole GPU and leave their tokens
// in stage[] / tbits[].  The CSR is then assembled per tile RANGE [t * TB, (t + 1) * TB): its
// tokens are the window tokens of tile t inside the range (A own), those of tile t - 1 that start
// beyond ITS range (A spill, at most the right halo) and the queue tokens of the range (B).
//   k_range_count: tokens per range -> tcnt[t], group sums
//   k_range_out  : base of the range (as k_tile_out), tokens in position order from the three
//                  sources, document offsets as ranks in the merged bitmap
template <int TB_, int RH_>
__device__ __forceinline__ void range_words(const Batch& b, uint32_t t, int j, uint32_t& a_own, uint32_t& a_spill,
                                            uint32_t& bq) {
    constexpr int W0 = LH / 32;                          // window word of the tile's first own byte
    constexpr int NOWN = TB_ / 32;                       // words of a range
    constexpr int NSP = RH_ / 32;                        // words the previous tile can spill into
    a_own = j < NOWN ? b.tile_bits[(size_t)t * TILE_BITS_W + W0 + j] : 0u;
    a_spill = (t > 0 && j < NSP) ? b.tile_bits[(size_t)(t - 1) * TILE_BITS_W + W0 + NOWN + j] : 0u;
    const uint64_t wg = (uint64_t)t * NOWN + (uint32_t)j;
    bq = (j < NOWN && wg * 32 < b.n_bytes) ? b.tbits[wg] : 0u;
}
template <int TB_, int RH_>
__global__ __launch_bounds__(64) void k_range_count(Batch b) {
    const uint32_t t = blockIdx.x;
    const int j = threadIdx.x;
    uint32_t ao, as, bq;
    range_words<TB_, RH_>(b, t, j, ao, as, bq);
    uint32_t c = __popc(ao | as | bq);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (j == 0) {
        b.tcnt[t] = c;
        atomicAdd(&b.tctl[16 + b.tpar * b.tgroups + (t >> 6)], c);
    }
}
template <int TB_, int RH_>
__global__ __launch_bounds__(64) void k_range_out(Batch b) {
    __shared__ uint32_t s_m[TB_ / 32 + 1], s_pre[TB_ / 32 + 1];
    const uint32_t t = blockIdx.x;
    const int j = threadIdx.x;
    constexpr int NOWN = TB_ / 32;
    // base of the range
    const uint32_t* gs = b.tctl + 16 + b.tpar * b.tgroups;
    const uint32_t g = t >> 6;
    unsigned long long mine = 0;
    for (uint32_t k = (uint32_t)j; k < g; k += 64) mine += gs[k];
    { const uint32_t u = (g << 6) + (uint32_t)j; if (u < t) mine += b.tcnt[u]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    const unsigned long long base = mine;
    // merged bitmap of the range and per-source prefix counts
    uint32_t ao, as, bq;
    range_words<TB_, RH_>(b, t, j, ao, as, bq);
    const uint32_t m = ao | as | bq;
    const uint32_t pm = wave_scan_incl(__popc(m)) - __popc(m);       // tokens of the range before this word
    const uint32_t pao = wave_scan_incl(__popc(ao)) - __popc(ao);   // own window tokens before this word
    const uint32_t pas = wave_scan_incl(__popc(as)) - __popc(as);
    if (j <= NOWN) { s_m[j] = j < NOWN ? m : 0u; s_pre[j] = pm; }
    const TileDesc td = b.tdesc[t];
    const uint32_t c_own_prev = t > 0 ? b.tdesc[t - 1].c_own : 0u;
    const uint32_t slot_own = t * b.tslot, slot_prev = (t > 0 ? t - 1 : 0u) * b.tslot;
    const uint64_t p0 = ((uint64_t)t * NOWN + (uint32_t)j) * 32;
    uint32_t word = m, r = pm;
    while (word) {
        const int bit = __ffs(word) - 1;
        const uint32_t below = (1u << bit) - 1u;
        word &= word - 1;
        uint32_t id;
        if ((ao >> bit) & 1u) id = b.tile_ids[slot_own + pao + __popc(ao & below)];
        else if ((as >> bit) & 1u) id = b.tile_ids[slot_prev + c_own_prev + pas + __popc(as & below)];
        else id = b.stage[p0 + bit];
        if (base + r < b.ids_cap) b.ids_out[base + r] = id;
        r++;
    }
    __syncthreads();
    // documents that start in the range: rank of their first byte in the merged bitmap
    const bool last_tile = t == gridDim.x - 1;
    for (uint32_t k = (uint32_t)j; k < td.d_cnt; k += 64) {
        const uint32_t d = td.d_first + k;
        const uint64_t p = b.doc_off[d];
        const uint64_t i = p - (uint64_t)t * TB_;                    // offset inside the range (== TB_ at most)
        const uint32_t w = (uint32_t)(i >> 5) < (uint32_t)NOWN ? (uint32_t)(i >> 5) : (uint32_t)NOWN;
        const uint32_t inword = w < (uint32_t)NOWN ? __popc(s_m[w] & ((1u << (i & 31)) - 1u)) : 0u;
        (void)last_tile;
        b.off_out[d] = base + s_pre[w] + inword;
    }
    if (t == 0) {                                        // re-arm the other parity's group sums (as k_tile_out)
        uint32_t* other = b.tctl + 16 + (b.tpar ^ 1u) * b.tgroups;
        for (uint32_t k = (uint32_t)j; k < b.tgroups; k += 64) other[k] = 0u;
    }
}

// Long chunks from the global queue, first pass: the segment merge of the tile tail
// (bpe_tail_segments) over batches of queue items.  A chunk it leaves -- one with a segment beyond
// 128 bytes -- goes, whole or what remains of it, on the survivor list for the node-list loops of
// k_bpe_long.
constexpr uint32_t SEG_BATCH = 12;
__global__ __launch_bounds__(NT) void k_bpe_segments(DeviceTables T, Batch b) {
    __shared__ __attribute__((aligned(16))) uint32_t s_slab[SEG_ROWS * SUB_W];
    __shared__ uint32_t s_lq[2 * DIRECT_LQCAP];
    __shared__ uint32_t s_scr[SG_WORDS];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_first;
    const int tid = threadIdx.x;
    const uint32_t nq = min(b.qcount[2], b.qcaplong), nbig = min(b.qcount[4], b.qcaplong), total = nq + nbig;
    uint2* const qbig = b.qlong + (b.qcaplong - 1u);
    auto slot = [&](uint32_t i) -> uint2* { return i < nq ? b.qlong + i : qbig - (i - nq); };
    for (uint32_t first = blockIdx.x * SEG_BATCH;;) {              // (the first batch is the workgroup's own index)
        if (first >= total) break;
        const uint32_t cnt = total - first < SEG_BATCH ? total - first : SEG_BATCH;
        if ((uint32_t)tid < cnt) {
            const uint2 item = *slot(first + tid);
            s_lq[2 * tid] = item.x;
            s_lq[2 * tid + 1] = item.y;
        }
        __syncthreads();
        const uint32_t nl2 = bpe_tail_segments<2>(T, b, s_lq, cnt, s_slab, s_scr, s_wsum, nullptr, 0, 0,
                                               [&](uint32_t q, uint32_t id) { emit_token(b, q, id); });
        // what is left -- chunks not finished, segments set aside -- goes on the survivor list (q64: a
        // dense list that k_bpe_long walks one item per wavefront; the long queue itself is done with)
        if ((uint32_t)tid < nl2 && s_lq[2 * tid + 1] >= 2u) {
            const uint32_t qi = atomicAdd(&b.qcount[0], 1u);
            if (qi < b.qcap64) b.q64[qi] = make_uint2(s_lq[2 * tid], s_lq[2 * tid + 1]);
        }
        if (tid == 0) s_first = gridDim.x * SEG_BATCH + atomicAdd(&b.qcount[8], SEG_BATCH);
        __syncthreads();
        first = s_first;
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void k_bpe_long(DeviceTables T, Batch b, int survivors) {
    __shared__ uint32_t s_id[NT / 64][WAVE_NMAX];
    __shared__ uint32_t s_rk[NT / 64][WAVE_NMAX];
    __shared__ uint16_t s_nx[NT / 64][WAVE_NMAX];
    __shared__ uint16_t s_pv[NT / 64][WAVE_NMAX];
    __shared__ uint32_t s_red4[NT / 64];
    static_assert((NT / 64) * WAVE_NMAX == BLOCK_LDS_NMAX, "the four slabs together hold the workgroup-wide list");
    static_assert(LONG_SMALL_NMAX == GROUP_NMAX, "the front of the queue is what the group phase takes");
    const uint32_t nq = min(b.qcount[2], b.qcaplong);          // front: chunks of up to GROUP_NMAX bytes
    const uint32_t nbig = min(b.qcount[4], b.qcaplong);        // back: larger ones
    const uint2* const qbig = b.qlong + (b.qcaplong - 1u);     // item k of the back is qbig[-k]
    const int wv = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (NT / 64);
    // Work is pulled dynamically (one atomic per wavefront and pull): chunk lengths range from 65
    // to several hundred bytes, and a static split leaves most wavefronts idle behind the longest.
    // wavefront phase FIRST (the longest chains start earliest): GROUP_NMAX < n <= WAVE_NMAX
    {
        const int lane = threadIdx.x & 63;
        // (the first item of every wavefront is its own index, later ones come from the cursor: an
        //  empty or short queue costs no atomics at all)
        const uint32_t wgid = blockIdx.x * (NT / 64) + wv;
        if (survivors) {
            // after k_bpe_segments: the survivor list, any length, one item per wavefront and pull
            const uint32_t ns = min(b.qcount[0], b.qcap64);
            for (uint32_t it = wgid; it < ns;) {
                const uint2 item = b.q64[it];
                if ((int)item.y <= WAVE_NMAX)
                    bpe_wave(T, b, item.x, (int)item.y, s_id[wv], s_rk[wv], s_nx[wv], s_pv[wv],
                             [&](uint32_t q, uint32_t id) { emit_token(b, q, id); });
                uint32_t nxt = 0;
                if (lane == 0) nxt = atomicAdd(&b.qcount[6], 1u);
                it = nwaves + __builtin_amdgcn_readfirstlane(nxt);
            }
        } else {
        for (uint32_t it = wgid; it < nbig;) {
            const uint2 item = *(qbig - it);
            if ((int)item.y <= WAVE_NMAX)
                bpe_wave(T, b, item.x, (int)item.y, s_id[wv], s_rk[wv], s_nx[wv], s_pv[wv],
                         [&](uint32_t q, uint32_t id) { emit_token(b, q, id); });
            uint32_t nxt = 0;
            if (lane == 0) nxt = atomicAdd(&b.qcount[6], 1u);
            it = nwaves + __builtin_amdgcn_readfirstlane(nxt);
        }
        // group phase: chunks of up to GROUP_NMAX bytes, four per wavefront, nodes in registers
        for (uint32_t base = wgid * 4; base < nq;) {
            const uint32_t it = base + (lane >> 4);
            uint2 item = make_uint2(0, 0);
            if (it < nq) item = b.qlong[it];
            const bool has = it < nq;
            if (__any(has)) {
                const uint32_t pos = item.x;
                bpe_group16<GROUP_NMAX / 16>(T, has ? (int)item.y : 0, [&](int i) { return (uint32_t)b.text[pos + i]; },
                                             [&](int i, uint32_t id) { emit_token(b, pos + (uint32_t)i, id); });
            }
            uint32_t nxt = 0;
            if (lane == 0) nxt = atomicAdd(&b.qcount[7], 4u);
            base = nwaves * 4 + __builtin_amdgcn_readfirstlane(nxt);
        }
        }
    }
    __syncthreads();
    // workgroup phase: the oversize items among this workgroup's share (uniform loop for all threads)
    const uint32_t nover = survivors ? min(b.qcount[0], b.qcap64) : nbig;
    for (int w = 0; w < NT / 64; w++)
        for (uint32_t it = blockIdx.x * (NT / 64) + w; it < nover; it += nwaves) {
            const uint2 item = survivors ? b.q64[it] : *(qbig - it);
            if ((int)item.y > WAVE_NMAX && (int)item.y <= BLOCK_LDS_NMAX)     // the four wavefront slabs as ONE list
                bpe_block_lds(T, b, item.x, (int)item.y, &s_id[0][0], &s_rk[0][0], &s_nx[0][0], &s_pv[0][0], s_red4,
                              [&](uint32_t q, uint32_t id) { emit_token(b, q, id); });
            else if ((int)item.y > BLOCK_LDS_NMAX)
                bpe_block_rounds(T, b, item.x, (int)item.y, s_red4,
                                 [&](uint32_t q, uint32_t id) { emit_token(b, q, id); });
        }
}


// ------------------------------------------------------------------------------------------
// decode_bytes (reference src/core/tokenizer.rs:877-897, batch form :945-958): gather token byte
// strings.  The id -> bytes table covers the vocabulary AND the special tokens (the reference looks
// an id up in `decoder` first, then in `special_tokens_decoder`; an id in neither contributes
// nothing).  Three launches, no host round trip in between:
//   k_decode_len    length of every id + sums per block of DEC_BLK ids
//   k_decode_scan   exclusive scan of the block sums (one workgroup)
//   k_decode_copy   offset of every id (block base + scan inside the block), byte copy, and the
//                   byte offset of every document (doc d starts at id ids_off[d])
constexpr int DEC_BLK = 1024;
struct DecodeArgs {
    const uint32_t* ids; uint64_t n_ids;
    const uint32_t* tok_off; const uint8_t* tok_bytes; uint32_t max_id;
    // special tokens whose ids lie beyond the vocabulary's largest id: sorted ids, byte spans sp_off[k] .. sp_off[k + 1]
    // of tok_bytes (a dense table up to the largest SPECIAL id would be O(that id): spl_add_special takes any id < 2^31)
    const uint32_t* sp_ids; const uint32_t* sp_off; uint32_t n_sp;
    uint64_t* blk;          // [n_blk + 1] block sums, then exclusive offsets (+ total)
    uint64_t* id_off;       // [n_ids + 1] byte offset of every id (+ total)
    uint8_t* out;
    const uint64_t* doc_first; uint64_t n_docs; uint64_t* doc_off;   // doc d = ids [doc_first[d] - doc_first[0], ...)
};
__device__ __forceinline__ uint32_t dec_span(const DecodeArgs& a, uint32_t id, uint32_t& off) {
    if (id <= a.max_id) { off = a.tok_off[id]; return a.tok_off[id + 1] - off; }
    uint32_t lo = 0, hi = a.n_sp;                        // (a few dozen entries at most; ids of real text never get here)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.sp_ids[mid] < id) lo = mid + 1; else hi = mid; }
    if (lo < a.n_sp && a.sp_ids[lo] == id) { off = a.sp_off[lo]; return a.sp_off[lo + 1] - off; }
    off = 0;
    return 0u;
}
__device__ __forceinline__ uint32_t dec_len(const DecodeArgs& a, uint64_t i) {
    if (i >= a.n_ids) return 0u;
    uint32_t off;
    return dec_span(a, a.ids[i], off);
}
__global__ __launch_bounds__(NT) void k_decode_len(DecodeArgs a) {
    __shared__ uint32_t s_w[NT / 64];
    uint32_t sum = 0;
    const uint64_t base = (uint64_t)blockIdx.x * DEC_BLK;
    for (int k = 0; k < DEC_BLK / NT; k++) sum += dec_len(a, base + (uint64_t)k * NT + threadIdx.x);
    sum = wave_scan_incl(sum);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < NT / 64; w++) t += s_w[w];
        a.blk[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_decode_scan(uint64_t* blk, uint64_t n_blk) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_blk; base += 1024) {
        const uint64_t i = base + tid;
        const uint64_t v = i < n_blk ? blk[i] : 0ull;
        uint64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if ((tid & 63) >= d) x += y; }
        if ((tid & 63) == 63) s_w[tid >> 6] = x;
        __syncthreads();
        uint64_t pre = s_carry;
        for (int w = 0; w < (tid >> 6); w++) pre += s_w[w];
        if (i < n_blk) blk[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) blk[n_blk] = s_carry;
}
__global__ __launch_bounds__(NT) void k_decode_copy(DecodeArgs a) {
    __shared__ uint32_t s_w[NT / 64];
    const uint64_t base = (uint64_t)blockIdx.x * DEC_BLK;
    uint64_t run = a.blk[blockIdx.x];
    for (int k = 0; k < DEC_BLK / NT; k++) {
        const uint64_t i = base + (uint64_t)k * NT + threadIdx.x;
        const uint32_t len = dec_len(a, i);
        const uint32_t x = wave_scan_incl(len);
        __syncthreads();
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        uint64_t o = run + (x - len);
        uint32_t all = 0;
        for (int w = 0; w < NT / 64; w++) { if (w < (int)(threadIdx.x >> 6)) o += s_w[w]; all += s_w[w]; }
        if (i < a.n_ids) {
            a.id_off[i] = o;
            uint32_t soff;
            (void)dec_span(a, a.ids[i], soff);
            const uint8_t* src = a.tok_bytes + soff;
            for (uint32_t q = 0; q < len; q++) a.out[o + q] = src[q];
        }
        run += all;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.id_off[a.n_ids] = run;
}
__global__ void k_decode_docs(DecodeArgs a) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > a.n_docs) return;
    a.doc_off[d] = a.id_off[a.doc_first[d] - a.doc_first[0]];
}

// External chunk boundaries with special tokens: the host splitter found the literals too; their ids go where
// k_special_scan would have put them (the tile that owns a literal's first byte takes it as its token).
__global__ void k_ext_specials(Batch b, const uint32_t* pos, const uint32_t* id, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pos[i];
    b.stage[p] = id[i];
    atomicOr(&b.tbits[p >> 5], 1u << (p & 31));
}

// Host pipeline (spl_encode_batch): chunk-local output offsets -> offsets in the whole result.
__global__ void k_add_base(uint64_t* p, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += base;
}

// ------------------------------------------------------------------------------------------
// Ragged all-gather support (multi-GPU reassembly of the CSR result).  RCCL has no all-gatherv:
// every rank packs {T, N, local offsets[N+1], ids[T]} into a fixed-capacity slab, ONE
// all_gather_into_tensor moves the slabs over xGMI, and every rank unpacks them into the global
// CSR.  No host synchronisation: the token counts travel inside the slabs.
//   slab (u32 words): [0] T  [1] N  [2 .. 2+max_docs] local out_off (N+1 used)  [2+max_docs+1 ..] ids
__global__ void k_gatherv_pack(const uint32_t* ids, const uint64_t* out_off, uint32_t n_docs, uint32_t* slab,
                               uint32_t cap_words, uint32_t max_docs) {
    const uint32_t T = (uint32_t)out_off[n_docs];
    const uint32_t ids_at = 3 + max_docs, ids_cap = cap_words - ids_at;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i == 0) { slab[0] = T; slab[1] = n_docs; }
    for (uint32_t d = i; d <= n_docs; d += stride) slab[2 + d] = (uint32_t)out_off[d];
    const uint32_t ncopy = T < ids_cap ? T : ids_cap;        // T > ids_cap is reported by the unpacker
    for (uint32_t k = i; k < ncopy; k += stride) slab[ids_at + k] = ids[k];
}
// grid.y = source rank, grid.z = batch of the group (a rank sends `depth` slabs back to back per
// collective: rank_stride = depth * cap_words; batch j's slabs start at j * cap_words and its outputs
// at j * all_ids_cap / j * off_stride).  status[0] is set to 1 if any slab overflowed its id capacity.
__global__ void k_gatherv_unpack(const uint32_t* slabs_all, uint32_t world, uint32_t cap_words, uint32_t max_docs,
                                 uint32_t* all_ids_all, uint64_t all_ids_cap, uint64_t* all_off_all, uint32_t* status,
                                 uint64_t rank_stride, uint64_t off_stride) {
    const uint32_t r = blockIdx.y, j = blockIdx.z;
    const uint32_t* slabs = slabs_all + (size_t)j * cap_words;
    uint32_t* all_ids = all_ids_all + (size_t)j * all_ids_cap;
    uint64_t* all_off = all_off_all + (size_t)j * off_stride;
    const uint32_t ids_at = 3 + max_docs, ids_cap = cap_words - ids_at;
    uint64_t tbase = 0, dbase = 0;
    for (uint32_t q = 0; q < r; q++) { tbase += slabs[(size_t)q * rank_stride]; dbase += slabs[(size_t)q * rank_stride + 1]; }
    const uint32_t* slab = slabs + (size_t)r * rank_stride;
    const uint32_t T = slab[0], N = slab[1];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i == 0 && T > ids_cap) status[0] = 1;
    for (uint32_t d = i; d < N; d += stride) all_off[dbase + d] = tbase + slab[2 + d];
    if (r == world - 1 && i == 0) all_off[dbase + N] = tbase + T;
    const uint32_t ncopy = T < ids_cap ? T : ids_cap;
    for (uint32_t k = i; k < ncopy; k += stride)
        if (tbase + k < all_ids_cap) all_ids[tbase + k] = slab[ids_at + k];
}

// Exact ragged all-gather (spl_allgatherv_csr): every rank's {T, N} travel first, then exactly T ids and N
// offsets per rank land at their place of the global CSR by grouped send / recv.  These two kernels are the
// device side: the counts as the collective's input, and the received LOCAL offsets rebased by the tokens of the
// ranks before (+ the closing entry).
constexpr int COMM_MAX_WORLD = 64;
struct RankTable { uint64_t n_pre[COMM_MAX_WORLD + 1], t_pre[COMM_MAX_WORLD + 1]; };
__global__ void k_csr_counts(const uint64_t* out_off, uint64_t n_docs, uint64_t ids_cap, uint64_t off_cap, uint64_t* cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { cnt[0] = out_off[n_docs]; cnt[1] = n_docs; cnt[2] = ids_cap; cnt[3] = off_cap; }
}
__global__ void k_rebase_offsets(uint64_t* all_off, RankTable tab, uint32_t world) {
    const uint32_t r = blockIdx.y;
    const uint64_t lo = tab.n_pre[r], hi = tab.n_pre[r + 1], add = tab.t_pre[r];
    for (uint64_t d = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < hi; d += (uint64_t)gridDim.x * blockDim.x) all_off[d] += add;
    if (r == world - 1 && blockIdx.x == 0 && threadIdx.x == 0) all_off[tab.n_pre[world]] = tab.t_pre[world];
}

}  // namespace spl
