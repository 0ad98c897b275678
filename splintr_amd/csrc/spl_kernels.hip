// spl_kernels.hip -- gfx950 kernels of the batch encode path (DESIGN.md 4 has the full table).  ONE translation unit (spl_api.hip
// includes this file), in parts since round 4:
//
//   this file          constants, TileDesc, Batch (the per-call argument struct), small wave helpers
//   spl_k_special.h    k_mark_docs / k_special_scan / _ends / _select: text-start bitmap from the document offsets,
//                      special-token literals (SPL_WITH_SPECIAL)
//   spl_k_merge.h      text accessors, wave-level minima and scans, the vocabulary probes, byte_pair_encode
//                      (src/core/bpe.rs:67-197) as merge loops with one node per lane and tabulated pair ranks;
//                      queue mode's k_deferred_wave
//   spl_k_tile.h       tile geometry, LDS layout, the tail that finishes a tile's long chunks (bpe_tail_segments)
//   spl_k_fuse.h       the fused mode (ONE launch): tiles publish their token counts and place their part of the CSR themselves
//   spl_k_memo.h       the chunk memo: its probe, the log of what it did not hold, k_memo_fill (between two launches)
//   spl_k_pretok.h     k_pretok<tile, halo>: one workgroup per tile -- stage the window in LDS, classify code points into
//                      class bit masks (spl_scan_words.h), all match starts by bit-vector arithmetic (spl_scan_starts.h),
//                      whole-chunk vocabulary probe (spl_lookup.h), merge loops for the tile's misses, the tile's record
//                      -- reference: Tokenizer::encode, src/core/tokenizer.rs:729-808
//   spl_k_output.h     k_tile_out: tile records -> dense ids[] and per-document offsets (CSR); queue mode's
//                      k_range_count / k_range_out, k_bpe_segments, k_bpe_long
//   spl_k_decode.h     id -> bytes gather (k_decode_*), k_ext_specials, slabs around the RCCL all-gather, CSR rebase
//   (spl_rx_split.h, included by spl_api.hip: the device splitter for custom split patterns)
// (The multi-pass pipeline of rounds 1-3 -- k_bpe_lanes64, k_count, k_scan, k_compact_docs and the k_pretok
//  instantiations without tile records -- was removed in round 4: no BASELINE configuration reached it.)
//
// Token bookkeeping: a token is identified by the byte position where it starts; a tile keeps a bitmap of token starts and
// the ids at their positions in LDS, so ranks are popcount prefix sums and no lane needs to know how many tokens another
// produced.
#include <hip/hip_runtime.h>

#include "spl_common.h"
#include "spl_lookup.h"
#include "spl_scan.h"
#include "spl_scan_masks.h"
#include "spl_scan_starts.h"
#include "spl_scan_words.h"

#define SPL_DBG_WG (b.dbg_wg == 0xFFFFFFFFu ? gridDim.x / 2 : b.dbg_wg)


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
    uint32_t slab_p24;         // the slab's ids are packed three bytes each (ids < 2^21: a quarter less on the links)
    uint32_t tpar;             // parity of this call
    uint64_t* off_out2;        // tile-owned mode, optional: k_tile_out stores the final offsets here as well (pinned host memory)
    // EXTERNAL chunk boundaries (split patterns the scanner does not implement: the host splitter, spl_regex.h):
    // bit p of ext_starts: a chunk -- or a stretch of dropped bytes -- starts at byte p; bit p of ext_gaps: byte p is
    // dropped (no match covers it).  Tile-owned mode only; the scanner phases are skipped.
    const uint32_t* ext_starts; const uint32_t* ext_gaps;
    uint32_t* done; uint32_t done_seq;       // tile-owned mode, optional: k_tile_out's completion word in pinned host memory (spl_k_output.h)
    // fused mode (ONE launch, spl_k_fuse.h; ftc == nullptr: the two-launch form).  This call's parity: ftc[tile] = the tile's token
    // count + 1 (0: not known yet; 0xFFFF: ftb[tile] holds it, 32 bits).  fzc / fzb: the other parity's arrays, of which tile 0 zeroes
    // the first fz_n entries (what the previous fused launch used).
    uint16_t* ftc; uint32_t* ftb; uint16_t* fzc; uint32_t* fzb; uint32_t fz_n;
    const unsigned long long* gpre;     // tile-owned mode, batches of many tiles: the groups' exclusive prefix sums (k_group_scan, spl_k_output.h); null: k_tile_out adds the sums up itself
    uint32_t tile0;          // the first tile of this launch: a large batch goes out as several launches over ranges of its tiles (launch_all; 0 otherwise)
    // chunk memo (spl_k_memo.h; mlog == nullptr: the tiles log nothing): SPL_MEMO_LOG_REGIONS regions of mlog_cap entries of SPL_MEMO_LOG_WORDS
    // words each, their fill counters, and the word in pinned host memory that tells the host there is something to put in
    uint32_t* mlog; uint32_t* mlog_cnt; uint32_t mlog_cap; uint32_t* mflag;
    uint32_t* mlog2; uint32_t mlog2_cap;      // ... of chunks of 33..64 bytes (SPL_MEMO_LOG_WORDS2 words an entry; counters: mlog_cnt + SPL_MEMO_LOG_REGIONS; nullptr: not logged)
    uint32_t id_limit;         // DeviceTables::id_limit, for the kernels that are given no tables
};

// Workgroup barrier for hand-overs through LDS ONLY: __syncthreads() also waits for the wavefront's outstanding global stores (its release
// fence), which in the fused mode are the count just published -- a microsecond, write-through past the L2 (profiles/r06_one_launch.txt).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// Slab ids (the ragged all-gather's wire format, spl_k_decode.h): u32 each, or -- "slab_pack24" -- three bytes each, little endian.
__device__ __forceinline__ void slab_put_id(uint32_t* area, uint32_t r, uint32_t id, bool p24) {
    if (!p24) { area[r] = id; return; }
    uint8_t* p = reinterpret_cast<uint8_t*>(area) + 3u * (size_t)r;
    p[0] = (uint8_t)id; p[1] = (uint8_t)(id >> 8); p[2] = (uint8_t)(id >> 16);
}
__device__ __forceinline__ uint32_t slab_get_id(const uint32_t* area, uint32_t k, bool p24) {
    if (!p24) return area[k];
    const size_t bo = 3u * (size_t)k;
    const uint32_t lo = area[bo >> 2], hi = area[(bo >> 2) + 1];            // (the area is padded by a word)
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(bo & 3)) & 0xFFFFFFu;
}
__device__ __forceinline__ uint32_t slab_id_cap(uint32_t area_words, bool p24) { return p24 ? (uint32_t)(((size_t)(area_words - 1u) * 4u) / 3u) : area_words; }

__device__ __forceinline__ uint32_t tidx() {
    uint32_t x = threadIdx.x;
    asm volatile("" : "+v"(x));
    return x;
}

}  // namespace spl

#include "spl_k_special.h"
#include "spl_k_merge.h"
#include "spl_k_tile.h"
#include "spl_k_fuse.h"
#include "spl_k_memo.h"
#include "spl_k_pretok.h"
#include "spl_k_output.h"
#include "spl_k_decode.h"
