// spl_k_fuse.h -- part of spl_kernels.hip (included there, in this order; one translation unit): the fused mode -- every tile of a launch that is resident at once learns the number of tokens in front of it from the other tiles' published counts and writes its part of the CSR itself (round 6: ONE launch).
#pragma once

namespace spl {

// Round 6: the step as ONE launch (batches of up to fuse_max_tiles tiles -- every workgroup resident at once).
// Up to round 5 a second kernel (k_tile_out) added every tile's base -- the number of tokens in front of it -- once every tile's
// count was known: 4.6 us of a 31.5 us step, more than all merge loops together.  Now a tile
//   * PUBLISHES its token count the moment it is known (one 16-bit store, agent scope: ftc[tile] = count + 1), a few microseconds
//     before its own record would have been complete;
//   * one of its wavefronts READS the counts of all tiles in front of it -- four per lane and load, six loads in flight for 1536
//     tiles -- and goes on reading the words that are still incomplete (the others are summed once and never fetched again)
//     until none is;
//   * then writes its ids from LDS, its documents' offsets and its part of the optional slab to their FINAL place.
// No tile_ids[], no tile records, no second pass over them: from the slowest tile's count to the end of the launch it is one
// store, one load and the result stores.  Stores and loads only -- the r02 experiment (profiles/r02_single_launch_experiment.txt)
// found read-modify-write atomics on words that others poll to be ruinous, and release / acquire fences at agent scope to write
// back and invalidate the L2 with the vocabulary tables in it; relaxed agent-scope accesses do neither.
// A tile waits only for tiles with a LOWER index, and in this mode the tile index is the workgroup index (no XCD map): whatever a
// workgroup waits for was dispatched before it, so the launch makes progress whatever else runs on the GPU.  What does not hold
// beyond residency is the speed -- a slow tile would hold every later workgroup's slot (8 MB of CJK-heavy text: 4.4 against 1.3
// ms, round 2) --, hence the size limit; larger batches keep the two-launch form.
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int FUSE_NCH = 6;                                   // u64 words (four counts each) per lane of the polling wavefront
constexpr uint32_t FUSE_MAX_TILES = 4u * 64u * FUSE_NCH;      // 1536: what is resident at once (256 CUs x 6 workgroups)
// The counts are REPLICATED: a tile stores its count into FUSE_REPL copies of the array (one store instruction, one lane per copy) and
// reads the copy tile % FUSE_REPL.  A thousand wavefronts polling the same forty cache lines are served one after the other -- about
// 9 ns per request and line, measured (profiles/r06_one_launch.txt): a round of polls took 5-7 us, and the last tile to publish
// learned its own base 5 us later; with sixteen copies a line has a sixteenth of the readers.
#ifndef SPL_FUSE_REPL
#define SPL_FUSE_REPL 16
#endif
constexpr uint32_t FUSE_REPL = SPL_FUSE_REPL;
constexpr uint32_t FUSE_STRIDE = FUSE_MAX_TILES;              // u16 entries between two copies (a multiple of 64: copies start on a cache line)
static_assert(FUSE_REPL >= 1 && FUSE_REPL <= 63, "one lane per copy");
constexpr size_t FUSE_PARITY_BYTES = (size_t)FUSE_REPL * FUSE_STRIDE * 2 + (size_t)FUSE_STRIDE * 4;   // one parity's arrays
#ifndef SPL_FUSE_SLEEP
#define SPL_FUSE_SLEEP 4          /* s_sleep between two looks at the watched word, x 64 clocks */
#endif
#ifndef SPL_FUSE_FEW
#define SPL_FUSE_FEW 6            /* up to this many incomplete words are all read again in every round; beyond it ONE of them is watched */
#endif

// A tile's count, for the tiles behind it: lanes 0 .. FUSE_REPL - 1 of ONE wavefront, `total` the same in all of them.  Counts of 0xFFFE and
// more (a chain of 64 K tokens beyond the window) go to the 32-bit side array (one copy: rare), the 16-bit word says so.
__device__ __forceinline__ void fuse_publish(const Batch& b, uint32_t tile, uint32_t total) {
    const uint32_t r = tidx() & 63u;
    if (r >= FUSE_REPL) return;
    const bool big = total + 1u >= 0xFFFFu;
    if (big && r == 0u) st_agent(b.ftb + tile, total + 1u);
    st_agent(b.ftc + r * FUSE_STRIDE + tile, (uint16_t)(big ? 0xFFFFu : total + 1u));
}
// The tile's entries of the OTHER parity (what the previous fused launch left there): lanes 0 .. FUSE_REPL of one wavefront.  PLAIN stores:
// nobody reads them before the next launch, and a wavefront's memory operations complete in order -- a write-through store takes a
// microsecond and more to be acknowledged, and whatever the wavefront waits for next waits for it too (measured at the kernel's
// start: the tile's first barrier 1.5 us later).  fz: Batch::fzc; the 32-bit side array lies behind the copies.
__device__ __forceinline__ void fuse_rearm(uint16_t* fz, uint32_t tile) {
    const uint32_t r = tidx() & 63u;
    if (r < FUSE_REPL) fz[r * FUSE_STRIDE + tile] = 0;
    else if (r == FUSE_REPL) reinterpret_cast<uint32_t*>(fz + FUSE_REPL * FUSE_STRIDE)[tile] = 0u;
}

// One word of four counts: complete (every count that matters known)?  Adds them to `s` if so.
__device__ __forceinline__ bool fuse_word(const Batch& b, unsigned long long q, uint32_t t0, uint32_t tile, uint32_t& s) {
    bool all = true;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v = (uint32_t)(q >> (16 * k)) & 0xFFFFu;
        if (t0 + k >= tile) v = 1u;                                    // (this tile itself and the ones behind it: nothing)
        else if (v == 0xFFFFu) v = ld_agent(b.ftb + t0 + k);          // (rare: 64 K tokens and more; 0 until the side word has arrived)
        all = all && v != 0u;
        acc += v - 1u;
    }
    s = acc;
    return all;
}

// Tokens in front of `tile`: ONE wavefront (all 64 lanes), returns the same value in every lane.
// A round reads every word that is still incomplete (six loads in flight per lane).  While MANY are, the wavefront then WATCHES one of
// them -- one lane, one request per nap -- and only reads the others again once that one is complete: it has to wait for all of them
// anyway, and the requests of a thousand waiting tiles re-reading everything they miss were what made a round take 3 us
// instead of 1 (requests past the L2 are served at a few per nanosecond for the whole GPU -- profiles/r06_one_launch.txt).
__device__ __forceinline__ unsigned long long fuse_base(const Batch& b, const uint32_t tile) {
    const uint32_t lane = tidx() & 63u;
    const unsigned long long* const w64 = reinterpret_cast<const unsigned long long*>(b.ftc + (tile % FUSE_REPL) * FUSE_STRIDE);
    uint32_t need = 0;
#pragma unroll
    for (int c = 0; c < FUSE_NCH; c++) if (4u * (64u * c + lane) < tile) need |= 1u << c;
    unsigned long long sum = 0;
    while (__any(need != 0u)) {
        unsigned long long q[FUSE_NCH];
#pragma unroll
        for (int c = 0; c < FUSE_NCH; c++) { q[c] = 0; if ((need >> c) & 1u) q[c] = ld_agent(w64 + 64u * c + lane); }
#pragma unroll
        for (int c = 0; c < FUSE_NCH; c++) {
            uint32_t s4;
            if (((need >> c) & 1u) && fuse_word(b, q[c], 4u * (64u * c + lane), tile, s4)) { sum += s4; need &= ~(1u << c); }
        }
        const unsigned long long pend = __ballot(need != 0u);
        if (!pend) break;
        __builtin_amdgcn_s_setprio(0);
        uint32_t m = 0;                                                        // incomplete words of the wavefront (uniform)
#pragma unroll
        for (int c = 0; c < FUSE_NCH; c++) m += (uint32_t)__popcll(__ballot((need >> c) & 1u));
        if (m <= (uint32_t)SPL_FUSE_FEW) { __builtin_amdgcn_s_sleep(SPL_FUSE_SLEEP); continue; }   // a few stragglers: all of them again
        // many: watch the first incomplete word (of the first lane that has one) until it is complete
        const uint32_t wl = (uint32_t)(__ffsll((long long)pend) - 1);
        const uint32_t wc = (uint32_t)__ffs((int)__builtin_amdgcn_readlane((int)need, (int)wl)) - 1u;
        const uint32_t widx = 64u * wc + wl;                                   // (uniform)
        for (;;) {
            __builtin_amdgcn_s_sleep(SPL_FUSE_SLEEP);
            uint32_t s4 = 0;
            bool done = false;
            if (lane == wl) {
                done = fuse_word(b, ld_agent(w64 + widx), 4u * widx, tile, s4);
                if (done) { sum += s4; need &= ~(1u << wc); }
            }
            if (__any(done)) break;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    return sum;
}

}  // namespace spl
