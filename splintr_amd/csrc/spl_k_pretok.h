// spl_k_pretok.h -- part of spl_kernels.hip (included there, in this order; one translation unit): k_pretok: the tile kernel (stage, classify + masks, starts, probe, merge, tile record).
#pragma once

namespace spl {

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
constexpr uint32_t PRETOK_E_TSTART = 1u, PRETOK_E_SKIP = 2u, PRETOK_E_GAPS = 4u, PRETOK_E_EXT = 8u, PRETOK_E_FUSE = 64u;
inline uint32_t pretok_flags(const DeviceTables& T, const Batch& b) {
    return (b.tstart ? PRETOK_E_TSTART : 0u) | (b.skip ? PRETOK_E_SKIP : 0u) | (b.ext_gaps ? PRETOK_E_GAPS : 0u) |
           (b.ext_starts ? PRETOK_E_EXT : 0u) | (T.pattern << 4) | (b.ftc ? PRETOK_E_FUSE : 0u) | (b.tile0 << 8);        // (bits 8..31: the launch's first tile)
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
    const int KPAT = (int)((e_flags >> 4) & 3u);         // (the kernel specialised for one pattern: no faster, profiles/r03_launch_probes.txt)
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
    static_assert(sizeof(PretokTailLds) >= (size_t)SEG_ROWS * (SUB_W + 1) * 4, "the tail's slab must hold SEG_ROWS table rows and their sid[] words");
    uint32_t* const s_sid = s_u.t.slab[0] + SEG_ROWS * SUB_W;   // the tail's rows: byte | chunk slot << 8 | longest token that starts there << 24
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
#ifdef SPL_STAMP_FUSE     /* the fused mode's epilogue in place of the first two boundaries: [1] count published, [2] base known, [3] ids stored, [4] offsets stored (tools/dev/fuse_walls.py) */
#define SPL_STAMP_LO 7
#define SPL_FSTAMP(i) do { if (e_dbg && SPL_REC_BLK < SPL_DEBUG_BLOCKS / 2) e_dbg[16 + 8 * SPL_REC_BLK + (i)] = (unsigned long long)wall_clock64(); } while (0)
#define SPL_FSTAMP_T(t, i) do { if ((int)threadIdx.x == (t)) SPL_FSTAMP(i); } while (0)
#else
#define SPL_STAMP_LO 1
#endif
#define SPL_STAMP(i) do { if (e_dbg && threadIdx.x == 0 && SPL_REC_BLK < SPL_DEBUG_BLOCKS / 2 && (i) >= SPL_STAMP_LO && (i) <= 7 && (i) != 5) \
                              e_dbg[16 + 8 * SPL_REC_BLK + ((i) < 5 ? (i) : (i) - 1)] = (unsigned long long)wall_clock64(); } while (0)
#elif defined(SPL_DEBUG_STAMPS)
#define SPL_STAMP(i) do { if (e_dbg && blockIdx.x == SPL_DBG_WG && threadIdx.x == 0) e_dbg[i] = clock64(); \
                          if ((i) >= 1 && (i) <= 7 && b.stop_phase == (uint32_t)(i)) return; } while (0)
#else
#define SPL_STAMP(i) do { } while (0)
#endif
#ifndef SPL_FSTAMP
#define SPL_FSTAMP(i) do { } while (0)
#define SPL_FSTAMP_T(t, i) do { } while (0)
#endif

    const int tid = (int)threadIdx.x;                    // (wavefront indices rotated by the workgroup index, so that the phases of the low wavefronts
                                                         //  load different SIMDs in different workgroups: measured in round 2, no gain)
#define SPL_REC_BLK blockIdx.x
    if (DIRECT) __builtin_amdgcn_s_setprio(SPL_WORK_PRIO);
    // (fused mode: tile = workgroup index -- a tile waits for the tiles in front of it, which must have been dispatched before it)
#ifndef SPL_FUSE_XCD
#define SPL_FUSE_XCD 0           /* 1 (A/B only: not safe beside other launches): the XCD map in the fused mode too */
#endif
    const uint32_t tile_ix = ((!SPL_FUSE_XCD && (e_flags & PRETOK_E_FUSE)) ? blockIdx.x : xcd_tile()) + (e_flags >> 8);       // (+ the launch's first tile: launch_all's ranges)
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
    // (one lane per 64-byte line of the argument segment touching it with a vector load right behind the text loads: built, measured, no
    //  gain -- profiles/r04_flag_ab.txt)
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
    const uint64_t lim = (uint64_t)(w0 + (int64_t)(G::NBW + 1) * 32);
    auto more_documents = [&](uint32_t base) {         // documents base, base + 1, ... set their bits while they start inside the window
        for (;; base += NT) {
            const uint64_t d = (uint64_t)base + tid;
            uint64_t p = ~0ull;
            if (d < e_n_docs) p = e_doc_off[d];
            const bool in = p < lim && p < (uint64_t)B;
            if (in) { const uint32_t i = (uint32_t)(p - (uint64_t)w0); atomicOr(&s_ts[i >> 5], 1u << (i & 31)); }
            if (!__syncthreads_or(tid == NT - 1 && in)) break;
        }
    };
    uint32_t held_more = 0xFFFFFFFFu;                  // where the window's documents may go on behind the first round's loads
    if (DIRECT) {
        bool searched = false;                         // a barrier of the search has passed (uniform)
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
            searched = true;
            if (c == 0) hi = glo;                           // entry glo (if any) is not below the target
            else if (c == ghi - glo) lo = ghi;              // every probed entry is
            else { lo = hi = glo + c; p_held = p1; d_held = idx; d_held_end = ghi; }   // found: the entries behind it are already here
        }
        while (target != 0 && lo < hi) {
            const uint32_t span = hi - lo, st = (span + NT - 1) / NT;
            const uint64_t idx = (uint64_t)lo + (uint64_t)tid * st;
            const bool below = idx < hi && e_doc_off[idx] < target;
            const uint32_t c = (uint32_t)__syncthreads_count(below);
            searched = true;
            if (c == 0) { hi = lo; break; }
            const uint64_t nhi = (uint64_t)lo + (uint64_t)c * st;
            lo = lo + (c - 1) * st + 1;                // element lo + (c-1)*st is below the target
            hi = nhi < hi ? (uint32_t)nhi : hi;        // element lo + c*st (if any) is not
        }
        dw = lo;
        // (s_ts was zeroed before the search: any barrier of the search orders that in front of the bits set below)
        if (!searched) __syncthreads();
        if (d_held_end > dw) {                         // the window's documents from the first round's loads
            const bool in = d_held >= dw && d_held < d_held_end && p_held < lim && p_held < (uint64_t)B;
            if (in) { const uint32_t i = (uint32_t)(p_held - (uint64_t)w0); atomicOr(&s_ts[i >> 5], 1u << (i & 31)); }
            // more only if the last entry fetched is still inside the window: a flag that the phase's barrier publishes (a
            // barrier of its own here, and one in front of the bits, cost every tile 0.3 us for what a tile in a thousand needs)
            if (d_held == d_held_end - 1u && in) s_dq[10] = 1u;
            held_more = d_held_end;
        } else {
            more_documents(dw);
        }
    }
    SPL_STAMP(0);
    __syncthreads();
    if (held_more != 0xFFFFFFFFu && s_dq[10]) {        // (workgroup-uniform)
        more_documents(held_more);
        __syncthreads();
    }
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
        // (the bitmap is G::NBW <= 64 words: wavefront 0 holds them all -- one scan, no sums across wavefronts, one barrier)
        static_assert(G::NBW <= 64, "the window's bitmap words must fit one wavefront");
        if (tid < 64) {
            const uint32_t cnt = __popc(word);
            const uint32_t x = wave_scan_incl(cnt);
            uint32_t base = x - cnt;
            if (tid == 63) s_total = x;
            while (word) {
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                s_cpos[base++] = (uint16_t)(tid * 32 + bit);
            }
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
        if (!fast_starts) __syncthreads();             // (start masks: no chain ran, the barrier behind the enumeration is the phase's)
    }
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
                // the chunk memo (spl_k_memo.h): a chunk it holds is finished here -- its tokens go in place -- and never reaches the merge loops
                auto memo_put = [&](int q, uint32_t tid_) { s_ids[q] = tid_; atomicOr(&s_tbits[q >> 5], 1u << (q & 31)); };
                const bool m_short = T.memo && n <= SPL_MEMO_MAX_LEN && p + n <= Wv;
                const bool m_long = T.memo2 && n > SPL_MEMO_MAX_LEN && n <= SPL_MEMO_MAX_LEN2 && p + n <= Wv;       // (few: long identifiers, URLs)
                int held = m_short ? memo_probe<false>(T, tx, p, n, memo_put) : 0;
                if (m_long) held = memo_probe<true>(T, tx, p, n, memo_put);
#ifdef SPL_MEMO_STATS      /* profiling: chunks the vocabulary misses by what the memo said -- e_dbg[4..7]: not probed, not held, held, known as too long */
                if (e_dbg) atomicAdd(&e_dbg[4 + ((m_short || m_long) ? 1 + held : 0)], 1ull);
#endif
                if (held == 1) continue;
                const uint32_t item = (uint32_t)p | ((uint32_t)n << 16) | (held == 2 ? MISS_KNOWN : 0u);
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
                if (k < m16) { my_item[q] = s_miss[k]; my_r[q] = atomicAdd(&s_scnt[16 - MISS_N(my_item[q])], 1u); }
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
                    const uint32_t n = MISS_N(my_item[q]);
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
            if (id >= T.id_limit) return;                     // (the pseudo id of a single byte the vocabulary lacks: no token, bpe.rs:182-191)
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
#ifdef SPL_SKIP_MEDIUM          /* timing experiment only (tokens missing): what the 17..64-byte misses cost */
            continue;
#endif
#ifdef SPL_DEBUG_STAMPS
            ws_nmed++;
#endif
            if (DIRECT && SPL_MEDIUM_PRIO != SPL_MERGE_PRIO) __builtin_amdgcn_s_setprio(SPL_MEDIUM_PRIO);
            const uint32_t itemA = s_miss[G::C16 + it];
            const uint32_t itemB = (SPL_MEDIUM_PAIRS && it + 1u < m64) ? s_miss[G::C16 + it + 1u] : 0u;
            const int pA = (int)(itemA & 0xFFFFu), nA = (int)MISS_N(itemA), pB = (int)(itemB & 0xFFFFu), nB = (int)MISS_N(itemB);
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
#ifdef SPL_SKIP_SHORT           /* timing experiment only (tokens missing): what the misses of up to 16 bytes cost */
            continue;
#endif
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
            bpe_group16_tab(T, LdsAcc{s_rec, s_txt}, p, has ? (int)MISS_N(item) : 0, s_sub[tid >> 4],
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
    // chunk memo: what this tile had to merge goes into the handle's log (wavefront 3: in the fused mode it has nothing to do until the tile's
    // base is known; the lists are intact here -- the tail below overlays them)
    if (!TILE_LIST && b.mlog && (tid >> 6) == NT / 64 - 1 && (s_nq[0] | s_nq[1]) != 0u)
        memo_log(b, LdsAcc{s_rec, s_txt}, s_miss, s_nq[0], (uint32_t)G::C16, s_nq[1], tile_ix);
#ifdef SPL_DEBUG_STAMPS
    if (e_dbg) blk_w1 = blk_w2 = (unsigned long long)wall_clock64();
#endif
    {
        const int lane = tid & 63, wv = tid >> 6;
        const uint32_t ovf_lo = (uint32_t)(w0 + Wv);       // tokens from here on live in HBM (stage[] / tbits[])
        auto emit_g = [&](uint32_t q, uint32_t id) {
            if (id >= T.id_limit) return;                     // (as put: a byte the vocabulary lacks)
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
#ifndef SPL_TAIL_UNLIKELY
#define SPL_TAIL_UNLIKELY 1
#endif
        if (__builtin_expect(!SPL_SKIP_TAIL && (n_tm | s_dq[0] | s_dq[1] | s_dq[11]) != 0u, SPL_TAIL_UNLIKELY ? 0 : 1)) {         // workgroup-uniform
            // the single-byte ids in LDS for the tail (the ASCII kind table's 1 KB: dead since the classifier; the tail's first barrier orders it)
            static_assert(sizeof(s_aent) >= 256 * sizeof(uint32_t), "the byte-id table takes the kind table's place");
            uint32_t* const s_bid = reinterpret_cast<uint32_t*>(s_aent);
            { int t_b = tid; asm volatile("" : "+v"(t_b)); s_bid[t_b] = T.byte_id[t_b]; }
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
                        s_lq[2 * (have + tid) + 1] = MISS_N(item);
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
                    T, b, s_lq, nl0, s_u.t.slab[0], reinterpret_cast<uint32_t*>(s_cpos), s_sid, s_bid, s_txt, w0, w0 + iT, emit_g));
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
                            //  tools/gpu_custom_stress.py found its bytes tokenised, 46 of 2 883 batches)
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
                    int tid_c = tid;                          // (as tid_s above: what these loops derive from the thread index -- 64-bit addresses --
                    asm volatile("" : "+v"(tid_c));          //  is made here, not kept in registers, or in scratch, from in front of the tail)
                    for (int i = tid_c; i < DIRECT_WIN + 32; i += NT)
                        wtxt[i] = i < nst ? e_text[base + i + (i >= split ? (int64_t)removed : 0)] : (uint8_t)0;
                    __syncthreads();
                    for (int i = tid_c; i < nrec; i += NT) {
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
        const bool last_tile = t0 + TB_ >= B;               // (the tile that holds the corpus' end -- whatever range of the tiles this launch works)
        const uint64_t own_lo = (uint64_t)t0, own_hi = (uint64_t)(t0 + TB_);
        // (the documents are wavefront 2's: wavefront 0 counts and publishes, wavefront 1 reads the other tiles' counts in the fused mode -- a
        //  wavefront's memory operations complete in order, so whoever has just stored must not be the one that waits for a load)
        const bool docs_wave = (tid_late >> 6) == 2;
        const int dl = tid_late & 63;
        const uint64_t d_first = (uint64_t)dw + (uint32_t)dl;
        uint64_t p_first = ~0ull;
        if (docs_wave && d_first <= e_n_docs) p_first = e_doc_off[d_first];      // entry n_docs is the end of the corpus
        // what the result stores need of the argument segment: scalar loads of lines no earlier phase touched.  Wavefronts 2 and 3 -- idle
        // here -- wait for them now (0.3-0.7 us), so that they are in the scalar cache when the others ask
        uint32_t* o_ids = b.ids_out; uint64_t o_cap = b.ids_cap; uint64_t* o_off = b.off_out; uint64_t* o_off2 = b.off_out2;
        uint32_t* o_slab = b.slab; uint32_t* o_done = b.done;
        if ((tid_late >> 6) >= 2) asm volatile("" : "+s"(o_ids), "+s"(o_cap), "+s"(o_off), "+s"(o_off2), "+s"(o_slab), "+s"(o_done));
        // fused mode (spl_k_fuse.h): ONE launch -- the tile learns its base from the counts of the tiles in front of it and writes its part
        // of the CSR itself.  Wavefront 1 reads those counts NOW, beside wavefront 0's count of this tile's own tokens.
        const bool fuse = (e_flags & PRETOK_E_FUSE) != 0u;                    // (uniform)
#ifndef SPL_FUSE_EARLY
#define SPL_FUSE_EARLY 0         /* 1 (A/B): wavefront 1 reads the counts BESIDE this tile's own count instead of behind it: 28.85 against 27.93 us per step on the bench batch -- polls that cannot succeed yet only add to the traffic past the L2 */
#endif
        if (SPL_FUSE_EARLY && fuse && (tid_late >> 6) == 1) {
            SPL_FSTAMP_T(64, 4);
            const unsigned long long bb = fuse_base(b, tile_ix);
            if (tid_late == 64) { s_red[0] = bb; SPL_FSTAMP(2); }
            __builtin_amdgcn_s_setprio(SPL_WORK_PRIO);
        }
        // ---- token count of the tile: window bitmap + overflow range --------------------------------
        // (the bitmap is 34 words: ONE wavefront counts, ranks and lists them -- no sums across wavefronts, one barrier)
        static_assert(G::NBW + 2 <= 64, "the window's bitmap words must fit one wavefront");
        uint32_t c_win;
        {
            if (tid_late < 64) {
                uint32_t word = tid_late < G::NBW + 1 ? s_tbits[tid_late] : 0u;
                const uint32_t cnt = __popc(word);
                const uint32_t x = wave_scan_incl(cnt);
                uint32_t basew = x - cnt;
                if (tid_late == 63) s_total = x;
                // fused mode: nothing of this tile lies beyond its window (nearly always): x IS its token count -- published now, for the
                // tiles behind this one, before the positions are listed
                if (e_flags & PRETOK_E_FUSE) {
                    if (s_dq[4] == 0u && tile_ix + 1u < gridDim.x) fuse_publish(b, tile_ix, (uint32_t)__builtin_amdgcn_readlane((int)x, 63));   // (the last tile: nobody reads its count)
                    fuse_rearm(b.fzc, tile_ix);
                    if (tid_late == 0) SPL_FSTAMP(1);
                }
                if (tid_late < G::NBW + 2) s_wpre[tid_late] = basew;
                while (word) {                                  // token positions in order
                    const int bit = __ffs(word) - 1;
                    word &= word - 1;
                    s_cpos[basew++] = (uint16_t)(tid_late * 32 + bit);
                }
            }
            lds_barrier();                                      // (the count, the positions -- and in the fused mode the base)
            c_win = s_total;
            if (tid_late == 64) SPL_FSTAMP(5);
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
        if (fuse) { if (tid_late < 64 && ovf_hi != 0u && tile_ix + 1u < gridDim.x) fuse_publish(b, tile_ix, (uint32_t)total); }
        else if (tid_late == 0 && !queue_mode) atomicAdd(&b.tctl[16 + b.tpar * b.tgroups + (tile_ix >> 6)], (uint32_t)total);
        // the first 64 documents of the window (one per lane of wavefront 0) are ranked NOW: their offsets were fetched before the count,
        // and a wait for that load placed behind the id stores below would wait for the stores as well (one counter, in order: 1-5 us
        // under load -- profiles/r06_one_launch.txt)
        bool in0 = false, own0 = false;
        uint64_t v0 = 0;
        if (docs_wave) {
            in0 = d_first <= e_n_docs && (p_first < own_hi || last_tile);
            own0 = in0 && p_first >= own_lo;
            if (own0 && !queue_mode) {
                const uint32_t i = (uint32_t)(p_first - (uint64_t)w0);
                v0 = (uint64_t)(s_wpre[i >> 5] + __popc(s_tbits[i >> 5] & ((1u << (i & 31)) - 1u))) + ((last_tile && p_first >= (uint64_t)B) ? c_ovf : 0u);
            }
        }
        if (!SPL_FUSE_EARLY && fuse) {
            if ((tid_late >> 6) == 1) { const unsigned long long bb = fuse_base(b, tile_ix); if (tid_late == 64) s_red[0] = bb; __builtin_amdgcn_s_setprio(SPL_WORK_PRIO); }
            lds_barrier();
        }
        const unsigned long long fbase = fuse ? s_red[0] : 0ull;
        if (!fuse) o_slab = nullptr;
        if (queue_mode && tid_late < TILE_BITS_W)
            b.tile_bits[(size_t)tile_ix * TILE_BITS_W + tid_late] = tid_late < G::NBW + 1 ? s_tbits[tid_late] : 0u;
#ifdef SPL_DEBUG_STAMPS
        if (e_dbg) blk_w2 = (unsigned long long)wall_clock64();
#endif
        const uint32_t slot = tile_ix * b.tslot;                // fixed slots: nothing to wait for
        const uint32_t s_ids_at = 3 + b.slab_max_docs, s_ids_cap = (fuse && o_slab) ? slab_id_cap(b.slab_cap - s_ids_at, b.slab_p24 != 0u) : 0u;
        auto put_final = [&](unsigned long long r, uint32_t id) {
            if (r < o_cap) o_ids[r] = id;
            if (r < s_ids_cap) slab_put_id(o_slab + s_ids_at, (uint32_t)r, id, b.slab_p24 != 0u);
        };
        if (fuse) { for (uint32_t k = tid_late; k < c_win; k += NT) put_final(fbase + k, s_ids[s_cpos[k]]); }
        else for (uint32_t k = tid_late; k < c_win; k += NT) b.tile_ids[slot + k] = s_ids[s_cpos[k]];
        // the documents that start in the tile's own range: their output offsets, first index and count -- by ONE wavefront,
        // 64 documents per round (a tile of ordinary text holds a handful; no barrier, the other wavefronts are done)
        if (docs_wave) {
            uint32_t d_lo = 0xFFFFFFFFu, d_n = 0;
            for (uint32_t db = dw;; db += 64u) {
                const uint64_t d = (uint64_t)db + dl;
                bool in = in0, own = own0;
                uint64_t v_off = v0;
                if (db != dw) {                                   // (more than 64 documents start in the window: rare)
                    uint64_t p = ~0ull;
                    if (d <= e_n_docs) p = e_doc_off[d];
                    in = d <= e_n_docs && (p < own_hi || last_tile);
                    own = in && p >= own_lo;
                    if (own && !queue_mode) {
                        const uint32_t i = (uint32_t)(p - (uint64_t)w0);
                        v_off = (uint64_t)(s_wpre[i >> 5] + __popc(s_tbits[i >> 5] & ((1u << (i & 31)) - 1u))) + ((last_tile && p >= (uint64_t)B) ? c_ovf : 0u);
                    }
                }
                if (own && !queue_mode) {
                    const uint64_t v_fin = v_off + fbase;            // (fused: the tile's base is known: these ARE the final offsets)
                    o_off[d] = v_fin;
                    if (fuse && o_off2) o_off2[d] = v_fin;
                    if (fuse && o_slab && d <= b.slab_max_docs) o_slab[2 + d] = (uint32_t)v_fin;
                }
                const unsigned long long mo = __ballot(own);    // owned documents are consecutive
                if (mo) {
                    if (d_lo == 0xFFFFFFFFu) d_lo = db + (uint32_t)(__ffsll((long long)mo) - 1);
                    d_n += (uint32_t)__popcll(mo);
                }
                if (!((__ballot(in) >> 63) & 1ull)) break;      // more documents beyond these 64?
            }
            if (fuse && whi > wlo) {                             // rare: the tokens that start beyond the window, behind the window's (as k_tile_out)
                unsigned long long running = fbase + c_win;
                for (uint32_t wb = wlo; wb < whi; wb += 64u) {
                    const uint32_t w = wb + (uint32_t)dl;
                    uint32_t word = w < whi ? ld_agent(b.tbits + w) : 0u;
                    if (w == whi - 1u && (ovf_hi & 31u)) word &= (1u << (ovf_hi & 31u)) - 1u;
                    const uint32_t cnt = __popc(word);
                    const uint32_t x = wave_scan_incl(cnt);
                    unsigned long long r = running + (x - cnt);
                    if (word && !b.skip) st_agent(b.tbits + w, 0u);          // clean after use: the bitmap is all-zero between calls
                    while (word) {
                        const int bit = __ffs(word) - 1;
                        word &= word - 1;
                        put_final(r, ld_agent(b.stage + (size_t)w * 32u + (uint32_t)bit));
                        r++;
                    }
                    running += (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                }
            }
            if (fuse) {
                if (o_slab && last_tile && dl == 0) { o_slab[0] = (uint32_t)(fbase + total); o_slab[1] = e_n_docs; }   // header: T, N
            } else if (dl == 0) {
                TileDesc td;
                td.slot = slot; td.c_win = c_win; td.c_ovf = c_ovf; td.ovf_hi = whi > wlo ? ovf_hi : 0u;
                td.d_first = d_lo == 0xFFFFFFFFu ? 0u : d_lo; td.d_cnt = d_n; td.ovf_lo = ovf_lo;
                td.c_own = s_wpre[DIRECT ? (LH + TB_) >> 5 : 0];             // tile range ends on a word boundary
                b.tdesc[tile_ix] = td;
            }
        }
        if (fuse) {
            if (tile_ix == 0 && b.fz_n > gridDim.x) {            // the previous fused launch had more tiles than this one: their entries of the other parity
                for (uint32_t k = gridDim.x + (uint32_t)tid_late; k < b.fz_n; k += NT) {
                    for (uint32_t r = 0; r < FUSE_REPL; r++) st_agent(b.fzc + r * FUSE_STRIDE + k, (uint16_t)0u);
                    st_agent(b.fzb + k, 0u);
                }
            }
            if (o_done) {                                        // latency path: the last tile to get here stores the word the host spins on
                __threadfence_system();
                __syncthreads();
                if (tid_late == 0 && (gridDim.x == 1u || atomicAdd(&b.tctl[8], 1u) == gridDim.x - 1u)) {
                    if (gridDim.x != 1u) { b.tctl[8] = 0u; __threadfence_system(); }
                    __hip_atomic_store(o_done, b.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
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
#ifdef SPL_DEBUG_STAMPS
        if (e_dbg && tid_end == 0) atomicMax(&e_dbg[15], (unsigned long long)wall_clock64());
#else
        // (profiling: every workgroup's end in a word of its own, the host takes the maximum -- as ONE word updated by atomicMax the
        //  1248 workgroups of the fused mode, which all end at the same moment, queued up behind each other for 7 us)
        if (e_dbg && tid_end == 0) e_dbg[16 + (blockIdx.x & (4u * SPL_DEBUG_BLOCKS - 1u))] = (unsigned long long)wall_clock64();
#endif
    }
#undef SPL_REC_BLK
#undef SPL_STAMP
#undef SPL_FSTAMP
#undef SPL_FSTAMP_T
}

}  // namespace spl
