// spl_k_merge.h -- part of spl_kernels.hip (included there, in this order; one translation unit): LDS / global text accessors, wave-level minima and scans, the vocabulary probes of the tile kernel, the byte-pair merge loops (groups of 8 / 16 lanes, one wavefront, a workgroup; tabulated substring ids) and queue mode's k_deferred_wave.
#pragma once

namespace spl {

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
    if (id >= b.id_limit) return;                             // (the pseudo id of a single byte the vocabulary lacks: no token, bpe.rs:182-191)
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

}  // namespace spl
