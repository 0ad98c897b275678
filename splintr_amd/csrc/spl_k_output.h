// spl_k_output.h -- part of spl_kernels.hip (included there, in this order; one translation unit): k_tile_out (tile records -> CSR) and queue mode's k_range_count / k_range_out / k_bpe_segments / k_bpe_long.
#pragma once

namespace spl {

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
    uint32_t tpar, tgroups, tslot, slab_cap, slab_max_docs, n_docs, slab_p24;
    // latency path (small host batches): the LAST workgroup to finish stores done_seq to *done (pinned host memory, system scope) behind
    // everybody's result stores -- the host spins on that word instead of synchronising the stream (tctl[8]: workgroups finished)
    uint32_t* done; uint32_t done_seq;
    uint32_t tile0;           // the launch's first tile (launch_all's ranges of a large batch; 0 otherwise)
    const unsigned long long* gpre;     // null, or the exclusive prefix sums of the groups' counts (k_group_scan: batches of many tiles)
};
inline TileOutArgs tile_out_args(const Batch& b) {
    return TileOutArgs{b.tctl, b.tdesc, b.tile_ids, b.ids_out, b.ids_cap, b.off_out, b.off_out2, b.slab, b.tbits, b.stage, b.skip,
                       b.tpar, b.tgroups, b.tslot, b.slab_cap, b.slab_max_docs, b.n_docs, b.slab_p24, b.done, b.done_seq, b.tile0, b.gpre};
}
// Exclusive prefix sums of the groups' token counts, between k_pretok and k_tile_out of a batch of many tiles: every workgroup of k_tile_out
// adds up the counts of all groups in front of its tile's -- 4 200 of them at 215 MB, 243 000 workgroups: 1.9 GB of reads for what one
// workgroup does once (k_tile_out: 550 -> us of the 5.4 ms step).
__global__ __launch_bounds__(256) void k_group_scan(const uint32_t* gs, uint32_t n, unsigned long long* gpre) {
    __shared__ unsigned long long s_w[4];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t per = (n + 255u) / 256u, lo = (uint32_t)tid * per, hi = lo + per < n ? lo + per : n;
    unsigned long long sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += gs[k];
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    unsigned long long run = incl - sum;
    for (int w = 0; w < wv; w++) run += s_w[w];
    for (uint32_t k = lo; k < hi; k++) { gpre[k] = run; run += gs[k]; }
}

__global__ __launch_bounds__(TOUT_NT) void k_tile_out(TileOutArgs b) {
    __shared__ unsigned long long s_part[TOUT_NT / 64];
    __shared__ uint32_t s_wsum[TOUT_NT / 64];
    const uint32_t t = xcd_tile() + b.tile0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t* gs = b.tctl + 16 + b.tpar * b.tgroups;
    const uint32_t g = t >> 6;
    // the tile's slot is fixed, so its first TOUT_NT tokens are fetched before the counts are known
    // (most tiles hold fewer): the copy below then depends on ONE round of loads, not two
    const uint32_t slot0 = t * b.tslot;
    const uint32_t first_id = b.tile_ids[slot0 + tid];
    unsigned long long mine = 0;
    // (a batch of many tiles: the sums of the groups in front are ONE load -- k_group_scan has added them up; else every workgroup adds them itself)
    if (b.gpre) { if (tid == 0) mine = b.gpre[g]; }
    else for (uint32_t k = tid; k < g; k += TOUT_NT) mine += gs[k];
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
    const uint32_t s_ids_at = 3 + b.slab_max_docs, s_ids_cap = b.slab ? slab_id_cap(b.slab_cap - s_ids_at, b.slab_p24 != 0u) : 0u;
    for (uint32_t k = tid; k < td.c_win; k += TOUT_NT) {
        const unsigned long long r = base + k;
        const uint32_t id = k < (uint32_t)TOUT_NT ? first_id : b.tile_ids[td.slot + k];
        if (r < b.ids_cap) b.ids_out[r] = id;
        if (r < s_ids_cap) slab_put_id(b.slab + s_ids_at, (uint32_t)r, id, b.slab_p24 != 0u);
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
                if (r < s_ids_cap) slab_put_id(b.slab + s_ids_at, (uint32_t)r, b.stage[w * 32 + bit], b.slab_p24 != 0u);
                r++;
            }
            running += all;
        }
    }
    if (t == 0) {
        uint32_t* other = b.tctl + 16 + (b.tpar ^ 1u) * b.tgroups;
        for (uint32_t k = tid; k < b.tgroups; k += TOUT_NT) other[k] = 0u;
    }
    if (b.done) {                                            // (uniform: a kernel argument)
        __threadfence_system();                              // this workgroup's result stores, visible system-wide ...
        __syncthreads();
        if (tid == 0 && atomicAdd(&b.tctl[8], 1u) == gridDim.x - 1u) {        // ... before it counts as finished; the last one:
            b.tctl[8] = 0u;                                  // (re-armed for the next call on this context)
            __threadfence_system();
            __hip_atomic_store(b.done, b.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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
    __shared__ uint32_t s_sid[SEG_ROWS];
    __shared__ uint32_t s_bid[256];
    __shared__ uint32_t s_first;
    const int tid = threadIdx.x;
    s_bid[tid] = T.byte_id[tid];                                   // (NT == 256; the first barrier below orders it)
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
        const uint32_t nl2 = bpe_tail_segments<2>(T, b, s_lq, cnt, s_slab, s_scr, s_sid, s_bid, nullptr, 0, 0,
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

}  // namespace spl
