// spl_k_memo.h -- part of spl_kernels.hip (included there, in this order; one translation unit): the chunk memo -- its probe in the tile kernel, the log of the chunks it did not hold, and k_memo_fill, which encodes those between two launches and puts them in.
#pragma once

namespace spl {

// The reference keeps an LRU of encoded chunks in front of byte_pair_encode (src/core/tokenizer.rs:707-722: hash of the chunk -> its tokens).
// Here: a table in HBM (MemoEnt, spl_common.h), one 64-byte entry per chunk of up to 32 bytes and six tokens (fourteen with the slot's second line), the key
// compared in full; a second, smaller table of the same entries for chunks of 33..64 bytes (their keys' second half in a parallel array: MemoHi).  The tile kernel only READS it, right behind the whole-chunk probe of a chunk the vocabulary does not hold as one
// token: a hit puts the chunk's tokens in place, and the chunk never reaches the merge loops -- which are 40 % of the tile kernel's vector
// instructions on English / code and more on lexically wide text (profiles/r05_merge_bounds.txt).  A chunk the memo does not hold is
// merged as before and LOGGED (its bytes, 48 per chunk, one returning atomic per tile); k_memo_fill encodes the logged chunks -- one lane
// per chunk, byte_pair_encode by the pair table (bpe_serial, spl_lookup.h) -- and
// writes their entries between two launches on the handle's stream: kernel boundaries give the coherence that an insert from inside the
// tile kernel lacked (round 3: eight L2s, plain stores, torn lines).  Result-transparent: a hit returns exactly what the merge returns.

// meta0 of a valid entry for a chunk of n bytes with t tokens
__device__ __forceinline__ uint32_t memo_meta0(uint32_t n, uint32_t t) { return 0x80000000u | (t << 8) | n; }

// A miss-list item (window position | length << 16) of a chunk the memo KNOWS and cannot hold: merged as ever, not logged again
constexpr uint32_t MISS_KNOWN = 0x80000000u;
#define MISS_N(item) (((item) >> 16) & 0x7FFFu)

// The probe: text[p, p + n) lies inside the window; LONG: 33 <= n <= 64 (the second table), else 2 <= n <= 32.  emit(position, id) for every
// token of a hit.  0: not held; 1: held, its tokens are emitted; 2: held as "more than fourteen tokens" -- the chunk goes the usual way, and is not logged.
template <bool LONG, class TX, class Emit>
__device__ __forceinline__ int memo_probe(const DeviceTables& T, const TX& tx, int p, int n, Emit emit) {
    constexpr int KW = LONG ? 16 : 8;
    uint32_t k[KW];
#pragma unroll
    for (int i = 0; i < KW; i++) k[i] = 4 * i < n ? mask_tail(tx.load32(p + 4 * i), n - 4 * i) : 0u;
    const uint32_t h = hash_memo_w<KW>(k, (uint32_t)n);
    const MemoEnt* const tab = LONG ? T.memo2 : T.memo;
    const MemoExt* const ext = LONG ? T.memo2_ext : T.memo_ext;
    const uint32_t mask = LONG ? T.memo2_mask : T.memo_mask;
    const uint32_t nf = (uint32_t)(LONG ? n - SPL_MEMO_MAX_LEN : n);          // meta0's length field
    uint32_t slot = h & mask;
    // two candidate slots (k_memo_fill puts a chunk into the second one only where the first is taken by another chunk, and a slot never becomes
    // empty again): the second load goes out only where the first slot holds something else
#pragma nounroll
    for (int way = 0; way < 2; way++) {
        const Quad* q = reinterpret_cast<const Quad*>(tab + slot);
        const Quad a = q[0], c = q[1], m = q[2], d = q[3];        // key[0..3], key[4..7], meta0 meta1 ids[0..1], ids[2..5]
        bool eq = (a.x == k[0]) & (a.y == k[1]) & (a.z == k[2]) & (a.w == k[3]) & (c.x == k[4]) & (c.y == k[5]) & (c.z == k[6]) & (c.w == k[7]) &
                  ((m.x & 0x8000003Fu) == (0x80000000u | nf));
        if (LONG && eq) {                                         // bytes 32..63 of the key
            const Quad* hq = reinterpret_cast<const Quad*>(T.memo2_hi + slot);
            const Quad h0 = hq[0], h1 = hq[1];
            eq = (h0.x == k[KW - 8]) & (h0.y == k[KW - 7]) & (h0.z == k[KW - 6]) & (h0.w == k[KW - 5]) &
                 (h1.x == k[KW - 4]) & (h1.y == k[KW - 3]) & (h1.z == k[KW - 2]) & (h1.w == k[KW - 1]);
        }
        if (eq) {
            const uint32_t nt = (m.x >> 8) & 15u;
            if (nt == 0u) return 2;
            const uint32_t ids[6] = {m.z, m.w, d.x, d.y, d.z, d.w};
            int at = 0;
#pragma unroll
            for (int t = 0; t < SPL_MEMO_TOK1; t++)
                if ((uint32_t)t < nt) { emit(p + at, ids[t]); if (t < 5) at = (int)((m.y >> (6 * t)) & 63u); }
            if (nt > (uint32_t)SPL_MEMO_TOK1) {                   // seven to fourteen tokens: the rest from the slot's second line (rare)
                const Quad* x = reinterpret_cast<const Quad*>(ext + slot);
                const Quad e0 = x[0], e1 = x[1], e2 = x[2];       // ids[0..3], ids[4..7], ends[0..1]
                const uint32_t xi[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
                for (int t = 0; t < 8; t++)
                    if ((uint32_t)(t + SPL_MEMO_TOK1) < nt) {
                        at = (int)(((t < 5 ? e2.x : e2.y) >> (6 * (t % 5))) & 63u);      // the end of token t + 5 = the start of token t + 6
                        emit(p + at, xi[t]);
                    }
            }
            return 1;
        }
        if (!(m.x >> 31)) return 0;                               // an empty first slot: the chunk is in neither
        slot = memo_slot2(h, mask);
    }
    return 0;
}

// The tile's misses into the handle's log, by ONE wavefront (all 64 lanes): list = s_miss, the misses of up to 16 bytes at [0, m16), those
// of 17..64 at [c16, c16 + m64); item = window position | length << 16.  LONG: the misses of 33..64 bytes, into the second log (regions
// SPL_MEMO_LOG_REGIONS .. of the counters); else those of up to 32.
template <bool LONG, class TX>
__device__ __forceinline__ void memo_log_part(const Batch& b, const TX& tx, const uint32_t* list, uint32_t m16, uint32_t c16, uint32_t m64, uint32_t tile) {
    constexpr int KW = LONG ? 16 : 8;
    constexpr uint32_t LW = LONG ? SPL_MEMO_LOG_WORDS2 : SPL_MEMO_LOG_WORDS;
    const uint32_t lane = tidx() & 63u;
    // the eligible items, densely -- but for what the memo knows it cannot hold
    auto eligible = [&](uint32_t item) {
        const uint32_t n = MISS_N(item);
        return !(item & MISS_KNOWN) && (LONG ? n > (uint32_t)SPL_MEMO_MAX_LEN && n <= (uint32_t)SPL_MEMO_MAX_LEN2 : n <= (uint32_t)SPL_MEMO_MAX_LEN);
    };
    uint32_t n_el = 0;
    if (!LONG) for (uint32_t i0 = 0; i0 < m16; i0 += 64u) n_el += (uint32_t)__popcll(__ballot(i0 + lane < m16 && eligible(list[i0 + lane])));
    for (uint32_t i0 = 0; i0 < m64; i0 += 64u) n_el += (uint32_t)__popcll(__ballot(i0 + lane < m64 && eligible(list[c16 + i0 + lane])));
    if (!n_el) return;
    const uint32_t cap = LONG ? b.mlog2_cap : b.mlog_cap;
    const uint32_t region = tile % SPL_MEMO_LOG_REGIONS;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(&b.mlog_cnt[(LONG ? SPL_MEMO_LOG_REGIONS : 0) + region], n_el);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= cap) return;                                      // the region is full: the fill has not run yet
    // the region becomes half full (512 entries at most): the host enqueues a fill with one of its next launches (a word in pinned host memory).  Not before: a few
    // chunks per launch that lose their slot to another again and again would otherwise cost a fill every few launches
    // (a large log -- a context that takes large batches -- asks at 512 entries all the same: what a first fill left over trickles in slowly)
    const uint32_t ask = cap / 2u < 512u ? cap / 2u : 512u;
    if (lane == 0u && base < ask && base + n_el >= ask) *b.mflag = 1u;
    uint32_t* const out = (LONG ? b.mlog2 : b.mlog) + ((size_t)region * cap) * LW;
    auto put = [&](uint32_t slot, uint32_t item) {
        if (base + slot >= cap) return;
        const int p = (int)(item & 0xFFFFu), n = (int)MISS_N(item);
        uint32_t* e = out + (size_t)(base + slot) * LW;
        e[0] = (uint32_t)n;
#pragma unroll
        for (int i = 0; i < KW; i++) e[1 + i] = 4 * i < n ? mask_tail(tx.load32(p + 4 * i), n - 4 * i) : 0u;
    };
    uint32_t slot = 0;
    for (int part = LONG ? 1 : 0; part < 2; part++) {
        const uint32_t cnt = part ? m64 : m16, at = part ? c16 : 0u;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const uint32_t item = i < cnt ? list[at + i] : MISS_KNOWN;
            const bool el = i < cnt && eligible(item);
            const unsigned long long em = __ballot(el);
            if (el) put(slot + mbcnt64(em), item);
            slot += (uint32_t)__popcll(em);
        }
    }
}
template <class TX>
__device__ __forceinline__ void memo_log(const Batch& b, const TX& tx, const uint32_t* list, uint32_t m16, uint32_t c16, uint32_t m64, uint32_t tile) {
    memo_log_part<false>(b, tx, list, m16, c16, m64, tile);
    if (b.mlog2 && m64) memo_log_part<true>(b, tx, list, m16, c16, m64, tile);
}

// Between two launches: one lane per logged chunk.  claim[slot] == round: another lane of this launch has the slot.
// LONG: the second table (chunks of 33..64 bytes, 64 threads a workgroup: bpe_serial's nodes of a lane are 64 x 2 words of LDS).
constexpr int MEMO_FILL_NT = 128, MEMO_FILL_NT2 = 64;
template <int NT_> struct MemoNodes {   // bpe_serial's per-node storage for one lane: ids and ranks of its nodes, in LDS (node i of lane t at [i][t])
    uint32_t* id_; uint32_t* rk_;
    __device__ __forceinline__ uint32_t& id(int i) { return id_[i * NT_]; }
    __device__ __forceinline__ uint32_t& rk(int i) { return rk_[i * NT_]; }
};
struct MemoKeyAcc {           // the logged bytes as bpe_serial's text
    const uint32_t* k;
    __device__ __forceinline__ uint32_t txt(int q) const { return (k[q >> 2] >> (8 * (q & 3))) & 0xFFu; }
};
template <bool LONG>
__global__ __launch_bounds__(LONG ? MEMO_FILL_NT2 : MEMO_FILL_NT) void k_memo_fill(DeviceTables T, MemoEnt* memo, MemoExt* memo_ext, MemoHi* memo_hi, const uint32_t* mlog,
                                                                               const uint32_t* mlog_cnt, uint32_t mlog_cap, uint32_t* claim, uint32_t round, unsigned long long* stats) {
    constexpr int KW = LONG ? 16 : 8, NTF = LONG ? MEMO_FILL_NT2 : MEMO_FILL_NT, NMAX = LONG ? SPL_MEMO_MAX_LEN2 : SPL_MEMO_MAX_LEN;
    constexpr uint32_t LW = LONG ? SPL_MEMO_LOG_WORDS2 : SPL_MEMO_LOG_WORDS;
    const uint32_t region = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cnt = mlog_cnt[region] < mlog_cap ? mlog_cnt[region] : mlog_cap;
    if (i >= cnt) return;
    const uint32_t* e = mlog + ((size_t)region * mlog_cap + i) * LW;
    const uint32_t n = e[0];
    uint32_t k[KW];
#pragma unroll
    for (int w = 0; w < KW; w++) k[w] = e[1 + w];
    if (n < (LONG ? (uint32_t)SPL_MEMO_MAX_LEN + 1u : 2u) || n > (uint32_t)NMAX) return;
    const uint32_t mask = LONG ? T.memo2_mask : T.memo_mask;
    const uint32_t nf = LONG ? n - (uint32_t)SPL_MEMO_MAX_LEN : n;
    const uint32_t h = hash_memo_w<KW>(k, n);
    const uint32_t slots[2] = {h & mask, memo_slot2(h, mask)};
    auto holds = [&](uint32_t sl) {                               // the chunk itself (logged by many tiles of a cold batch, or put in by an earlier launch)?
        const MemoEnt* me = memo + sl;
        bool same = (me->meta0 & 0x8000003Fu) == (0x80000000u | nf);
#pragma unroll
        for (int w = 0; w < 8; w++) same = same && me->key[w] == k[w];
        if (LONG) {
#pragma unroll
            for (int w = 0; w < 8; w++) same = same && memo_hi[sl].k[w] == k[KW - 8 + w];
        }
        return same;
    };
    const bool taken0 = (memo[slots[0]].meta0 >> 31) != 0u, taken1 = (memo[slots[1]].meta0 >> 31) != 0u;
    if ((taken0 && holds(slots[0])) || (taken0 && taken1 && holds(slots[1]))) return;
    // the first slot if it is free, else the second if it is free, else the first slot's chunk gives way.  One writer per slot and launch
    // (claim[slot] == round: taken in this launch); a lane that loses both tries is logged again by a later launch.
    MemoEnt* me = nullptr;
    if (!taken0) { if (atomicExch(&claim[slots[0]], round) != round) me = memo + slots[0]; }
    if (!me && taken0 && !taken1) { if (atomicExch(&claim[slots[1]], round) != round) me = memo + slots[1]; }
    if (!me && taken0 && taken1) { if (atomicExch(&claim[slots[0]], round) != round) me = memo + slots[0]; }
    if (!me) return;
    __shared__ uint32_t s_id[NMAX * NTF], s_rk[NMAX * NTF];
    MemoNodes<NTF> s{s_id + threadIdx.x, s_rk + threadIdx.x};
    const MemoKeyAcc tx{k};
    bpe_serial(T, s, tx, 0, (int)n);
    uint32_t ids[SPL_MEMO_MAX_TOK], ends = 0, ends2[2] = {0u, 0u}, nt = 0;
#pragma unroll
    for (int t = 0; t < SPL_MEMO_MAX_TOK; t++) ids[t] = 0u;
    bool fits = true;
    for (uint32_t q = 0; q < n; q++) {
        const uint32_t id = s.id((int)q);
        if (id == SPL_DEAD) continue;
        // token nt - 1 ends where this one starts: ends of tokens 0..4 in the entry, of tokens 5..12 in the second line
        if (nt >= 1u && nt <= 5u) ends |= q << (6u * (nt - 1u));
        else if (nt >= 6u && nt <= 13u) ends2[(nt - 6u) / 5u] |= q << (6u * ((nt - 6u) % 5u));
        if (id >= T.id_limit || nt >= (uint32_t)SPL_MEMO_MAX_TOK) { fits = false; break; }   // (a byte the vocabulary lacks -- a pseudo id: never memoized)
        ids[nt++] = id;
    }
    me->meta0 = 0u;                                               // (invalid while it is rewritten: nobody reads it before the launch ends)
#pragma unroll
    for (int w = 0; w < 8; w++) me->key[w] = k[w];
    if (LONG) {
        MemoHi* const mh = memo_hi + (me - memo);
#pragma unroll
        for (int w = 0; w < 8; w++) mh->k[w] = k[KW - 8 + w];
    }
    me->meta1 = ends;
#pragma unroll
    for (int t = 0; t < SPL_MEMO_TOK1; t++) me->ids[t] = ids[t];
    if (fits && nt > (uint32_t)SPL_MEMO_TOK1) {
        MemoExt* const mx = memo_ext + (me - memo);
#pragma unroll
        for (int t = 0; t < 8; t++) mx->ids[t] = ids[SPL_MEMO_TOK1 + t];
        mx->ends[0] = ends2[0]; mx->ends[1] = ends2[1];
    }
    me->meta0 = 0x80000000u | ((fits ? nt : 0u) << 8) | nf;
    if (stats) atomicAdd(&stats[fits ? 0 : 1], 1ull);
}

}  // namespace spl
