// spl_lookup.h -- vocabulary probes and the per-chunk merge loop.
//
// probe_chunk()  : the whole-chunk fast path of encode_chunk_with_position
//                  (reference src/core/tokenizer.rs:703-705) and byte_pair_encode's own
//                  whole-piece check (src/core/bpe.rs:78-80).
// bpe_serial()   : byte_pair_encode's merge loop (src/core/bpe.rs:83-197) for ONE chunk worked by
//                  ONE lane: node i lives at byte offset i, dead nodes are tomb-stoned instead
//                  of unlinked, "rank of the pair (i, next)" comes from the (left id, right id)
//                  pair table instead of re-hashing the concatenated byte slice (exact because
//                  every node's bytes are a token -- DESIGN.md "Pair table equivalence").
// Templates over accessors so the same code runs in the kernels and in tests/hostsim.
#pragma once
#include "spl_common.h"

namespace spl {

constexpr uint32_t SPL_DEAD = 0xFFFFFFFEu;

// TX: uint32_t load32(int p) -> bytes p..p+3 little endian (bytes past the text read as anything;
// callers mask).  Key words beyond n are zero.
SPL_HD uint32_t mask_tail(uint32_t w, int nbytes) {     // keep the low nbytes (0..4) bytes
    return nbytes >= 4 ? w : nbytes <= 0 ? 0u : (w & ((1u << (8 * nbytes)) - 1u));
}

// NOTE: the bucket compares (short table, pair table) are written branch-free on purpose.  With early returns the
// compiler sinks the later loads into the "not found yet" branches and a miss costs several DEPENDENT memory round
// trips; with selects every load of the bucket is issued before the first wait.
// last = the id word of a bucket's last slot: did a key that belongs here (or passed through) go on to
// the next bucket?  (The builder sets SPL_OVF_BIT there; an empty slot is all-ones.)
SPL_HD bool bucket_overflowed(uint32_t last) { return last != SPL_EMPTY && (last & SPL_OVF_BIT) != 0u; }

// (k0 & 0xFFFF = the key's first two bytes, the second one zero for a one-byte key)
SPL_HD uint32_t key_salt(const DeviceTables& T, uint32_t k0) { return (uint32_t)T.len_mask[k0 & 0xFFFFu] >> 8; }
SPL_HD uint32_t tiny_salt(const DeviceTables& T, uint32_t k0) { return T.pfx[k0 & 0xFFFFu].lm >> 16; }
SPL_HD uint32_t t8_salt(const DeviceTables& T, uint32_t k0) { return (uint32_t)T.filt4[hash_f4(k0) >> T.filt4_shift] >> SPL_F4_MASK_BITS; }

// keys of 1..4 bytes: ONE entry (8 bytes); salt = tiny_salt(first two bytes)
SPL_HD uint32_t probe_tiny(const DeviceTables& T, uint32_t k0, uint32_t n, uint32_t salt) {
    const uint32_t* e = T.tiny_tab + (size_t)(hash_tiny(k0, n, salt) & T.tiny_mask) * SPL_TINY_WORDS;
    const uint32_t ek = e[0], ei = e[1];
    return ((ek == k0) & ((ei >> 24) == n)) ? (ei & SPL_ID_MASK) : SPL_NO_RANK;
}
// keys of 5..8 bytes: ONE entry (12 bytes); salt = t8_salt(first four bytes)
SPL_HD uint32_t probe_t8(const DeviceTables& T, uint32_t k0, uint32_t k1, uint32_t n, uint32_t salt) {
    const uint32_t* e = T.t8_tab + (size_t)(hash_t8(k0, k1, n, salt) & T.t8_mask) * SPL_T8_WORDS;
    const uint32_t e0 = e[0], e1 = e[1], ei = e[2];
    return ((e0 == k0) & (e1 == k1) & ((ei >> 24) == n)) ? (ei & SPL_ID_MASK) : SPL_NO_RANK;
}
// keys of 9..12 bytes: buckets of four, salt = key_salt(first two bytes)
SPL_HD uint32_t probe_short12(const DeviceTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t n, uint32_t salt) {
    uint32_t bkt = hash_short(k0, k1, k2, n, salt) & T.short_mask;
    for (;;) {
        const Quad* q = reinterpret_cast<const Quad*>(T.short_tab + (size_t)bkt * SPL_SHORT_BUCKET);
        const Quad e0 = q[0], e1 = q[1], e2 = q[2], e3 = q[3];     // four independent 16-byte loads
        const bool f0 = (e0.x == k0) & (e0.y == k1) & (e0.z == k2) & ((e0.w >> 24) == n);
        const bool f1 = (e1.x == k0) & (e1.y == k1) & (e1.z == k2) & ((e1.w >> 24) == n);
        const bool f2 = (e2.x == k0) & (e2.y == k1) & (e2.z == k2) & ((e2.w >> 24) == n);
        const bool f3 = (e3.x == k0) & (e3.y == k1) & (e3.z == k2) & ((e3.w >> 24) == n);
        uint32_t r = SPL_NO_RANK;
        r = f3 ? (e3.w & SPL_ID_MASK) : r;
        r = f2 ? (e2.w & SPL_ID_MASK) : r;
        r = f1 ? (e1.w & SPL_ID_MASK) : r;
        r = f0 ? (e0.w & SPL_ID_MASK) : r;
        if ((f0 | f1 | f2 | f3) | !bucket_overflowed(e3.w)) return r;   // found, or nothing overflowed from here
        bkt = (bkt + 1) & T.short_mask;
    }
}
// keys of up to 12 bytes, by length class (each class looks its own salt up)
SPL_HD uint32_t probe_short(const DeviceTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t n) {
    if (n <= (uint32_t)SPL_TINY_MAX) return probe_tiny(T, k0, n, tiny_salt(T, k0));
    if (n <= (uint32_t)SPL_T8_MAX) return probe_t8(T, k0, k1, n, t8_salt(T, k0));
    return probe_short12(T, k0, k1, k2, n, key_salt(T, k0));
}

template <class TX> SPL_HD uint32_t probe_long(const DeviceTables& T, const TX& tx, int p, int n) {
    uint32_t h = 0;
    for (int i = 0; i < n; i += 4) h = hash_long_step(h, mask_tail(tx.load32(p + i), n - i));
    h = hash_long_fin(h, (uint32_t)n);
    const uint32_t tag = hash_long_tag(h);
    uint32_t slot = h & T.long_mask;
    for (;;) {
        const LongEnt e = T.long_tab[slot];
        if (e.id == SPL_EMPTY) return SPL_NO_RANK;
        if (e.tag == tag && e.len == (uint32_t)n) {
            const uint32_t* kw = reinterpret_cast<const uint32_t*>(T.key_blob + e.off);
            bool eq = true;
            for (int i = 0; i < n && eq; i += 4) eq = kw[i >> 2] == mask_tail(tx.load32(p + i), n - i);
            if (eq) return e.id;
        }
        slot = (slot + 1) & T.long_mask;
    }
}

// Whole-chunk lookup: id of the token whose bytes are text[p, p+n), or SPL_NO_RANK.
template <class TX> SPL_HD uint32_t probe_chunk(const DeviceTables& T, const TX& tx, int p, int n) {
    if (n <= SPL_SHORT_MAX) {
        const uint32_t k0 = mask_tail(tx.load32(p), n);
        const uint32_t k1 = n > 4 ? mask_tail(tx.load32(p + 4), n - 4) : 0u;
        const uint32_t k2 = n > 8 ? mask_tail(tx.load32(p + 8), n - 8) : 0u;
        return probe_short(T, k0, k1, k2, (uint32_t)n);
    }
    if ((uint32_t)n > T.max_key_len) return SPL_NO_RANK;
    return probe_long(T, tx, p, n);
}

SPL_HD uint32_t pair_rank(const DeviceTables& T, uint32_t l, uint32_t r) {
    if ((l | r) > SPL_ID_MASK) return SPL_NO_RANK;      // an unknown single byte never merges
    const uint64_t key = pair_key(l, r);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);       // key occupies bits 0..41
    uint32_t bkt = hash_pair(l, r) & T.pair_mask;
    for (;;) {
        const Quad* q = reinterpret_cast<const Quad*>(T.pair_tab + (size_t)bkt * SPL_PAIR_BUCKET);
        const Quad a = q[0], c = q[1];                              // 4 entries, two 16-byte loads
        const bool f0 = (a.x == klo) & ((a.y & 0x3FFu) == khi);
        const bool f1 = (a.z == klo) & ((a.w & 0x3FFu) == khi);
        const bool f2 = (c.x == klo) & ((c.y & 0x3FFu) == khi);
        const bool f3 = (c.z == klo) & ((c.w & 0x3FFu) == khi);
        uint32_t res = SPL_NO_RANK;
        res = f3 ? (c.w >> 10) : res;
        res = f2 ? (c.y >> 10) : res;
        res = f1 ? (a.w >> 10) : res;
        res = f0 ? (a.y >> 10) : res;
        if ((f0 | f1 | f2 | f3) | ((c.z & c.w) == 0xFFFFFFFFu)) return res;   // found / bucket not full
        bkt = (bkt + 1) & T.pair_mask;
    }
}

// Split form of pair_rank for callers that want several probes in flight: issue the bucket loads
// of all of them, then resolve each (falling back to the generic loop on a bucket overflow).
struct PairProbe { Quad a, c; uint32_t bkt, klo, khi; bool valid; };
SPL_HD void pair_issue(const DeviceTables& T, uint32_t l, uint32_t r, PairProbe& pp) {
    pp.valid = (l | r) <= SPL_ID_MASK;
    const uint64_t key = pair_key(l & SPL_ID_MASK, r & SPL_ID_MASK);
    pp.klo = (uint32_t)key; pp.khi = (uint32_t)(key >> 32);
    pp.bkt = hash_pair(l & SPL_ID_MASK, r & SPL_ID_MASK) & T.pair_mask;
    const Quad* q = reinterpret_cast<const Quad*>(T.pair_tab + (size_t)pp.bkt * SPL_PAIR_BUCKET);
    pp.a = q[0]; pp.c = q[1];
}
SPL_HD uint32_t pair_finish(const DeviceTables& T, const PairProbe& pp) {
    if (!pp.valid) return SPL_NO_RANK;
    const bool f0 = (pp.a.x == pp.klo) & ((pp.a.y & 0x3FFu) == pp.khi);
    const bool f1 = (pp.a.z == pp.klo) & ((pp.a.w & 0x3FFu) == pp.khi);
    const bool f2 = (pp.c.x == pp.klo) & ((pp.c.y & 0x3FFu) == pp.khi);
    const bool f3 = (pp.c.z == pp.klo) & ((pp.c.w & 0x3FFu) == pp.khi);
    uint32_t res = SPL_NO_RANK;
    res = f3 ? (pp.c.w >> 10) : res;
    res = f2 ? (pp.c.y >> 10) : res;
    res = f1 ? (pp.a.w >> 10) : res;
    res = f0 ? (pp.a.y >> 10) : res;
    if ((f0 | f1 | f2 | f3) | ((pp.c.z & pp.c.w) == 0xFFFFFFFFu)) return res;
    // home bucket full without a match (rare): continue with the generic probe from the next bucket
    uint32_t bkt = (pp.bkt + 1) & T.pair_mask;
    for (;;) {
        const Quad* q = reinterpret_cast<const Quad*>(T.pair_tab + (size_t)bkt * SPL_PAIR_BUCKET);
        const Quad a = q[0], c = q[1];
        const bool g0 = (a.x == pp.klo) & ((a.y & 0x3FFu) == pp.khi);
        const bool g1 = (a.z == pp.klo) & ((a.w & 0x3FFu) == pp.khi);
        const bool g2 = (c.x == pp.klo) & ((c.y & 0x3FFu) == pp.khi);
        const bool g3 = (c.z == pp.klo) & ((c.w & 0x3FFu) == pp.khi);
        uint32_t r2 = SPL_NO_RANK;
        r2 = g3 ? (c.w >> 10) : r2;
        r2 = g2 ? (c.y >> 10) : r2;
        r2 = g1 ? (a.w >> 10) : r2;
        r2 = g0 ? (a.y >> 10) : r2;
        if ((g0 | g1 | g2 | g3) | ((c.z & c.w) == 0xFFFFFFFFu)) return r2;
        bkt = (bkt + 1) & T.pair_mask;
    }
}

// One lane, one chunk.  S provides per-node storage:  uint32_t& id(int i), uint32_t& rk(int i).
// On return node i is alive iff id(i) != SPL_DEAD; alive nodes in index order are the tokens
// (id SPL_NO_RANK = an unknown single byte: emits nothing, bpe.rs:187-191).
template <class S, class TX> SPL_HD void bpe_serial(const DeviceTables& T, S& s, const TX& tx, int p, int n) {
    // bytes -> ids and the initial pair ranks in batches, so that the independent loads of a batch
    // are all in flight before the first one is consumed
    for (int i0 = 0; i0 < n; i0 += 8) {
        uint32_t bb[8], iv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) bb[k] = i0 + k < n ? tx.txt(p + i0 + k) : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) iv[k] = T.byte_id[bb[k]];
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < n) s.id(i0 + k) = iv[k];
    }
    for (int i0 = 0; i0 + 1 < n; i0 += 4) {
        PairProbe pp[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool on = i0 + k + 1 < n;
            pair_issue(T, on ? s.id(i0 + k) : 0u, on ? s.id(i0 + k + 1) : 0u, pp[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) if (i0 + k + 1 < n) s.rk(i0 + k) = pair_finish(T, pp[k]);
    }
    s.rk(n - 1) = SPL_NO_RANK;
    for (;;) {
        uint32_t mn = SPL_NO_RANK;
        int mi = -1;
        for (int i0 = 0; i0 < n; i0 += 8) {             // strict '<' => leftmost minimum (bpe.rs:133);
            uint32_t r[8];                              // eight independent reads per round
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = i0 + k < n ? s.rk(i0 + k) : SPL_NO_RANK;
#pragma unroll
            for (int k = 0; k < 8; k++) if (r[k] < mn) { mn = r[k]; mi = i0 + k; }
        }
        if (mi < 0) break;
        int j = mi + 1;
        while (s.id(j) == SPL_DEAD) j++;                // right neighbour (exists: rk(mi) is a rank)
        s.id(mi) = mn;                                  // the merged token's id IS the pair's rank
        s.id(j) = SPL_DEAD;
        s.rk(j) = SPL_NO_RANK;
        int j2 = j + 1;
        while (j2 < n && s.id(j2) == SPL_DEAD) j2++;
        int h = mi - 1;
        while (h >= 0 && s.id(h) == SPL_DEAD) h--;
        // both re-rank probes in flight together (bpe.rs:160-166)
        PairProbe pr, ph;
        pair_issue(T, mn, j2 < n ? s.id(j2) : 0u, pr);
        pair_issue(T, h >= 0 ? s.id(h) : 0u, mn, ph);
        s.rk(mi) = j2 < n ? pair_finish(T, pr) : SPL_NO_RANK;
        if (h >= 0) s.rk(h) = pair_finish(T, ph);
    }
}

}  // namespace spl
