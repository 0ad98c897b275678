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

SPL_HD uint32_t probe_short(const DeviceTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t n) {
    uint32_t slot = hash_short(k0, k1, k2, n) & T.short_mask;
    for (;;) {
        const ShortEnt e = T.short_tab[slot];
        if (e.id_len == SPL_EMPTY) return SPL_NO_RANK;
        if (e.k0 == k0 && e.k1 == k1 && e.k2 == k2 && (e.id_len >> 24) == n) return e.id_len & 0xFFFFFFu;
        slot = (slot + 1) & T.short_mask;
    }
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
    uint32_t slot = hash_pair(l, r) & T.pair_mask;
    for (;;) {
        const uint64_t e = T.pair_tab[slot];
        if (e == SPL_PAIR_EMPTY) return SPL_NO_RANK;
        if ((e & SPL_PAIR_KEY_MASK) == key) return (uint32_t)(e >> (2 * SPL_ID_BITS));
        slot = (slot + 1) & T.pair_mask;
    }
}

// One lane, one chunk.  S provides per-node storage:  uint32_t& id(int i), uint32_t& rk(int i).
// On return node i is alive iff id(i) != SPL_DEAD; alive nodes in index order are the tokens
// (id SPL_NO_RANK = an unknown single byte: emits nothing, bpe.rs:187-191).
template <class S, class TX> SPL_HD void bpe_serial(const DeviceTables& T, S& s, const TX& tx, int p, int n) {
    for (int i = 0; i < n; i++) s.id(i) = T.byte_id[tx.txt(p + i)];
    for (int i = 0; i + 1 < n; i++) s.rk(i) = pair_rank(T, s.id(i), s.id(i + 1));
    s.rk(n - 1) = SPL_NO_RANK;
    for (;;) {
        uint32_t mn = SPL_NO_RANK;
        int mi = -1;
        for (int i = 0; i < n; i++) {                   // strict '<' => leftmost minimum (bpe.rs:133)
            const uint32_t r = s.rk(i);
            if (r < mn) { mn = r; mi = i; }
        }
        if (mi < 0) break;
        int j = mi + 1;
        while (s.id(j) == SPL_DEAD) j++;                // right neighbour (exists: rk(mi) is a rank)
        s.id(mi) = mn;                                  // the merged token's id IS the pair's rank
        s.id(j) = SPL_DEAD;
        s.rk(j) = SPL_NO_RANK;
        int j2 = j + 1;
        while (j2 < n && s.id(j2) == SPL_DEAD) j2++;
        s.rk(mi) = j2 < n ? pair_rank(T, mn, s.id(j2)) : SPL_NO_RANK;
        int h = mi - 1;
        while (h >= 0 && s.id(h) == SPL_DEAD) h--;
        if (h >= 0) s.rk(h) = pair_rank(T, s.id(h), mn);
    }
}

}  // namespace spl
