// spl_k_special.h -- part of spl_kernels.hip (included there, in this order; one translation unit): document-start marks and special-token literals (k_mark_docs, k_special_scan / _ends / _select): SPL_WITH_SPECIAL batches, in front of k_pretok.
#pragma once

namespace spl {

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

}  // namespace spl
