// spl_scan.h -- the split patterns as a deterministic, backtracking-free scanner.
//
// Replaces RegexBackend::find_iter (reference src/core/tokenizer.rs:244-257) for the
// in-scope patterns CL100K_BASE_PATTERN (:39), O200K_BASE_PATTERN (:42, also llama3 :45 and
// deepseek_v3, src/python/bindings.rs:116-129) and MISTRAL_V3_PATTERN (:64: the o200k letter
// alternatives without the contraction suffix, ONE \p{N} per match, and [\r\n/]* behind the
// "other" run).  `match_end(p)` returns the end of the
// leftmost-first match that starts at p; because the patterns tile the text, the match
// starts of a text are the orbit of 0 under match_end.  Each alternative's greedy /
// backtracking behaviour has been reduced to a closed form over character classes (derivation
// in DESIGN.md "Scanner"); `is_sync` gives the context-free match starts ("sync points") that
// let many lanes scan one text independently.
//
// All functions are templates over an accessor A with
//     uint32_t rec(int q)   class record of the byte at q (spl_common.h CB_*), C_WEND past the end
//     uint32_t txt(int q)   text byte at q
// so the same code runs over an LDS window (kernel), over global memory (deferred path) and on
// the host (tests/hostsim).
#pragma once
#include "spl_common.h"

namespace spl {

constexpr int SPL_DEFER = -1;   // the window end was reached: the caller must re-scan elsewhere

SPL_HD uint32_t utf8_len(uint32_t b) { return b < 0xC0u ? 1u : b < 0xE0u ? 2u : b < 0xF0u ? 3u : 4u; }

// Class seen when LOOKING AHEAD at q: a text start reads as end-of-text.
template <class A> SPL_HD uint32_t peek(const A& a, int q, int& len) {
    uint32_t r = a.rec(q);
    len = (int)(r >> CB_LEN_SHIFT) + 1;
    return (r & CB_TSTART) ? (uint32_t)C_EOT : (r & CB_CLASS);
}

// Greedy run over `mask` starting at q; returns the stop position and the class that stopped it.
template <class A> SPL_HD int run_mask(const A& a, int q, uint32_t mask, uint32_t& stopc) {
    for (;;) {
        int l;
        uint32_t c = peek(a, q, l);
        if (!(SPL_BIT(c) & mask)) { stopc = c; return q; }
        q += l;
    }
}

// (?i:'s|'t|'re|'ve|'m|'ll|'d) with the apostrophe at `ap`.  Caseless partners under UTF|UCP
// (probed from PCRE2, tools/gen_unicode_tables.py): ASCII case pairs plus U+017F for 's'.
// Returns the end, 0 if no contraction, SPL_DEFER if the window ends first.
template <class A> SPL_HD int contraction(const A& a, int ap) {
    const int q1 = ap + 1;
    const uint32_t r1 = a.rec(q1);
    if ((r1 & CB_CLASS) == C_WEND) return SPL_DEFER;
    if (r1 & CB_TSTART) return 0;
    const uint32_t c1 = r1 & CB_CLASS;
    if (c1 != C_LU && c1 != C_LL) return 0;
    const uint32_t b = a.txt(q1);
    if (b < 0x80u) {
        const uint32_t lo = b | 0x20u;
        if (lo == 's' || lo == 't' || lo == 'm' || lo == 'd') return q1 + 1;
        if (lo == 'r' || lo == 'v' || lo == 'l') {
            const int q2 = q1 + 1;
            const uint32_t r2 = a.rec(q2);
            if ((r2 & CB_CLASS) == C_WEND) return SPL_DEFER;
            if (r2 & CB_TSTART) return 0;
            const uint32_t want = (lo == 'l') ? 'l' : 'e';
            return ((a.txt(q2) | 0x20u) == want) ? q2 + 1 : 0;
        }
        return 0;
    }
    if (b == 0xC5u && a.txt(q1 + 1) == 0xBFu) return q1 + 2;   // U+017F LATIN SMALL LETTER LONG S
    return 0;
}

// Alternatives 3..7, identical in both patterns:
//   \p{N}{1,3} | ?[^\s\p{L}\p{N}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
// c = real class at p (not L), q1/c1/l1 = next position, its look-ahead class and length.
// MISTRAL_V3_PATTERN's  ?[^\s\p{L}\p{N}]+[\r\n/]* : behind the maximal "other" run (which ends before q
// and took every '/' it could) the longest stretch of CR, LF and '/'.
template <class A> SPL_HD int run_nl_slash(const A& a, int q, uint32_t& stopc) {
    for (;;) {
        int l;
        const uint32_t c = peek(a, q, l);
        if (c == C_NL || (c == C_P && a.txt(q) == '/')) { q += l; continue; }
        stopc = c;
        return q;
    }
}
template <class A> SPL_HD int match_tail(const A& a, int p, uint32_t c, int q1, uint32_t c1, int l1, bool mistral = false) {
    uint32_t stopc;
    if (c == C_N) {                                  // up to three numbers (mistral: exactly one)
        if (mistral || c1 != C_N) return q1;
        const int q2 = q1 + l1;
        int l2;
        const uint32_t c2 = peek(a, q2, l2);
        if (c2 == C_WEND) return SPL_DEFER;
        return c2 == C_N ? q2 + l2 : q2;
    }
    int start = -1;
    if (SPL_BIT(c) & M_OTHER) start = q1;                                  // " ?" matched nothing
    else if (c == C_SP && (SPL_BIT(c1) & M_OTHER)) start = q1 + l1;        // " ?" took the space
    if (start >= 0) {
        int q = run_mask(a, start, M_OTHER, stopc);
        if (stopc == C_NL) q = mistral ? run_nl_slash(a, q, stopc) : run_mask(a, q, SPL_BIT(C_NL), stopc);
        return stopc == C_WEND ? SPL_DEFER : q;
    }
    // c is whitespace.  One pass over the maximal \s run [p, q) remembering the end of its
    // last CR/LF and the start of its last character.
    int q = q1, last_nl_end = (c == C_NL) ? q1 : -1, last_char = p;
    for (;;) {
        int l;
        const uint32_t cc = peek(a, q, l);
        if (!(SPL_BIT(cc) & M_S)) { stopc = cc; break; }
        last_char = q;
        q += l;
        if (cc == C_NL) last_nl_end = q;
    }
    if (stopc == C_WEND) return SPL_DEFER;
    if (last_nl_end >= 0) return last_nl_end;        // \s*[\r\n]+  : through the LAST newline
    if (stopc == C_EOT) return q;                    // \s+(?!\S)   : at end of text takes all
    if (last_char > p) return last_char;             // \s+(?!\S)   : gives back its last char
    return q;                                        // \s+         : lone whitespace char
}

template <class A> SPL_HD int match_end_cl100k(const A& a, int p) {
    const uint32_t r0 = a.rec(p);
    const uint32_t c = r0 & CB_CLASS;
    const int q1 = p + (int)(r0 >> CB_LEN_SHIFT) + 1;
    int l1;
    const uint32_t c1 = peek(a, q1, l1);
    if (c1 == C_WEND) return SPL_DEFER;
    uint32_t stopc;
    if (c == C_AP && (SPL_BIT(c1) & M_L)) {          // (?i:'s|'t|'re|'ve|'m|'ll|'d)
        const int e = contraction(a, p);
        if (e != 0) return e;
    }
    if (SPL_BIT(c) & M_L) {                          // [^\r\n\p{L}\p{N}]?\p{L}+ , prefix empty
        const int q = run_mask(a, q1, M_L, stopc);
        return stopc == C_WEND ? SPL_DEFER : q;
    }
    if ((SPL_BIT(c) & M_X) && (SPL_BIT(c1) & M_L)) { // one-char prefix then letters
        const int q = run_mask(a, q1 + l1, M_L, stopc);
        return stopc == C_WEND ? SPL_DEFER : q;
    }
    return match_tail(a, p, c, q1, c1, l1);
}

// o200k letter bodies starting at s (first char class/len given explicitly):
//   alt1  [Lu Lt Lm Lo M]* [Ll Lm Lo M]+      (U* W+, U* gives back down to its last W member)
//   alt2  [Lu Lt Lm Lo M]+ [Ll Lm Lo M]*      (only reachable when alt1 failed => W* is empty)
// Returns end (>s), 0 on failure, SPL_DEFER.
template <class A> SPL_HD int letters_o200k(const A& a, int s, uint32_t cfirst, int lfirst, bool allow_alt2) {
    int q = s, last_w_end = -1, l = lfirst;
    uint32_t cc = cfirst;
    while (SPL_BIT(cc) & M_U) {
        q += l;
        if (SPL_BIT(cc) & M_UW) last_w_end = q;
        cc = peek(a, q, l);
    }
    if (cc == C_WEND) return SPL_DEFER;
    if (cc == C_LL) {
        uint32_t stopc;
        q = run_mask(a, q + l, M_W, stopc);
        return stopc == C_WEND ? SPL_DEFER : q;
    }
    if (last_w_end >= 0) return last_w_end;
    if (allow_alt2 && q > s) return q;
    return 0;
}
// optional trailing contraction
template <class A> SPL_HD int with_contraction(const A& a, int e) {
    int l;
    const uint32_t c = peek(a, e, l);
    if (c == C_WEND) return SPL_DEFER;
    if (c != C_AP) return e;
    const int ce = contraction(a, e);
    return ce == 0 ? e : ce;
}

// (mistral: MISTRAL_V3_PATTERN -- no contraction suffix, single numbers, [\r\n/]* tail)
template <class A> SPL_HD int match_end_o200k(const A& a, int p, bool mistral = false) {
    const uint32_t r0 = a.rec(p);
    const uint32_t c = r0 & CB_CLASS;
    const int l0 = (int)(r0 >> CB_LEN_SHIFT) + 1;
    const int q1 = p + l0;
    int l1;
    const uint32_t c1 = peek(a, q1, l1);
    if (c1 == C_WEND) return SPL_DEFER;
    constexpr uint32_t M_LM = M_L | SPL_BIT(C_M);
    if (SPL_BIT(c) & (M_X & ~SPL_BIT(C_M))) {        // prefix char that cannot itself be a body char
        if (SPL_BIT(c1) & M_LM) {
            const int e = letters_o200k(a, q1, c1, l1, true);   // never fails for c1 in L|M
            if (e == SPL_DEFER) return e;
            if (e > 0) return mistral ? e : with_contraction(a, e);
        }
    } else if (c == C_M) {                           // a mark is a legal prefix AND a legal body char
        if (SPL_BIT(c1) & M_LM) {
            const int e = letters_o200k(a, q1, c1, l1, false);  // alt1 with the mark as prefix
            if (e == SPL_DEFER) return e;
            if (e > 0) return mistral ? e : with_contraction(a, e);
        }
        const int e = letters_o200k(a, p, c, l0, false);        // alt1, empty prefix: always matches
        if (e == SPL_DEFER) return e;
        return mistral ? e : with_contraction(a, e);
    } else if (SPL_BIT(c) & M_L) {
        const int e = letters_o200k(a, p, c, l0, true);
        if (e == SPL_DEFER) return e;
        return mistral ? e : with_contraction(a, e);
    }
    return match_tail(a, p, c, q1, c1, l1, mistral);
}

template <class A> SPL_HD int match_end(const A& a, int p, int pattern) {
    return pattern == PAT_CL100K ? match_end_cl100k(a, p) : match_end_o200k(a, p, pattern == PAT_MISTRAL_V3);
}

// Context-free match starts.  prev = class of the previous character of the SAME text, cur =
// class of the character at the position.  (Proofs in DESIGN.md "Sync points"; brute-force
// checked against PCRE2 in tests/test_hostsim.py.)
SPL_HD bool is_sync(int pattern, uint32_t prev, uint32_t cur) {
    const uint32_t pb = SPL_BIT(prev), cb = SPL_BIT(cur);
    // mistral_v3: \p{N} matches ONE number, so every number starts a match and ends one
    if (pattern == PAT_MISTRAL_V3 && prev < C_EOT && (cur == C_N || prev == C_N)) return true;
    // (f) a number after anything but a number: no alternative has a digit behind a non-digit
    //     inside one match (only \p{N}{1,3} consumes digits, and it starts with one)
    if (cur == C_N && prev < C_EOT) return prev != C_N;
    // (g) cl100k: "other" after a newline -- every match that consumes a newline ends with its
    //     newline run (o200k's [\r\n/]* may go on with '/', so not there)
    if (pattern == PAT_CL100K && prev == C_NL && (cb & M_OTHER)) return true;
    if (pb & M_L)      // (o200k: a contraction suffix may follow the letters; mistral_v3 has none)
        return pattern == PAT_CL100K ? !(cb & M_L)
             : pattern == PAT_O200K  ? !(cb & (M_L | SPL_BIT(C_M) | SPL_BIT(C_AP))) : !(cb & (M_L | SPL_BIT(C_M)));
    if (prev == C_N) return cur != C_N;
    if (prev == C_NL) return (cb & (M_L | SPL_BIT(C_N))) != 0;
    if ((pb & (SPL_BIT(C_P) | SPL_BIT(C_AP))) || (pattern == PAT_CL100K && prev == C_M))
        return (cb & (SPL_BIT(C_SP) | SPL_BIT(C_WS))) != 0;
    return false;
}

// Two-stage class lookup of a code point.
SPL_HD uint32_t cp_class(const DeviceTables& T, uint32_t cp) {
    if (cp >= 0x110000u) return C_P;
    if (T.cjk_fast && ((cp - 0x4E00u) < 0x5200u || (cp - 0xAC00u) < 0x2BA4u)) return C_LO;
    const uint32_t blk = T.ucls_stage1[cp >> T.ucls_shift];
    return T.ucls_stage2[(blk << T.ucls_shift) | (cp & ((1u << T.ucls_shift) - 1u))];
}
// Decode the character whose lead byte b0 sits at q (continuation bytes through tx).
template <class TX> SPL_HD uint32_t decode_at(const TX& tx, int q, uint32_t b0) {
    if (b0 < 0xE0u) return ((b0 & 0x1Fu) << 6) | (tx.txt(q + 1) & 0x3Fu);
    if (b0 < 0xF0u) return ((b0 & 0x0Fu) << 12) | ((tx.txt(q + 1) & 0x3Fu) << 6) | (tx.txt(q + 2) & 0x3Fu);
    return ((b0 & 0x07u) << 18) | ((tx.txt(q + 1) & 0x3Fu) << 12) | ((tx.txt(q + 2) & 0x3Fu) << 6) |
           (tx.txt(q + 3) & 0x3Fu);
}

// Class record (class | (length - 1) << CB_LEN_SHIFT) of the byte at q -- the ONE place that decides what
// a character is, including for text that is not valid UTF-8 (policy in include/splintr_hip.h):
//   * a lead byte takes the continuation bytes that follow it IN THE SAME TEXT, at most as many as it
//     announces; all of them present: the class of the decoded value (looked up as it is), otherwise
//     the bytes it got form one character of class "other";
//   * a continuation byte that no lead byte of its text reaches is a character of class "other";
//     one that is reached is C_CONT (not a character start).
// tx.txt(i): text byte; ts(i): a text starts at i; [lo, hi): the bytes that exist for this purpose
// (ts is asked for positions in [lo, hi] only); ascii(c): record of an ASCII byte.
template <class TX, class TS, class ASC>
SPL_HD uint32_t byte_record(const DeviceTables& T, const TX& tx, const TS& ts, const ASC& ascii, int q, int lo, int hi) {
    const uint32_t c0 = tx.txt(q);
    if (c0 < 0x80u) return ascii(c0);
    if (c0 < 0xC0u) {
        for (int k = 1; k <= 3; k++) {
            if (ts(q - k + 1) || q - k < lo) return C_P;
            const uint32_t b = tx.txt(q - k);
            if (b >= 0xC0u) return utf8_len(b) > (uint32_t)k ? (uint32_t)C_CONT : (uint32_t)C_P;
            if (b < 0x80u) return C_P;
        }
        return C_P;
    }
    const uint32_t want = utf8_len(c0);
    uint32_t len = 1;
    while (len < want && q + (int)len < hi && !ts(q + (int)len) && (tx.txt(q + (int)len) & 0xC0u) == 0x80u) len++;
    const uint32_t cls = len == want ? cp_class(T, decode_at(tx, q, c0)) : (uint32_t)C_P;
    return cls | ((len - 1) << CB_LEN_SHIFT);
}

}  // namespace spl
