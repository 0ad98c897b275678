// spl_scan_masks.h -- the scanner's run searches on class BITMASKS instead of byte loops.
//
// spl_scan.h walks a run one character (one dependent LDS read) at a time.  Here the window's
// class records are first condensed into position-ordered bitmasks (one bit per byte, 32-bit
// words; continuation bytes inherit the kind of their lead byte so runs are contiguous in byte
// space), and "where does this run end", "where is its last newline", "where does its last
// character start" become a handful of word operations (v_ffbl / v_ffbh).  The match semantics
// are exactly those of spl_scan.h (same alternatives, same closed forms); tests/hostsim checks
// the two against each other and against the CPU reference restatement kept under tests.
//
// Accessor MA:
//     uint32_t mw(int which, int w)   word w of mask `which` (MK_*), zero past the window
//     uint32_t rec(int q), txt(int q) as in spl_scan.h (class records are still consulted for the
//                                     dispatch on the first character and for o200k letter runs)
//     int      wbits()                window size W in bytes (multiple of 32)
//     bool     end_is_eot()           true if the text ends at or before W (then TS is set there)
#pragma once
#include "spl_scan.h"

namespace spl {

// (MK_SP: U+0020 only.  MK_UP: Lu and Lt.  MK_LB: Lm and Lo.  Set by the mask builder, not kinds of a class: MK_SL: the byte '/';
//  MK_BAD: bytes that keep a window off the bit-vector start computation of spl_scan_starts.h, see bad_for_starts)
enum : int { MK_L = 0, MK_N, MK_S, MK_NL, MK_O, MK_M, MK_AP, MK_SP, MK_UP, MK_LB, MK_SL, MK_BAD, MK_CS, MK_TS, MK_SY, MK_COUNT };

// kind of one class code, as mask membership bits (bit MK_x)
SPL_HD uint32_t kind_bits(uint32_t cls) {
    const uint32_t b = SPL_BIT(cls);
    uint32_t k = 0;
    if (b & M_L) k |= 1u << MK_L;
    if (cls == C_N) k |= 1u << MK_N;
    if (b & M_S) k |= 1u << MK_S;
    if (cls == C_NL) k |= 1u << MK_NL;
    if (b & M_OTHER) k |= 1u << MK_O;
    if (cls == C_M) k |= 1u << MK_M;
    if (cls == C_AP) k |= 1u << MK_AP;
    if (cls == C_SP) k |= 1u << MK_SP;
    if (cls == C_LU || cls == C_LT) k |= 1u << MK_UP;
    if (cls == C_LM || cls == C_LO) k |= 1u << MK_LB;
    return k;
}

// Does this byte keep its window off the bit-vector start computation?  r = its class record, kc = the class
// of its character (a continuation byte: of its lead), in_text = the byte lies before the end of the text.
//   all patterns : a multi-byte number or whitespace character (the formulas count those by bytes; letters
//                  and "other" characters only enter through run logic and whole-character masks);
//                  a span without text (special-token literal)
//   o200k family : also marks (a mark is a body character of the letter alternatives AND a legal prefix)
SPL_HD bool bad_for_starts(int pattern, uint32_t r, uint32_t kc, bool in_text) {
    const uint32_t cls = r & CB_CLASS;
    const bool multi = cls == C_CONT || (r >> CB_LEN_SHIFT) != 0u;
    if (multi && !(kc < C_EOT && (SPL_BIT(kc) & (M_L | M_OTHER)))) return true;
    if (cls == C_EOT && in_text) return true;
    if (pattern != PAT_CL100K && kc == C_M) return true;
    return false;
}

// The classes of each kind as a set (bit c = class c): a byte's kind bit is (1 << class) & kind_classes(k).
SPL_HD constexpr uint32_t kind_classes(int k) {
    return k == MK_L ? M_L : k == MK_N ? SPL_BIT(C_N) : k == MK_S ? M_S : k == MK_NL ? SPL_BIT(C_NL) : k == MK_O ? M_OTHER
         : k == MK_M ? SPL_BIT(C_M) : k == MK_AP ? SPL_BIT(C_AP) : k == MK_SP ? SPL_BIT(C_SP)
         : k == MK_UP ? (SPL_BIT(C_LU) | SPL_BIT(C_LT)) : k == MK_LB ? (SPL_BIT(C_LM) | SPL_BIT(C_LO)) : 0u;
}

SPL_HD int ctz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}
SPL_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return __builtin_clz(x);
#endif
}

// First position >= q where the run of `which` stops: the kind bit is clear OR a text starts.
// Returns W if the run reaches the end of the window.
template <class MA> SPL_HD int run_end_m(const MA& m, int which, int q) {
    const int W = m.wbits();
    if (q >= W) return W;
    int w = q >> 5;
    uint32_t stop = (~m.mw(which, w) | m.mw(MK_TS, w)) & (~0u << (q & 31));
    const int nw = W >> 5;
    while (!stop) {
        if (++w >= nw) return W;
        stop = ~m.mw(which, w) | m.mw(MK_TS, w);
    }
    return (w << 5) + ctz32(stop);
}
// Highest set bit of `which` in [lo, hi), or -1.
template <class MA> SPL_HD int last_bit_in_m(const MA& m, int which, int lo, int hi) {
    if (hi <= lo) return -1;
    int w = (hi - 1) >> 5;
    const int wlo = lo >> 5;
    uint32_t x = m.mw(which, w);
    const int top = hi - (w << 5);                       // 1..32 bits of this word are below hi
    if (top < 32) x &= (1u << top) - 1u;
    for (;;) {
        if (w == wlo) x &= ~0u << (lo & 31);
        if (x) return (w << 5) + 31 - clz32(x);
        if (--w < wlo) return -1;
        x = m.mw(which, w);
    }
}
template <class MA> SPL_HD bool bit_m(const MA& m, int which, int q) { return (m.mw(which, q >> 5) >> (q & 31)) & 1u; }

// "Look-ahead class" at q with the sentinels of spl_scan.h: text start -> C_EOT, past the window
// -> C_WEND (or C_EOT when the text ends there).
template <class MA> SPL_HD uint32_t peek_m(const MA& m, int q, int& len) {
    if (q >= m.wbits()) { len = 1; return m.end_is_eot() ? (uint32_t)C_EOT : (uint32_t)C_WEND; }
    return peek(m, q, len);
}
#define SPL_RUN_OR_DEFER(e) do { if ((e) >= m.wbits() && !m.end_is_eot()) return SPL_DEFER; } while (0)

// Alternatives 3..7 (see match_tail in spl_scan.h) on masks.
template <class MA> SPL_HD int match_tail_m(const MA& m, int p, uint32_t c, int q1, uint32_t c1, int l1, bool mistral = false) {
    if (c == C_N) {
        if (mistral || c1 != C_N) return q1;
        const int q2 = q1 + l1;
        int l2;
        const uint32_t c2 = peek_m(m, q2, l2);
        if (c2 == C_WEND) return SPL_DEFER;
        return c2 == C_N ? q2 + l2 : q2;
    }
    int start = -1;
    if (SPL_BIT(c) & M_OTHER) start = q1;
    else if (c == C_SP && (SPL_BIT(c1) & M_OTHER)) start = q1 + l1;
    if (start >= 0) {
        int e = run_end_m(m, MK_O, start);
        e = run_end_m(m, MK_NL, e);                      // [\r\n]* (no-op when the next char is no newline)
        if (mistral) {                                   // [\r\n/]*: newline runs and '/' in turn (rare)
            while (e < m.wbits() && !bit_m(m, MK_TS, e) && m.txt(e) == '/') {
                do e++; while (e < m.wbits() && !bit_m(m, MK_TS, e) && m.txt(e) == '/');
                e = run_end_m(m, MK_NL, e);
            }
        }
        SPL_RUN_OR_DEFER(e);
        return e;
    }
    // whitespace: maximal \s run [p, r)
    const int r = run_end_m(m, MK_S, q1);
    SPL_RUN_OR_DEFER(r);
    const int nl = last_bit_in_m(m, MK_NL, p, r);
    if (nl >= 0) return nl + 1;                          // \s*[\r\n]+ : through the LAST newline
    const bool eot = r >= m.wbits() || bit_m(m, MK_TS, r);
    if (eot) return r;                                   // \s+(?!\S) at end of text
    const int lc = last_bit_in_m(m, MK_CS, p, r);        // start of the run's last character
    return lc > p ? lc : r;
}

template <class MA> SPL_HD int match_end_cl100k_m(const MA& m, int p) {
    const uint32_t r0 = m.rec(p);
    const uint32_t c = r0 & CB_CLASS;
    const int q1 = p + (int)(r0 >> CB_LEN_SHIFT) + 1;
    int l1;
    const uint32_t c1 = peek_m(m, q1, l1);
    if (c1 == C_WEND) return SPL_DEFER;
    if (c == C_AP && (SPL_BIT(c1) & M_L)) {
        const int e = contraction(m, p);
        if (e != 0) return e;
    }
    // The lanes of a wavefront take different alternatives.  Letters, "other" runs and whitespace
    // all begin with ONE run search that differs only in the mask and the start, so that search is
    // shared (one inlined loop for every lane) and only the short tails stay divergent.
    const uint32_t cb = SPL_BIT(c), c1b = SPL_BIT(c1);
    const bool letters = (cb & M_L) || ((cb & M_X) && (c1b & M_L));
    const bool other = !letters && ((cb & M_OTHER) || (c == C_SP && (c1b & M_OTHER)));
    const bool space = !letters && !other && c != C_N;            // whitespace: maximal \s run from q1
    if (c == C_N && !letters) {
        if (c1 != C_N) return q1;
        const int q2 = q1 + l1;
        int l2;
        const uint32_t c2 = peek_m(m, q2, l2);
        if (c2 == C_WEND) return SPL_DEFER;
        return c2 == C_N ? q2 + l2 : q2;
    }
    const int which = letters ? MK_L : other ? MK_O : MK_S;
    const int start = letters ? ((cb & M_L) ? q1 : q1 + l1) : other ? ((cb & M_OTHER) ? q1 : q1 + l1) : q1;
    int e = run_end_m(m, which, start);
    if (letters) {
        SPL_RUN_OR_DEFER(e);
        return e;
    }
    if (other) {
        e = run_end_m(m, MK_NL, e);                      // [\r\n]* (no-op when the next char is no newline)
        SPL_RUN_OR_DEFER(e);
        return e;
    }
    (void)space;
    const int r = e;                                     // maximal \s run [p, r)
    SPL_RUN_OR_DEFER(r);
    const int nl = last_bit_in_m(m, MK_NL, p, r);
    if (nl >= 0) return nl + 1;                          // \s*[\r\n]+ : through the LAST newline
    const bool eot = r >= m.wbits() || bit_m(m, MK_TS, r);
    if (eot) return r;                                   // \s+(?!\S) at end of text
    const int lc = last_bit_in_m(m, MK_CS, p, r);        // start of the run's last character
    return lc > p ? lc : r;
}

// o200k: letter bodies keep the character-wise closed form of spl_scan.h (case structure inside
// a run is not a run-end query); everything else goes through the masks.
template <class MA> SPL_HD int match_end_o200k_m(const MA& m, int p, bool mistral = false) {
    const uint32_t r0 = m.rec(p);
    const uint32_t c = r0 & CB_CLASS;
    const int l0 = (int)(r0 >> CB_LEN_SHIFT) + 1;
    const int q1 = p + l0;
    int l1;
    const uint32_t c1 = peek_m(m, q1, l1);
    if (c1 == C_WEND) return SPL_DEFER;
    constexpr uint32_t M_LM = M_L | SPL_BIT(C_M);
    if (SPL_BIT(c) & (M_X & ~SPL_BIT(C_M))) {
        if (SPL_BIT(c1) & M_LM) {
            const int e = letters_o200k(m, q1, c1, l1, true);
            if (e == SPL_DEFER) return e;
            if (e > 0) return mistral ? e : with_contraction(m, e);
        }
    } else if (c == C_M) {
        if (SPL_BIT(c1) & M_LM) {
            const int e = letters_o200k(m, q1, c1, l1, false);
            if (e == SPL_DEFER) return e;
            if (e > 0) return mistral ? e : with_contraction(m, e);
        }
        const int e = letters_o200k(m, p, c, l0, false);
        if (e == SPL_DEFER) return e;
        return mistral ? e : with_contraction(m, e);
    } else if (SPL_BIT(c) & M_L) {
        const int e = letters_o200k(m, p, c, l0, true);
        if (e == SPL_DEFER) return e;
        return mistral ? e : with_contraction(m, e);
    }
    return match_tail_m(m, p, c, q1, c1, l1, mistral);
}

template <class MA> SPL_HD int match_end_m(const MA& m, int p, int pattern) {
    return pattern == PAT_CL100K ? match_end_cl100k_m(m, p) : match_end_o200k_m(m, p, pattern == PAT_MISTRAL_V3);
}

// One word of the sync-point mask from the kind masks (same rules as is_sync):
//   cur/prev = this word / the masks shifted by one position (carry = top bit of the previous word).
// kw[k] = word w of mask k, kp[k] = word w-1 of mask k (0 for w == 0).
SPL_HD uint32_t sync_word(int pattern, const uint32_t (&kw)[MK_COUNT], const uint32_t (&kp)[MK_COUNT]) {
    auto prev = [&](int k) { return (kw[k] << 1) | (kp[k] >> 31); };
    const uint32_t L = kw[MK_L], N = kw[MK_N], S = kw[MK_S], NL = kw[MK_NL];
    const uint32_t pL = prev(MK_L), pN = prev(MK_N), pNL = prev(MK_NL), pO = prev(MK_O), pM = prev(MK_M);
    uint32_t sy;
    if (pattern == PAT_CL100K) {
        sy = (pL & ~L) | (pN & ~N) | (pNL & (L | N)) | (pO & (S & ~NL)) | (N & ~pN) | (pNL & kw[MK_O]);
    } else if (pattern == PAT_O200K) {
        sy = (pL & ~(L | kw[MK_M] | kw[MK_AP])) | (pN & ~N) | (pNL & (L | N)) | ((pO & ~pM) & (S & ~NL)) | (N & ~pN);
    } else {     // mistral_v3: no contraction suffix behind letters; every number is a match of its own
        sy = (pL & ~(L | kw[MK_M])) | pN | N | (pNL & L) | ((pO & ~pM) & (S & ~NL));
    }
    // only real character starts can be sync points; a text start always is one
    const uint32_t real = kw[MK_CS];
    return (sy | kw[MK_TS]) & real;
}

}  // namespace spl
