// spl_scan_starts.h -- ALL match starts of CL100K_BASE_PATTERN in a window of ASCII text by bit-vector
// arithmetic on the class masks: no per-lane chains, no match_end calls.
//
// Replaces the chain phase of k_pretok (spl_scan_masks.h: one chain per sync point, each lane walking
// its matches one after the other) for the windows it applies to; the same RegexBackend::find_iter
// (reference src/core/tokenizer.rs:244-257, pattern :39) is what it restates.  A window qualifies if
// it is ASCII throughout (then a character is a byte and the masks need no inheritance), holds no
// special-token span, and the pattern is cl100k; any other window keeps the chains.
//
// With p(X) = "the previous byte of the same text is in X" and n(X) = "the next byte of the same
// text is in X", the starts are (derivation in DESIGN.md 4.1a):
//   text starts                      TS
//   letters   first of a letter run, unless a one-character prefix joins it:
//             Lf & (p(NL) | p(N) | p(O) & (pp(O) | pp(SP)))          (Lf = L & ~p(L))
//             (a prefix joins when it starts a match itself: any non-newline whitespace does -- it is
//              the last whitespace before a non-space --, an "other" character does iff it is alone
//              and not behind U+0020)
//   numbers   first of a number run and every third one behind it
//   other     first of an "other" run unless behind U+0020 (which then starts the match: " ?")
//             the newlines directly behind an "other" run belong to it ([\r\n]*): ONL, not starts
//   space     (runs of whitespace that is not ONL)  the first; the one behind the LAST newline of the
//             run; and the last one if at least two non-newline characters stand behind the last
//             newline and the text goes on (\s+(?!\S) gives back its last character)
//   contractions  an apostrophe that starts a match and is followed by s t m d / re ve ll (either case)
//             ends its match behind them: a start there (found from the text bytes at the few
//             such apostrophes, `contraction_ends`)
// Everything is exact from a context-free match start (sync point) on, and nothing before a sync point
// depends on what follows it -- which is what lets a tile use only its window.
//
// BV is a window-wide bit vector: one 32-bit word per lane on the device (shifts fetch the neighbour
// lane's word), a vector of words in tests/hostsim.  Required: operator& | ~, shl1(), shr1(), any().
#pragma once
#include "spl_common.h"

namespace spl {

template <class BV> struct Cl100kStartMasks {
    BV L, N, S, NL, O, AP, SP, TS;      // class masks of the window (SP = U+0020 only; S = all whitespace)
};

// Returns the start mask B; CA = apostrophes that start a match (candidates for a contraction).
// `max_iter` bounds the three propagation loops (number thirds, newlines behind "other", "a newline
// follows in this run"); the return value of `ok` tells whether they all converged within it.
template <class BV>
SPL_HD BV cl100k_starts(const Cl100kStartMasks<BV>& m, BV& CA, bool& ok, int max_iter = 64) {
    const BV nTS = ~m.TS;
    auto p = [&](const BV& x) { return x.shl1() & nTS; };                 // previous byte, same text
    auto n = [&](const BV& x) { return (x & nTS).shr1(); };               // next byte, same text
    ok = true;
    const BV pL = p(m.L), pN = p(m.N), pO = p(m.O), pSP = p(m.SP), pNL = p(m.NL);
    // letters
    const BV Lf = m.L & ~pL;
    const BV BL = Lf & (pNL | pN | (pO & (p(pO) | p(pSP))));
    // numbers: run starts, then every third
    const BV Nf = m.N & ~pN;
    const BV N3 = m.N & pN & p(pN);                                        // bytes i-2 .. i are numbers of one text
    BV BN = Nf, X = Nf;
    for (int it = 0;; it++) {
        X = p(p(p(X))) & N3;
        if (!X.any()) break;
        if (it >= max_iter) { ok = false; break; }
        BN = BN | X;
    }
    // other
    const BV Of = m.O & ~pO;
    const BV BO = Of & ~pSP;
    CA = m.AP & BO;
    // newlines that an "other" run takes with it
    BV ONL = m.NL & pO, Y = ONL;
    for (int it = 0;; it++) {
        Y = m.NL & p(Y) & ~ONL;
        if (!Y.any()) break;
        if (it >= max_iter) { ok = false; break; }
        ONL = ONL | Y;
    }
    // whitespace runs (without those newlines)
    const BV S1 = m.S & ~ONL, NL1 = m.NL & S1;
    const BV Sf = S1 & ~p(S1);
    auto nS = [&](const BV& x) { return n(x) & S1; };                      // x holds for the next byte, which is of the same run
    BV H = nS(NL1);                                                        // a newline follows in this run
    for (int it = 0;; it++) {
        const BV H2 = nS(H) & ~H;
        if (!H2.any()) break;
        if (it >= max_iter) { ok = false; break; }
        H = H | H2;
    }
    const BV NLlast = NL1 & ~H;
    const BV BS2 = S1 & p(NLlast);
    const BV Sl = S1 & ~n(S1);
    const BV BS3 = Sl & ~NL1 & p(S1 & ~NL1) & ~m.TS.shr1();
    return m.TS | BL | BN | BO | Sf | BS2 | BS3;
}

// End of the contraction that starts at the apostrophe `ap` (text bytes through txt(i), `is_ts(i)` = a
// text starts at i, `end` = first position past the readable window), or 0 if there is none:
// (?i:'s|'t|'re|'ve|'m|'ll|'d) on ASCII letters.
template <class TXT, class TSF>
SPL_HD int cl100k_contraction_end(const TXT& txt, const TSF& is_ts, int ap, int end) {
    const int q1 = ap + 1;
    if (q1 >= end || is_ts(q1)) return 0;
    const uint32_t a = txt(q1) | 0x20u;
    if (a == 's' || a == 't' || a == 'm' || a == 'd') return q1 + 1;
    if (a == 'r' || a == 'v' || a == 'l') {
        const int q2 = q1 + 1;
        if (q2 >= end || is_ts(q2)) return 0;
        const uint32_t b = txt(q2) | 0x20u;
        return b == (a == 'l' ? (uint32_t)'l' : (uint32_t)'e') ? q2 + 1 : 0;
    }
    return 0;
}

}  // namespace spl
