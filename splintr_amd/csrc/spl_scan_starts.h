// spl_scan_starts.h -- ALL match starts of CL100K_BASE_PATTERN in a window of ASCII text by bit-vector
// arithmetic on the class masks: no per-lane chains, no match_end calls.
//
// Replaces the chain phase of k_pretok (spl_scan_masks.h: one chain per sync point, each lane walking
// its matches one after the other) for the windows it applies to; the same RegexBackend::find_iter
// (reference src/core/tokenizer.rs:244-257, pattern :39) is what it restates.  A window qualifies if
// its numbers and whitespace are single bytes (letters and "other" characters may be of any width:
// continuation bytes inherit their lead's kind) and it holds no special-token span (bad_for_starts,
// spl_scan_masks.h); any other window keeps the chains.  The o200k family: further down.
//
// With p(X) = "the previous byte of the same text is in X" and n(X) = "the next byte of the same
// text is in X", the starts are (derivation in DESIGN.md 4.1a):
//   text starts                      TS
//   letters   first of a letter run, unless a one-character prefix joins it:
//             Lf & (p(NL) | p(N) | p(O & ~BOx))      (Lf = L & ~p(L); BOx: "other" characters that start a match, whole)
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
// lane's word), a vector of words in tests/hostsim.  Required: copying, operator& | ~, shl1(), shr1(), any().
#pragma once
#include "spl_scan.h"

namespace spl {

template <class BV> struct Cl100kStartMasks {
    BV L, N, S, NL, O, AP, SP, CS, TS;  // class masks of the window (SP = U+0020 only; S = all whitespace; CS = character starts)
};

// The computation in three independent parts (the kernel gives each to a wavefront of its own; their
// union, with the text starts, is the start mask).  `max_iter` bounds the propagation loops (number
// thirds, newlines behind "other", "a newline follows in this run"); `ok` tells whether they converged.
template <class BV> struct Cl100kShift {
    const BV& TS; BV nTS;
    SPL_HD explicit Cl100kShift(const BV& ts) : TS(ts), nTS(~ts) {}
    SPL_HD BV p(const BV& x) const { return x.shl1() & nTS; }                // previous byte, same text
    SPL_HD BV n(const BV& x) const { return (x & nTS).shr1(); }              // next byte, same text
};

// the characters whose first bytes are in X, with their continuation bytes (K = their kind's mask)
template <class BV>
SPL_HD BV whole_chars(const Cl100kShift<BV>& sh, const BV& X, const BV& K, const BV& CS) {
    BV R = X;
    for (int it = 0; it < 3; it++) R = R | (sh.p(R) & K & ~CS);
    return R;
}

// letters and numbers
template <class BV>
SPL_HD BV cl100k_starts_ln(const Cl100kStartMasks<BV>& m, bool& ok, int max_iter) {
    const Cl100kShift<BV> sh(m.TS);
    ok = true;
    const BV pL = sh.p(m.L), pN = sh.p(m.N), pNL = sh.p(m.NL);
    const BV Lf = m.L & ~pL;
    // (the "other" character before the run joins it iff it starts a match: all of its bytes, then)
    const BV BL = Lf & (pNL | pN | sh.p(m.O & ~whole_chars(sh, m.O & ~sh.p(m.O) & ~sh.p(m.SP), m.O, m.CS)));
    const BV Nf = m.N & ~pN;
    const BV N3 = m.N & pN & sh.p(pN);                                     // bytes i-2 .. i are numbers of one text
    BV BN = Nf, X = Nf;
    for (int it = 0;; it++) {                                              // run starts, then every third
        X = sh.p(sh.p(sh.p(X))) & N3;
        if (!X.any()) break;
        if (it >= max_iter) { ok = false; break; }
        BN = BN | X;
    }
    return BL | BN;
}

// newlines that an "other" run takes with it
template <class BV>
SPL_HD BV cl100k_onl(const Cl100kStartMasks<BV>& m, const Cl100kShift<BV>& sh, bool& ok, int max_iter) {
    BV ONL = m.NL & sh.p(m.O), Y = ONL;
    for (int it = 0;; it++) {
        Y = m.NL & sh.p(Y) & ~ONL;
        if (!Y.any()) break;
        if (it >= max_iter) { ok = false; break; }
        ONL = ONL | Y;
    }
    return ONL;
}

// "other" runs; CA = apostrophes that start a match (candidates for a contraction)
template <class BV>
SPL_HD BV cl100k_starts_o(const Cl100kStartMasks<BV>& m, BV& CA) {
    const Cl100kShift<BV> sh(m.TS);
    const BV BO = m.O & ~sh.p(m.O) & ~sh.p(m.SP);
    CA = m.AP & BO;
    return BO;
}

// whitespace runs (without the newlines behind "other" runs)
template <class BV>
SPL_HD BV cl100k_starts_s(const Cl100kStartMasks<BV>& m, bool& ok, int max_iter) {
    const Cl100kShift<BV> sh(m.TS);
    ok = true;
    const BV ONL = cl100k_onl(m, sh, ok, max_iter);
    const BV S1 = m.S & ~ONL, NL1 = m.NL & S1;
    const BV Sf = S1 & ~sh.p(S1);
    auto nS = [&](const BV& x) { return sh.n(x) & S1; };                   // x holds for the next byte, which is of the same run
    BV H = nS(NL1);                                                        // a newline follows in this run
    for (int it = 0;; it++) {
        const BV H2 = nS(H) & ~H;
        if (!H2.any()) break;
        if (it >= max_iter) { ok = false; break; }
        H = H | H2;
    }
    const BV NLlast = NL1 & ~H;
    const BV BS2 = S1 & sh.p(NLlast);
    const BV Sl = S1 & ~sh.n(S1);
    const BV BS3 = Sl & ~NL1 & sh.p(S1 & ~NL1) & ~m.TS.shr1();
    return Sf | BS2 | BS3;
}

// Returns the start mask B; CA = apostrophes that start a match.
template <class BV>
SPL_HD BV cl100k_starts(const Cl100kStartMasks<BV>& m, BV& CA, bool& ok, int max_iter = 64) {
    bool ok1, ok2;
    const BV a = cl100k_starts_ln(m, ok1, max_iter);
    const BV c = cl100k_starts_s(m, ok2, max_iter);
    ok = ok1 && ok2;
    return m.TS | a | cl100k_starts_o(m, CA) | c;
}

// ---------------------------------------------------------------------------------------------------------
// O200K_BASE_PATTERN / MISTRAL_V3_PATTERN (reference src/core/tokenizer.rs:42, :64) on the same terms (no marks
// in the window: bad_for_starts; caseless letters -- Lm, Lo -- count as lower case, which is exact unless
// an upper-case letter stands right behind one).  Against cl100k:
//   letters   a letter run is cut where an upper-case letter follows a lower-case one (U* W+ | U+ W*: a
//             match is upper-case letters, then lower-case ones): UP & CS & p(L & ~UP).  The one-character
//             prefix joins iff it starts a match: any non-newline whitespace, or an "other" character in BO.
//   suffix    o200k: letters may take a contraction with them ('s 't 're 've 'm 'll 'd): the apostrophe
//             and the letter behind it then start nothing, the byte behind the contraction does
//             (o200k_contractions, from the text bytes; mistral has no such suffix)
//   other     behind an "other" run the match takes newlines AND '/' ([\r\n/]*): the absorbed bytes A start
//             nothing, an "other" character right behind them starts a match
//   numbers   mistral: every number is a match of its own
template <class BV> struct O200kStartMasks {
    BV L, UP, LB, N, S, NL, O, AP, SP, SL, CS, TS;   // UP = Lu | Lt, LB = Lm | Lo (in both letter sets of the pattern)
};

// bytes that the [\r\n/]* behind an "other" run takes: a newline behind an "other" byte, then newlines and '/'
template <class BV>
SPL_HD BV o200k_absorbed(const O200kStartMasks<BV>& m, const Cl100kShift<BV>& sh, bool slash, bool& ok, int max_iter) {
    const BV Z = slash ? (m.NL | m.SL) : m.NL;
    BV A = m.NL & sh.p(m.O), Y = A;
    for (int it = 0;; it++) {
        Y = Z & sh.p(Y) & ~A;
        if (!Y.any()) break;
        if (it >= max_iter) { ok = false; break; }
        A = A | Y;
    }
    return A;
}
// "other" bytes that start a match (a contraction's apostrophe still among them)
template <class BV>
SPL_HD BV o200k_other_starts(const O200kStartMasks<BV>& m, const Cl100kShift<BV>& sh, const BV& A) {
    return m.O & ~A & (~sh.p(m.O) | sh.p(A)) & ~sh.p(m.SP);
}

// letters and numbers (with the text starts)
template <class BV>
SPL_HD BV o200k_starts_ln(const O200kStartMasks<BV>& m, bool mistral, bool& ok, int max_iter) {
    const Cl100kShift<BV> sh(m.TS);
    ok = true;
    const BV A = o200k_absorbed(m, sh, mistral, ok, max_iter);
    const BV BO = o200k_other_starts(m, sh, A);
    const BV pL = sh.p(m.L), pN = sh.p(m.N);
    const BV Lf = m.L & ~pL;
    const BV BL = Lf & (sh.p(m.NL) | pN | sh.p(m.O & ~whole_chars(sh, BO, m.O, m.CS)));
    const BV BC = m.UP & m.CS & sh.p(m.L & ~m.UP);
    // An upper-case letter right behind a caseless one: whether a match ends between them depends on what
    // follows the whole run of upper-case and caseless letters (U* W+ backs off to its last caseless
    // member unless a lower-case letter follows) -- not decided here: the tile keeps the chains.
    if ((m.UP & m.CS & sh.p(m.LB)).any()) ok = false;
    BV BN = m.N;                                                           // mistral: \p{N}, one at a time
    if (!mistral) {
        const BV Nf = m.N & ~pN;
        const BV N3 = m.N & pN & sh.p(pN);
        BV X = Nf;
        BN = Nf;
        for (int it = 0;; it++) {
            X = sh.p(sh.p(sh.p(X))) & N3;
            if (!X.any()) break;
            if (it >= max_iter) { ok = false; break; }
            BN = BN | X;
        }
    }
    return m.TS | BL | BC | BN;
}

// "other" runs; CAND = apostrophes behind a letter (o200k: candidates for a contraction suffix)
template <class BV>
SPL_HD BV o200k_starts_o(const O200kStartMasks<BV>& m, bool mistral, BV& CAND, bool& ok, int max_iter) {
    const Cl100kShift<BV> sh(m.TS);
    ok = true;
    const BV A = o200k_absorbed(m, sh, mistral, ok, max_iter);
    CAND = m.AP & sh.p(m.L);
    return o200k_other_starts(m, sh, A);
}

// whitespace runs (without the absorbed newlines)
template <class BV>
SPL_HD BV o200k_starts_s(const O200kStartMasks<BV>& m, bool mistral, bool& ok, int max_iter) {
    const Cl100kShift<BV> sh(m.TS);
    ok = true;
    const BV A = o200k_absorbed(m, sh, mistral, ok, max_iter);
    const BV S1 = m.S & ~A, NL1 = m.NL & S1;
    const BV Sf = S1 & ~sh.p(S1);
    auto nS = [&](const BV& x) { return sh.n(x) & S1; };
    BV H = nS(NL1);
    for (int it = 0;; it++) {
        const BV H2 = nS(H) & ~H;
        if (!H2.any()) break;
        if (it >= max_iter) { ok = false; break; }
        H = H | H2;
    }
    const BV NLlast = NL1 & ~H;
    const BV BS2 = S1 & sh.p(NLlast);
    const BV Sl = S1 & ~sh.n(S1);
    const BV BS3 = Sl & ~NL1 & sh.p(S1 & ~NL1) & ~m.TS.shr1();
    return Sf | BS2 | BS3;
}

// One candidate of o200k's contraction suffix: the apostrophe at `ap` stands behind a letter.  end = where
// contraction(a, ap) says the suffix ends (0: none).  Returns false if the candidate itself stands right
// behind another suffix ("it's's": the second apostrophe starts a match -- the caller keeps the chains then).
// acc: rec(q), txt(q) as in spl_scan.h.
template <class A>
SPL_HD bool o200k_contraction_at(const A& acc, int ap, int& end) {
    end = contraction(acc, ap);
    if (end <= 0) { end = 0; return true; }
    for (int back = 2; back <= 3; back++) {                               // a suffix of one or two characters ending here?
        const int q = ap - back;
        if (q < 1) break;
        if ((acc.rec(q) & CB_CLASS) != C_AP) continue;
        if (acc.rec(q + 1) & CB_TSTART) continue;
        const uint32_t pl = acc.rec(q - 1) & CB_CLASS;
        if (acc.rec(q) & CB_TSTART) continue;                              // (no letter of the same text before it)
        if (pl != C_CONT && !(SPL_BIT(pl) & M_L)) continue;               // (a continuation byte: of a letter, in a window that qualifies)
        if (contraction(acc, q) == ap) return false;
    }
    return true;
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
