// spl_scan_words.h -- class records AND class bitmasks of a window in ONE pass, four bytes per lane.
//
// Up to round 3 the tile kernel classified the window (one record per byte, four bytes per lane, into LDS), crossed a
// barrier, and then read the records back one byte per lane to condense them into the kind masks of spl_scan_masks.h
// with a dozen ballots per 64-byte row: a fifth of the kernel's vector instructions for the two steps
// (profiles/r04_phase_instruction_mix.txt).  Here a lane turns its four bytes into the records AND into two words of
// NIBBLES, one nibble per kind: bit k of nibble j = "byte k of this lane's word is of kind j".  For ASCII -- nearly every
// word of English / code -- that is four reads of a 128-entry LDS table and a few shifts; only a word with a byte beyond
// ASCII (or at a text edge) takes the general path below.  Eight neighbouring lanes then transpose their 8 x 8 nibbles
// with three DPP exchanges (k_pretok), after which lane 8w + j holds mask word w of kind j, position-ordered as before.
// The split semantics are untouched (same records, same masks: tests/hostsim checks this header against the
// record-based mask builder on adversarial and malformed text); what it restates is still RegexBackend::find_iter over
// CL100K_BASE_PATTERN / O200K_BASE_PATTERN / MISTRAL_V3_PATTERN, reference src/core/tokenizer.rs:39, :42, :64, 244-257.
#pragma once
#include "spl_scan.h"
#include "spl_scan_masks.h"

namespace spl {

// The kinds of the two nibble words, nibble j of V0 / V1 (MK_* of spl_scan_masks.h):
//   V0: L N S NL O AP SP UP        V1: CS BAD M LB SL TS (two nibbles spare)
constexpr uint32_t V0_KINDS = 0x87643210u;     // nibble j = MK id: MK_L 0, MK_N 1, MK_S 2, MK_NL 3, MK_O 4, MK_AP 6, MK_SP 7, MK_UP 8
constexpr uint32_t V1_KINDS = 0x00DA95BCu;     // MK_CS 12, MK_BAD 11, MK_M 5, MK_LB 9, MK_SL 10, MK_TS 13
static_assert(MK_L == 0 && MK_N == 1 && MK_S == 2 && MK_NL == 3 && MK_O == 4 && MK_AP == 6 && MK_SP == 7 && MK_UP == 8, "V0_KINDS");
static_assert(MK_CS == 12 && MK_BAD == 11 && MK_M == 5 && MK_LB == 9 && MK_SL == 10 && MK_TS == 13, "V1_KINDS");
constexpr int V1_NKINDS = 6;
constexpr uint32_t V1_CS = 1u, V1_BAD = 1u << 4, V1_M = 1u << 8, V1_LB = 1u << 12, V1_SL = 1u << 16, V1_TS = 1u << 20;

// What one class contributes to the nibble words of a byte whose CHARACTER is of that class (bit 0 of each nibble;
// the byte's index in its word shifts it): x for V0, y for V1 (the kinds that depend on the class alone: M, LB).
struct KindEnt { uint32_t x, y; };
SPL_HD KindEnt kind_entry(uint32_t cls) {
    KindEnt e{0u, 0u};
    if (cls >= C_EOT) return e;
    const uint32_t oh = SPL_BIT(cls);
    e.x = ((oh & kind_classes(MK_L)) ? 1u : 0u) | ((oh & kind_classes(MK_N)) ? 1u << 4 : 0u) | ((oh & kind_classes(MK_S)) ? 1u << 8 : 0u) |
          ((oh & kind_classes(MK_NL)) ? 1u << 12 : 0u) | ((oh & kind_classes(MK_O)) ? 1u << 16 : 0u) | ((oh & kind_classes(MK_AP)) ? 1u << 20 : 0u) |
          ((oh & kind_classes(MK_SP)) ? 1u << 24 : 0u) | ((oh & kind_classes(MK_UP)) ? 1u << 28 : 0u);
    e.y = ((oh & kind_classes(MK_M)) ? V1_M : 0u) | ((oh & kind_classes(MK_LB)) ? V1_LB : 0u);
    return e;
}
// Entry of an ASCII byte c of class cls inside a text: its V0 contribution; its V1 contribution (a character start;
// '/' for mistral's [\r\n/]*; a mark is "bad" outside cl100k -- no ASCII byte is one, kept for uniformity) with the
// class code -- the byte's whole record -- in the top nibble.
SPL_HD KindEnt ascii_entry(int pattern, uint32_t c, uint32_t cls) {
    KindEnt e = kind_entry(cls);
    e.y |= V1_CS | ((pattern == PAT_MISTRAL_V3 && c == '/') ? V1_SL : 0u) | ((pattern != PAT_CL100K && cls == C_M) ? V1_BAD : 0u) | (cls << 28);
    return e;
}

struct WordKinds { uint32_t rec, v0, v1; };

// the four flag bits x (bit k) spread to the bytes of a word as CB_TSTART | CB_SYNC
SPL_HD uint32_t spread_ts(uint32_t x4) { return ((x4 * 0x00204081u) & 0x01010101u) * (CB_TSTART | CB_SYNC); }

// A word of four ASCII bytes, all of them text (inside the text, no special-literal span): e[k] = the table entry of byte k.
SPL_HD WordKinds classify_word_ascii(const KindEnt (&e)[4], uint32_t ts4) {
    WordKinds o;
    o.v0 = e[0].x | (e[1].x << 1) | (e[2].x << 2) | (e[3].x << 3);
    o.v1 = ((e[0].y | (e[1].y << 1) | (e[2].y << 2) | (e[3].y << 3)) & 0x00FFFFFFu) | (ts4 << 20);
    o.rec = ((e[0].y >> 28) | ((e[1].y >> 28) << 8) | ((e[2].y >> 28) << 16) | ((e[3].y >> 28) << 24)) | spread_ts(ts4);
    return o;
}

// bytes sh .. sh + 3 of the eight bytes lo (first), hi
SPL_HD uint32_t bytes_at(uint32_t lo, uint32_t hi, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (sh & 3u)));
#endif
}

// A lead byte and the three bytes behind it (w: the lead in byte 0), without the class yet -- byte_record (spl_scan.h)
// split in two, so that the table lookups of a word's characters can go out TOGETHER, and on VALUES instead of an
// accessor: indexing a three-word register window by a run-time position sent it to scratch memory, a dependent load per
// byte looked at.  ts3: bit m - 1 = "a text starts at the byte m behind the lead"; q = the lead's window index, hi = end of
// the staged text.  The lead takes the continuation bytes that follow it in the same text, at most as many as it
// announces; need: all of them present -- the class is cp_class(cp); otherwise the bytes it got are one character "other".
struct ByteDec { uint32_t rec, cp; bool need; };
SPL_HD ByteDec lead_decode(uint32_t w, uint32_t ts3, int q, int hi) {
    const uint32_t c0 = w & 0xFFu, b1 = (w >> 8) & 0xFFu, b2 = (w >> 16) & 0xFFu, b3 = w >> 24;
    const uint32_t want = utf8_len(c0);
    const bool ok1 = want > 1u && q + 1 < hi && !(ts3 & 1u) && (b1 & 0xC0u) == 0x80u;
    const bool ok2 = ok1 && want > 2u && q + 2 < hi && !(ts3 & 2u) && (b2 & 0xC0u) == 0x80u;
    const bool ok3 = ok2 && want > 3u && q + 3 < hi && !(ts3 & 4u) && (b3 & 0xC0u) == 0x80u;
    const uint32_t len = 1u + (ok1 ? 1u : 0u) + (ok2 ? 1u : 0u) + (ok3 ? 1u : 0u);
    ByteDec d;
    d.rec = (len - 1u) << CB_LEN_SHIFT;                 // class to be filled in (C_P == 0 if the character is incomplete)
    d.need = len == want;
    d.cp = want == 2u ? ((c0 & 0x1Fu) << 6) | (b1 & 0x3Fu)
         : want == 3u ? ((c0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu)
                      : ((c0 & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
    return d;
}
// A byte of 0x80..0xBF at window index q: how far back is the lead byte of its text that reaches it (1..3), 0 if none
// does (the byte is then a character of class "other" by itself).  back: the three bytes before it, the nearest in byte 0;
// tsb: bit m = "a text starts at q - m" (m = 0..2); lo: first window index that exists.
SPL_HD uint32_t cont_lead_dist(uint32_t back, uint32_t tsb, int q, int lo) {
#pragma unroll
    for (uint32_t m = 1; m <= 3; m++) {
        if (((tsb >> (m - 1u)) & 1u) || q - (int)m < lo) return 0u;
        const uint32_t b = (back >> (8u * (m - 1u))) & 0xFFu;
        if (b >= 0xC0u) return utf8_len(b) > m ? m : 0u;
        if (b < 0x80u) return 0u;
    }
    return 0u;
}

// The general path: any word.  wp, tw, wn: the words before, of and behind the lane's (window bytes [i0 - 4, i0 + 8));
// ts16: bit d = "a text starts at window index i0 - 4 + d"; kent(c): KindEnt of class c (a 16-entry table); ascii(c): class of
// an ASCII byte.
//   i0       window index of the word's first byte
//   ts4, sk4 text-start / special-literal-span bits of the four bytes
//   iB       first window index past the text; Wv the window size; lo the first window index that exists (bytes before the
//            corpus: w0 < 0); iT end of the staged text
// Records exactly as the round-3 classify step made them (byte_record); kinds exactly as its mask step: a continuation
// byte has the kinds of its character (its lead byte's class), its own CS / TS / BAD.
template <class KENT, class ASC>
SPL_HD WordKinds classify_word(const DeviceTables& T, int pattern, uint32_t wp, uint32_t tw, uint32_t wn, uint32_t ts16, const KENT& kent,
                               const ASC& ascii, uint32_t ts4, uint32_t sk4, int i0, int iB, int Wv, int lo, int iT) {
    // (1) everything that needs no table: slot k = byte k of the word, slot 4 = the lead byte of the character the word
    //     begins inside (one of the three bytes before it), if any
    uint32_t rec[5], cp[5];
    bool need[5], text[4];
    uint32_t dist0 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k;
        const uint32_t c0 = (tw >> (8 * k)) & 0xFFu;
        need[k] = false; cp[k] = 0u; text[k] = false;
        if (i >= iB) rec[k] = (i == iB && iB < Wv) ? (uint32_t)(C_EOT | CB_TSTART | CB_SYNC) : (uint32_t)C_WEND;
        else if (i < lo) rec[k] = C_CONT;
        else if ((sk4 >> k) & 1u) rec[k] = C_EOT | CB_TSTART;
        else {
            text[k] = true;
            if (c0 < 0x80u) rec[k] = ascii(c0);
            else if (c0 < 0xC0u) {
                // the three bytes before byte k, nearest first: bytes k + 3, k + 2, k + 1 of (wp, tw); text starts at i, i - 1, i - 2
                const uint32_t fwd = k == 3 ? tw : bytes_at(wp, tw, (uint32_t)k + 1u);   // bytes i - 3, i - 2, i - 1 in bytes 0, 1, 2
                const uint32_t back = ((fwd >> 16) & 0xFFu) | (fwd & 0xFF00u) | ((fwd & 0xFFu) << 16);
                const uint32_t t3 = (ts16 >> (2 + k)) & 7u;                        // text starts at i - 2, i - 1, i
                const uint32_t tsb = ((t3 >> 2) & 1u) | (t3 & 2u) | ((t3 & 1u) << 2);
                const uint32_t dist = cont_lead_dist(back, tsb, i, lo);
                rec[k] = dist ? (uint32_t)C_CONT : (uint32_t)C_P;
                if (k == 0) dist0 = dist;
            } else {
                const ByteDec d = lead_decode(bytes_at(tw, wn, (uint32_t)k), (ts16 >> (5 + k)) & 7u, i, iT);
                rec[k] = d.rec; cp[k] = d.cp; need[k] = d.need;
            }
        }
    }
    rec[4] = C_CONT; cp[4] = 0u; need[4] = false;
    if (dist0) {                                         // the word begins inside a character: its lead, dist0 bytes back
        const ByteDec d = lead_decode(bytes_at(wp, tw, 4u - dist0), (ts16 >> (5u - dist0)) & 7u, i0 - (int)dist0, iT);
        rec[4] = d.rec; cp[4] = d.cp; need[4] = d.need;
    }
    // (2) the classes of the characters that are whole: all lookups of the word in flight together (two dependent loads)
    uint32_t cls[5];
    bool lk[5], any = false;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const bool cjk = T.cjk_fast && ((cp[j] - 0x4E00u) < 0x5200u || (cp[j] - 0xAC00u) < 0x2BA4u);
        cls[j] = cjk ? (uint32_t)C_LO : (uint32_t)C_P;                     // (cp >= 0x110000: "other")
        lk[j] = need[j] && !cjk && cp[j] < 0x110000u;
        any = any || lk[j];
    }
    if (any) {
        const uint32_t sh = T.ucls_shift, lowm = (1u << sh) - 1u;
        uint32_t blk[5];
#pragma unroll
        for (int j = 0; j < 5; j++) blk[j] = T.ucls_stage1[lk[j] ? cp[j] >> sh : 0u];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const uint32_t c = T.ucls_stage2[(blk[j] << sh) | (lk[j] ? cp[j] & lowm : 0u)];
            cls[j] = lk[j] ? c : cls[j];
        }
    }
    // (3) records and kind nibbles
    WordKinds o{0u, 0u, 0u};
    uint32_t cur_kc = !dist0 ? (uint32_t)C_CONT : need[4] ? cls[4] : (rec[4] & CB_CLASS);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k;
        const uint32_t c0 = (tw >> (8 * k)) & 0xFFu;
        uint32_t r = need[k] ? (rec[k] | cls[k]) : rec[k];
        if (text[k] && ((ts4 >> k) & 1u)) r |= CB_TSTART | CB_SYNC;
        const uint32_t c = r & CB_CLASS;
        const uint32_t kc = c == C_CONT ? cur_kc : c;
        cur_kc = kc;
        const KindEnt e = kent(kc);
        const bool slash = pattern == PAT_MISTRAL_V3 && c0 == '/';
        const uint32_t y = e.y | (c < C_EOT ? V1_CS : 0u) | (bad_for_starts(pattern, r, kc, i < iB) ? V1_BAD : 0u) |
                           (slash ? V1_SL : 0u) | ((r & CB_TSTART) ? V1_TS : 0u);
        o.v0 |= e.x << k;
        o.v1 |= y << k;
        o.rec |= r << (8 * k);
    }
    return o;
}

// the four byte flags x & 0x80808080 as a 4-bit mask (bit k = byte k)
SPL_HD uint32_t flags4(uint32_t x) { return (((x >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu; }
// a 4-bit mask spread to the bytes of a word: bit k -> 0x01 in byte k
SPL_HD uint32_t spread4(uint32_t m4) { return (m4 * 0x00204081u) & 0x01010101u; }

// The COMMON word beyond ASCII: well-formed UTF-8, away from every edge -- an accented letter, a dash, a run of CJK.  The
// general path above decides each of its five byte slots on its own (every one a lead byte? a continuation byte of what?),
// nine hundred instructions that every wavefront with ONE such word walks through; here the word is taken as what it
// nearly always is: ASCII bytes from the table, plus at most three whole characters -- the one the word begins inside
// (A) and up to two that start in it (B, C).  Anything else -- a text start or an edge within three bytes, a stray
// continuation byte, an incomplete character -- returns false, and the caller takes the general path: same results,
// checked against each other on every window of the host simulation's corpora (tests/hostsim).
//   precondition (caller): no special-literal span in the word, i0 + 3 < iB, i0 >= lo
//   aent(b): the ASCII table entry of byte b < 0x80 (ascii_entry); kent(c): kind_entry of class c
template <class AENT, class KENT>
SPL_HD bool classify_word_text(const DeviceTables& T, int pattern, uint32_t wp, uint32_t tw, uint32_t wn, uint32_t ts16, const AENT& aent,
                               const KENT& kent, int i0, int lo, int iT, WordKinds& o) {
    if (((ts16 >> 1) & 0x3FFu) != 0u || i0 - 3 < lo || i0 + 7 > iT) return false;    // window indices i0 - 3 .. i0 + 6: one text, all staged
    const uint32_t hi_t = tw & 0x80808080u, lead_t = tw & (tw << 1) & 0x80808080u;
    const uint32_t non4 = flags4(hi_t), lead4 = flags4(lead_t);
    const uint32_t lead_p = flags4(wp & (wp << 1) & 0x80808080u), cont_p = flags4(wp & ~(wp << 1) & 0x80808080u);
    // the (at most three) characters: first own byte offset (A: negative), the word that holds the lead in byte 0
    uint32_t dist = 0;                                          // A: lead `dist` bytes before the word
    if (non4 & ~lead4 & 1u) {
        dist = (lead_p & 8u) ? 1u : ((cont_p & 8u) && (lead_p & 4u)) ? 2u : ((cont_p & 12u) == 12u && (lead_p & 2u)) ? 3u : 0u;
        if (!dist) return false;
    }
    const uint32_t k1 = lead4 ? (uint32_t)ctz32(lead4) : 0u;
    const uint32_t rest = lead4 & (lead4 - 1u);
    const uint32_t k2 = rest ? (uint32_t)ctz32(rest) : 0u;
    if (rest & (rest - 1u)) return false;                       // three lead bytes in four: no well-formed text
    const bool hasA = dist != 0u, hasB = lead4 != 0u, hasC = rest != 0u;
    const ByteDec dA = lead_decode(bytes_at(wp, tw, 4u - dist), 0u, 0, 1 << 20);
    const ByteDec dB = lead_decode(bytes_at(tw, wn, k1), 0u, 0, 1 << 20);
    const ByteDec dC = lead_decode(bytes_at(tw, wn, k2), 0u, 0, 1 << 20);
    if ((hasA && !dA.need) || (hasB && !dB.need) || (hasC && !dC.need)) return false;
    const uint32_t lenA = (dA.rec >> CB_LEN_SHIFT) + 1u, lenB = (dB.rec >> CB_LEN_SHIFT) + 1u, lenC = (dC.rec >> CB_LEN_SHIFT) + 1u;
    if (hasA && lenA <= dist) return false;                     // (the lead before the word does not reach it)
    const uint32_t mA = hasA ? ((1u << (lenA - dist)) - 1u) & 0xFu : 0u;
    const uint32_t mB = hasB ? (((1u << lenB) - 1u) << k1) & 0xFu : 0u;
    const uint32_t mC = hasC ? (((1u << lenC) - 1u) << k2) & 0xFu : 0u;
    if ((mA | mB | mC) != non4 || (mA & mB) || (mB & mC) || (mA & mC)) return false;     // every byte beyond ASCII belongs to exactly one of them
    // their classes: the lookups of the word in flight together
    uint32_t cls[3];
    const uint32_t cp[3] = {dA.cp, dB.cp, dC.cp};
    const bool has[3] = {hasA, hasB, hasC};
    bool lk[3], any = false;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const bool cjk = T.cjk_fast && ((cp[j] - 0x4E00u) < 0x5200u || (cp[j] - 0xAC00u) < 0x2BA4u);
        cls[j] = cjk ? (uint32_t)C_LO : (uint32_t)C_P;
        lk[j] = has[j] && !cjk && cp[j] < 0x110000u;
        any = any || lk[j];
    }
    if (any) {
        const uint32_t sh = T.ucls_shift, lowm = (1u << sh) - 1u;
        uint32_t blk[3];
#pragma unroll
        for (int j = 0; j < 3; j++) blk[j] = T.ucls_stage1[lk[j] ? cp[j] >> sh : 0u];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t c = T.ucls_stage2[(blk[j] << sh) | (lk[j] ? cp[j] & lowm : 0u)];
            cls[j] = lk[j] ? c : cls[j];
        }
    }
    // the ASCII bytes from the table (the other bytes' entries masked out)
    const KindEnt e0 = aent(tw & 0x7Fu), e1 = aent((tw >> 8) & 0x7Fu), e2 = aent((tw >> 16) & 0x7Fu), e3 = aent((tw >> 24) & 0x7Fu);
    const uint32_t asc4 = ~non4 & 0xFu, ascn = asc4 * 0x11111111u, ascb = spread4(asc4) * 0xFFu;
    o.v0 = (e0.x | (e1.x << 1) | (e2.x << 2) | (e3.x << 3)) & ascn;
    o.v1 = (e0.y | (e1.y << 1) | (e2.y << 2) | (e3.y << 3)) & ascn & 0x00FFFFFFu;
    o.rec = ((e0.y >> 28) | ((e1.y >> 28) << 8) | ((e2.y >> 28) << 16) | ((e3.y >> 28) << 24)) & ascb;
    // the characters: every byte the kinds of its character; BAD for all bytes of a multi-byte number / whitespace / (o200k
    // family) mark; CS and the record's class and length on the lead byte, C_CONT on the others
    const uint32_t mm[3] = {mA, mB, mC}, kk[3] = {0u, k1, k2};
    const uint32_t lrec[3] = {0u, dB.rec, dC.rec};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t c = cls[j];
        const KindEnt e = kent(c);
        const bool bad = !(SPL_BIT(c) & (M_L | M_OTHER)) || (pattern != PAT_CL100K && c == C_M);
        o.v0 |= e.x * mm[j];
        o.v1 |= (e.y | (bad ? V1_BAD : 0u)) * mm[j];
        uint32_t r = spread4(mm[j]) * (uint32_t)C_CONT;
        if (j > 0 && has[j]) {
            o.v1 |= V1_CS << kk[j];
            r = (r & ~(0xFFu << (8u * kk[j]))) | ((lrec[j] | c) << (8u * kk[j]));
        }
        o.rec |= r;
    }
    return true;
}

}  // namespace spl
