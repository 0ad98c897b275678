// spl_rx_split.h -- custom split patterns ON THE DEVICE: the host splitter's matcher (spl_regex.cpp), one text position per lane.
//
// Tokenizer::new compiles any pattern and encode() walks find_iter's non-overlapping leftmost-first matches from the start of
// each text (reference src/core/tokenizer.rs:410-456, 244-257, 729-808).  That walk is sequential: where match k + 1 starts is
// where match k ended.  Position-parallel form:
//
//   k_rx_match   every byte position p that starts a character finds where a match anchored there ends, inside p's document, with the
//                host compiler's own output (regex_device_image: instruction list, class sets, first-character filters, the table of
//                the pattern's top-level alternatives -- copied to LDS).  One position per lane; the wavefront walks the ALTERNATIVES
//                in step: a SIMPLE one (a straight line of items where giving characters back cannot help) is evaluated by all lanes
//                at once, greedy item by item -- a run of a tabulated class is a bit scan over the window --; any other runs the SAME
//                backtracking program the host splitter runs, from that alternative's first instruction.
//                nx[p] = how far find_iter would move from p -- the length of the match, or the length of the ONE character it
//                skips when nothing matches there (bit 15: those bytes are dropped).  Then the block of RXB positions composes
//                its hops by pointer doubling: gx[p] = where the walk from p first leaves the block (offset behind the block's
//                end; bit 15: the last hop was a skip).  A block is CLOSED if every one of its positions leaves it at the same
//                place -- true almost everywhere: walks from different positions of ordinary text fall into step within a word
//                or two.  A hop that crosses whole blocks marks them "maybe skipped".
//   k_rx_mark    block b finds where the walk from position 0 enters it: the exit of the nearest closed, not-skipped block in
//                front of it (usually b - 1: one load), carried forward through gx over the blocks in between; repeats the
//                doubling keeping every level, marks the positions of its own stretch of the walk top-down (the node 2^k hops
//                behind a marked node is on the walk), and ORs the result into the two bitmaps k_pretok takes in place of its
//                own scanner: chunk starts and dropped bytes, bit for bit what regex_split_bits leaves.
//
// What the matcher gives up on is REPORTED, never approximated: a match or a look-ahead that reaches RX_REACH bytes beyond its start
// (every position of a run scans to the run's end -- the work is quadratic in the run length, so it is bounded), RX_STEPS matcher
// steps in one attempt, RX_DEPTH entries on the backtracking stack.  Since round 5 per DOCUMENT: the 256-byte block of such a position goes
// on a list, with the first and last such position in it (k_rx_mark leaves it in pinned host memory), the caller splits the documents those stretches touch on the host cores and
// patches their stretch of the two bitmaps (k_rx_patch) -- a walk that went wrong inside a document still lands on the next document's
// start, because no hop crosses the end of its document.  Only a list that overflows (RX_BAD_CAP blocks: a pattern that gives up
// everywhere) or RX_MAX_OPEN blocks in a row whose walks never fall into step set the status word: the whole batch then goes to the host.
// With SPL_WITH_SPECIAL the literals come from the GPU's own scan (k_special_scan / the general matcher, spl_k_special.h: text-start and
// token bitmaps); a literal is a position whose hop is its length, dropped, with the text ending in front of it and beginning anew behind it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "spl_common.h"
#include "spl_regex.h"

namespace spl {

constexpr int RXB = 256;                       // positions per block (and threads of k_rx_mark)
constexpr int RXT = 256;                       // threads of k_rx_match: a lane per position
constexpr int RX_REACH = 1032;                 // bytes behind its start a match attempt may look at (a hop is at most RX_REACH - 8)
constexpr int RX_BACK = 4;                     // staged bytes in front of the block (\b looks at the character before)
constexpr int RX_LDS_TEXT = RX_BACK + RXB + 276;   // staged text; beyond it the matcher reads global memory
constexpr int RX_TAB = RX_LDS_TEXT - 4;         // window positions whose characters are tabulated (a character reads up to 3 bytes on)
constexpr int RX_BMW = (RX_TAB + 63) / 64 * 2;  // words of one window bitmap
constexpr int RX_DEPTH = 10;                     // entries of a lane's backtracking stack (LDS: two words each, RXT lanes)
constexpr uint32_t RX_STEPS = 8192;
constexpr int RX_MAX_OPEN = 1024;                // blocks in a row that do not close (k_rx_mark carries the walk through them one by one)
constexpr uint32_t RX_FAIL = 0xFFFFFFFFu, RX_ABORT = 0xFFFFFFFEu, RX_NOK = 0xFFFFFFFFu;
constexpr uint32_t RX_BAD_CAP = 1024;            // blocks with a position the matcher gave up on, per call: beyond that the whole batch goes to the host
enum : uint32_t { RXS_REACH = 1, RXS_STEPS = 2, RXS_DEPTH = 4 };
enum : uint32_t { RXO_CHAR = 0, RXO_CHAR_FOLD, RXO_CLASS, RXO_ANY, RXO_SPLIT, RXO_JMP, RXO_MATCH, RXO_LOOK, RXO_NLOOK, RXO_REP1, RXO_ATOMIC, RXO_ASSERT };
enum : uint32_t { RXA_BOL = 0, RXA_EOL, RXA_EOT, RXA_WORDB, RXA_NWORDB };
// blk[k]: the call's generation << 16 | the last hop was a skip << 15 | where every walk leaves block k -- written (plainly) only if the block is CLOSED;
// bskip[k]: the generation of the last call in which a hop jumped over block k.  Entries of other generations are stale: no fill per call.
constexpr uint32_t RXJ_EXIT = 1u << 15, RXJ_GAP = 1u << 14, RXJ_VAL = 0x3FFFu;

struct RxArgs {
    const uint32_t* image; uint32_t image_words;
    const uint8_t* text; const uint64_t* doc_off; uint32_t n_bytes, n_docs;
    const uint16_t* ucls1; const uint8_t* ucls2; uint32_t shift;
    const uint16_t* gc1; const uint8_t* gc2;
    uint16_t* nx; uint16_t* gx; uint32_t* blk; uint32_t* bskip; uint32_t* dstart;      // workspace: per byte, per byte, per block x 2, bitmap
    uint32_t gen;                                                      // this call's generation (1 .. 65535)
    uint32_t bm_words;                                                 // words of each result bitmap (n_bytes / 32 + 2)
    uint32_t* starts; uint32_t* gaps;                                  // out: the two bitmaps (every word written by k_rx_mark)
    uint32_t* status;                                                  // out: RXS_* bits, OR-ed
    uint32_t* status_host;                                             // null, or the status word's copy in pinned host memory: k_rx_mark leaves the word there (no copy back)
    uint32_t* status_next;                                             // null, or a word k_rx_mark clears for the NEXT batch (the contexts rotate through a few status words: no fill per batch)
    // per-document fallback: bad_hi[k] / bad_lo[k] = gen << 8 | (last / 255 - first) offset of a position of block k that was given up on
    // (first setter of a call appends k to bad_list, device memory: [0] count, [1 ..] blocks); k_rx_mark copies the list to bad_host
    // (pinned: [0] count, then per block two words: the block, first << 8 | last) and re-arms the count
    uint32_t* bad_hi; uint32_t* bad_lo; uint32_t* bad_list; uint32_t* bad_host;
    uint32_t bad_sets_status;                                          // spl_split_device: a listed block also sets the status word (the caller has nothing else)
    // SPL_WITH_SPECIAL: the bitmaps k_mark_docs / k_special_scan have left (null: none) -- text starts (documents AND behind every literal),
    // tokens so far (= where a literal starts); sp_words words each.  A literal is a stretch of dropped bytes with a start bit at either end
    // (encode_with_special runs the pattern over the stretches between the literals, tokenizer.rs:842-874): here a position whose hop is
    // the literal's length, with the text ending in front of it and beginning anew behind it.
    const uint32_t* sp_tstart; const uint32_t* sp_tbits; uint32_t sp_words;
};

struct RxCh { uint32_t cp, len, cls; };

// One lane's view of the text and of the program.
template <int LT> struct RxCtxT {
    static constexpr int LDS_TEXT = LT, TAB = LT - 4, BMW = (TAB + 63) / 64 * 2;     // staged bytes; tabulated positions; words of one window bitmap
    const uint32_t* img;              // LDS
    const uint8_t* s_txt;             // LDS: text[wbase, wbase + RX_LDS_TEXT)
    uint32_t wb;                      // wbase modulo 2^32 (-4 for block 0): q - wb is q's window index
    const RxArgs* a;
    uint32_t n;                       // end of this position's document
    const uint32_t* bm;               // LDS: window bitmaps, RX_BMW words each -- slot s: the characters of run set s (every byte of such a
                                      // character), slot RX_MAX_RUNSETS: the bytes that START a character
    __device__ __forceinline__ uint32_t rd(uint32_t q) const {
        const uint32_t i = q - wb;
        return i < (uint32_t)LDS_TEXT ? (uint32_t)s_txt[i] : (uint32_t)a->text[q];
    }
    // first 0 bit of bitmap m at or behind window index i0, below lim (lim if none)
    __device__ __forceinline__ uint32_t first_zero(const uint32_t* m, uint32_t i0, uint32_t lim) const {
        uint32_t w = i0 >> 5, bits = ~m[w] & (~0u << (i0 & 31));
        while (!bits && (w + 1) * 32 < lim) bits = ~m[++w];
        const uint32_t at = bits ? w * 32 + (uint32_t)__ffs((int)bits) - 1u : lim;
        return at < lim ? at : lim;
    }
    __device__ __forceinline__ uint32_t last_set_below(const uint32_t* m, uint32_t i1) const {     // highest 1 bit below index i1 (one exists)
        uint32_t w = (i1 - 1) >> 5, bits = m[w] & (0xFFFFFFFFu >> (31 - ((i1 - 1) & 31)));
        while (!bits && w > 0) bits = m[--w];
        return w * 32 + 31u - (uint32_t)__clz((int)bits);
    }
    __device__ __forceinline__ uint32_t count_set(const uint32_t* m, uint32_t i0, uint32_t i1) const {      // 1 bits in [i0, i1)
        if (i1 <= i0) return 0;
        uint32_t w = i0 >> 5, cnt = 0;
        const uint32_t w1 = (i1 - 1) >> 5;
        uint32_t bits = m[w] & (~0u << (i0 & 31));
        while (w < w1) { cnt += (uint32_t)__popc(bits); bits = m[++w]; }
        return cnt + (uint32_t)__popc(bits & (0xFFFFFFFFu >> (31 - ((i1 - 1) & 31))));
    }
    __device__ __forceinline__ uint32_t cls_of(uint32_t cp) const {
        if (cp >= 0x110000u) return C_P;
        const uint32_t blk = a->ucls1[cp >> a->shift];
        return a->ucls2[(blk << a->shift) | (cp & ((1u << a->shift) - 1u))];
    }
    __device__ __forceinline__ uint32_t cat_of(uint32_t cp) const {
        if (cp >= 0x110000u || !a->gc1) return 0;
        const uint32_t blk = a->gc1[cp >> a->shift];
        return a->gc2[(blk << a->shift) | (cp & ((1u << a->shift) - 1u))];
    }
    // (spl_regex.cpp decode: a lead byte takes the continuation bytes actually present, at most as many as it announces)
    __device__ __forceinline__ RxCh decode(uint32_t pos) const {
        const uint32_t b = rd(pos);
        if (b < 0x80) return RxCh{b, 1, cls_of(b)};
        if (b < 0xC0) return RxCh{0xFFFFFFFFu, 1, (uint32_t)C_P};
        const uint32_t want = b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
        uint32_t len = 1, cp = b & (0xFFu >> (want + 1));
        while (len < want && pos + len < n) {
            const uint32_t c = rd(pos + len);
            if ((c & 0xC0u) != 0x80u) break;
            cp = (cp << 6) | (c & 0x3Fu);
            len++;
        }
        if (len != want) return RxCh{0xFFFFFFFFu, len, (uint32_t)C_P};
        return RxCh{cp, want, cls_of(cp)};
    }
    __device__ __forceinline__ uint32_t char_len(uint32_t pos) const { return rd(pos) < 0x80 ? 1u : decode(pos).len; }
    __device__ __forceinline__ uint4 inst(uint32_t pc) const { return *reinterpret_cast<const uint4*>(img + RX_HDR_WORDS + RX_INST_WORDS * pc); }
    __device__ __forceinline__ const uint32_t* set(uint32_t s) const { return img + img[1] + RX_SET_WORDS * s; }
    __device__ __forceinline__ RxCh decode_win(uint32_t li, uint32_t B, const uint32_t* ds) const {      // the character at window index li (tabulation)
        const uint32_t b = s_txt[li];
        if (b < 0x80) return RxCh{b, 1, cls_of(b)};
        if (b < 0xC0) return RxCh{0xFFFFFFFFu, 1, (uint32_t)C_P};
        const uint32_t want = b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
        uint32_t len = 1, cp = b & (0xFFu >> (want + 1));
        while (len < want && li + len + wb < B) {
            const uint32_t i = li + len, c = s_txt[i];
            if ((c & 0xC0u) != 0x80u || ((ds[i >> 5] >> (i & 31)) & 1u)) break;           // (a document that starts here ends the character)
            cp = (cp << 6) | (c & 0x3Fu);
            len++;
        }
        if (len != want) return RxCh{0xFFFFFFFFu, len, (uint32_t)C_P};
        return RxCh{cp, want, cls_of(cp)};
    }
    __device__ __forceinline__ bool in_set(uint32_t s, const RxCh& c) const {
        const uint32_t* cs = set(s);
        bool in = ((cs[0] >> c.cls) & 1u) != 0;
        if (!in && cs[1]) in = ((cs[1] >> (c.cp == 0xFFFFFFFFu ? 0u : cat_of(c.cp))) & 1u) != 0;
        if (!in && c.cp != 0xFFFFFFFFu) {
            const uint32_t* r = img + img[5] + 2 * cs[7];
            for (uint32_t k = 0; k < cs[8]; k++) if (c.cp >= r[2 * k] && c.cp <= r[2 * k + 1]) { in = true; break; }
        }
        return in != (cs[2] != 0u);
    }
    __device__ __forceinline__ bool one_ascii(uint32_t op, uint32_t x, uint32_t b) const {
        if (op == RXO_CHAR) return b == x;
        if (op == RXO_CHAR_FOLD) return (b | 0x20u) == (x | 0x20u) && ((b | 0x20u) - 'a') < 26u;
        if (op == RXO_ANY) return b != '\n';
        return ((set(x)[3 + (b >> 5)] >> (b & 31)) & 1u) != 0;
    }
    __device__ __forceinline__ bool one(uint32_t op, uint32_t x, const RxCh& c) const {
        if (op == RXO_CHAR) return c.cp == x;
        if (op == RXO_CHAR_FOLD) {
            const uint32_t lo = x | 0x20u;
            if (c.cp == 0xFFFFFFFFu) return false;
            if (c.cp < 0x80) return (c.cp | 0x20u) == lo && ((c.cp | 0x20u) - 'a') < 26u;
            return (lo == 's' && c.cp == 0x17F) || (lo == 'k' && c.cp == 0x212A);
        }
        if (op == RXO_ANY) return c.cp != '\n';
        return in_set(x, c);
    }
    __device__ __forceinline__ bool is_word(uint32_t q) const {
        const RxCh c = decode(q);
        return c.cp == '_' || ((SPL_BIT(c.cls) & (M_L | SPL_BIT(C_N))) != 0 && c.cp != 0xFFFFFFFFu);
    }
};
using RxCtx = RxCtxT<RX_LDS_TEXT>;

// Every position of one block through the matcher (spl_regex.cpp Matcher::run with its recursion for look-aheads and atomic
// groups unrolled onto the one stack: a CALL entry below the sub-run's floor).  ONE flat loop: a lane that has no attempt pulls
// the next position of the block off a counter in LDS; a lane that has one executes ONE instruction of it, or takes the next
// way on from its stack, or finishes.  Attempts differ in length by a factor of twenty (a letter inside a word: four steps;
// a blank in front of a digit: every alternative); one position per lane for the whole kernel left a tenth of the lanes
// working (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU, profiles/r04_device_split.txt).
// Stack entry e of a lane: stk[(2 e) * RXT], stk[(2 e + 1) * RXT] (stk points at the lane's column: no bank conflicts) --
// word 0: pc | (k + 1) << 16 for "REP1 at pc, retry with k characters" (0 in the upper half: go on at pc), or pc | floor << 16
// for a CALL entry; word 1: text positions relative to the attempt's start (REP1: the run's start | its end << 16).
struct RxBlock {
    uint32_t start, B;
    const uint32_t* s_ds;             // document starts of the window
    unsigned long long ds_nz;         // bit w: word w of s_ds holds a document start (the window has at most 64 words)
    const uint32_t* s_ls;             // where special-token literals start (null: none)
    uint16_t* s_j;                    // out: level-0 pointers of the block's positions
};

__device__ __forceinline__ uint32_t rx_hop(uint32_t q, uint32_t nxv);

// One run of the matcher program from pc0, anchored at p (a whole pattern, or ONE alternative of its top-level alternation: what is
// on the stack when the run ends with nothing left to try are only its own entries).  Returns the end, RX_FAIL or RX_ABORT.
struct RxAt { uint32_t p, n, back; bool p_is_start; };
template <class C>
__device__ __forceinline__ uint32_t rx_vm(C& c, uint32_t* stk, const RxArgs& a, const RxAt& at, uint32_t pc0, uint32_t& steps) {
    constexpr uint32_t NOTYET = 0xFFFFFFFDu;
    const uint32_t* const cs = c.bm + RX_MAX_RUNSETS * C::BMW;
    const uint32_t p = at.p, n = at.n, back = at.back;
    const bool p_is_start = at.p_is_start;
    const bool trunc = n > p + (uint32_t)RX_REACH;
    const uint32_t lim = p + (uint32_t)RX_REACH;
    uint32_t pc = pc0, pos = p;
    int sp = 0, floor = 0;
    bool run = true;
    for (;;) {
        uint32_t fin = NOTYET, r = RX_FAIL;
        bool ended = false;
#define RX_PUSH(A, B) do { if (sp == RX_DEPTH) { fin = RX_ABORT; break; } stk[(2 * sp) * RXT] = (A); stk[(2 * sp + 1) * RXT] = (B); sp++; } while (0)
        if (run) do {                                       // ONE instruction (break: done with it; run == false: this way has failed)
            if (++steps > RX_STEPS) { fin = RX_ABORT; break; }                 // (the caller marks the position's block: rx_bad)
            if (trunc && pos + 8 > lim) { fin = RX_ABORT; break; }
            const uint4 in = c.inst(pc);
            const uint32_t op = in.x, x = in.y, y = in.z, f = in.w;
            if (op == RXO_MATCH) { ended = true; r = pos; run = false; break; }
            if (op == RXO_JMP) { pc = x; break; }
            if (op == RXO_SPLIT) {
                if (f != 0xFFFFFFFFu) {
                    bool can = false;
                    if (pos < n) {
                        const uint32_t b = c.rd(pos);
                        const uint32_t* fs = c.img + c.img[3] + RX_FIRST_WORDS * f;
                        can = b < 0x80 ? ((fs[b >> 5] >> (b & 31)) & 1u) != 0 : fs[4] != 0u;
                    }
                    if (!can) { pc = y; break; }
                }
                RX_PUSH(y, pos - p);
                if (fin != NOTYET) break;
                pc = x;
                break;
            }
            if (op == RXO_LOOK || op == RXO_NLOOK || op == RXO_ATOMIC) {          // the sub-program at pc + 1 as a run of its own
                RX_PUSH(pc | ((uint32_t)floor << 16), pos - p);
                if (fin != NOTYET) break;
                floor = sp;
                pc = pc + 1;
                break;
            }
            if (op == RXO_ASSERT) {
                bool ok;
                if (x == RXA_BOL) ok = pos == p && p_is_start;
                else if (x == RXA_EOT) ok = pos == n;
                else if (x == RXA_EOL) ok = pos == n || (pos + 1 == n && c.rd(pos) == '\n');
                else {
                    const bool after = pos < n && c.is_word(pos);
                    bool before = false;
                    const uint32_t room = pos - p + back;                       // bytes of the document in front of pos (as far as it matters)
                    if (room > 0) {
                        uint32_t q = pos - 1, went = 1;
                        while (went < room && went < 4 && (c.rd(q) & 0xC0u) == 0x80u) { q--; went++; }
                        if (q + c.decode(q).len == pos) before = c.is_word(q);
                    }
                    ok = (before != after) == (x == RXA_WORDB);
                }
                if (ok) pc++; else run = false;
                break;
            }
            if (op == RXO_REP1) {
                const uint4 a1 = c.inst(pc + 1);
                const uint32_t aop = a1.x, ax = a1.y;
                uint32_t q = pos, k = 0;
                bool far = false, more = true;
                if (aop == RXO_CLASS && y > 8u && pos < n) {
                    // a long run of a tabulated class: a bit scan over the window (every byte of a member character is a 1 bit)
                    const uint32_t slot = c.set(ax)[9], i0 = pos - c.wb;
                    if (slot != 0xFFFFFFFFu && i0 < (uint32_t)C::TAB) {
                        const uint32_t ni = n - c.wb;
                        const uint32_t tl = ni < (uint32_t)C::TAB ? ni : (uint32_t)C::TAB;
                        uint32_t e = c.first_zero(c.bm + slot * C::BMW, i0, tl);
                        const bool open_end = e == (uint32_t)C::TAB && e < ni;          // the table ends here, perhaps inside a character:
                        if (open_end && e > i0) e = c.last_set_below(cs, e);             // the loop below goes on from that character's start
                        const uint32_t kk = c.count_set(cs, i0, e);
                        if (kk <= y) { k = kk; q = pos + (e - i0); more = open_end; }
                    }
                }
                while (more && k < y && q < n) {
                    if (trunc && q + 8 > lim) { far = true; break; }
                    const uint32_t b = c.rd(q);
                    if (b < 0x80) {
                        if (!c.one_ascii(aop, ax, b)) break;
                        q++;
                    } else {
                        const RxCh ch = c.decode(q);
                        if (!c.one(aop, ax, ch)) break;
                        q += ch.len;
                    }
                    k++;
                }
                if (far) { fin = RX_ABORT; break; }
                steps += k;
                if (k < x) { run = false; break; }
                if (k > x && f != 1u) {                      // (k - 1 characters next time: stored as (k - 1) + 1; the run's start and end)
                    RX_PUSH(pc | (k << 16), (pos - p) | ((q - p) << 16));
                    if (fin != NOTYET) break;
                }
                pos = q;
                pc += 2;
                break;
            }
            if (pos >= n) { run = false; break; }
            {
                const uint32_t b = c.rd(pos);
                if (b < 0x80) {
                    if (!c.one_ascii(op, x, b)) { run = false; break; }
                    pos++;
                } else {
                    const RxCh ch = c.decode(pos);
                    if (!c.one(op, x, ch)) { run = false; break; }
                    pos += ch.len;
                }
            }
            pc++;
        } while (0);
#undef RX_PUSH
        if (fin == NOTYET && !run) {
            if (!ended) {
                if (sp > floor) {                                                          // the next way to go on
                    sp--;
                    const uint32_t w0 = stk[(2 * sp) * RXT], w1 = stk[(2 * sp + 1) * RXT];
                    pos = p + (w1 & 0xFFFFu);
                    pc = w0 & 0xFFFFu;
                    if (w0 >> 16) {                                                        // a REP1 gives a character back: k of them from the run's start
                        const uint32_t k = (w0 >> 16) - 1u;
                        const uint32_t e_old = p + (w1 >> 16), ei = e_old - c.wb;
                        uint32_t e_new;
                        // (what is charged against RX_STEPS is the work done: one bit scan inside the tabulated window, k characters re-counted
                        //  beyond it.  Charging k either way made `\s*[\r\n]+` give up on 200 blanks without a newline -- 200 + 199 + ... steps
                        //  for 200 bit scans; found by the PCRE2 pin of tests/test_gpu_custom_pattern.py, round 5)
                        if (ei <= (uint32_t)C::TAB) { e_new = k ? c.wb + c.last_set_below(cs, ei) : pos; steps += 1u; }     // the start of the run's last character
                        else { e_new = pos; for (uint32_t j = 0; j < k; j++) e_new += c.char_len(e_new); steps += k; }
                        if (k > c.inst(pc).y) { stk[(2 * sp) * RXT] = pc | (k << 16); stk[(2 * sp + 1) * RXT] = (pos - p) | ((e_new - p) << 16); sp++; }
                        pos = e_new;
                        pc += 2;
                    }
                    run = true;
                } else {
                    ended = true; r = RX_FAIL;
                }
            }
            if (ended) {
                // a run has ended with r: the top-level one, or the sub-run of the CALL entry below the floor
                sp = floor;
                if (floor == 0) fin = r;
                else {
                    sp--;
                    const uint32_t w0 = stk[(2 * sp) * RXT];
                    const uint32_t psave = p + stk[(2 * sp + 1) * RXT];
                    floor = (int)(w0 >> 16);
                    const uint4 in = c.inst(w0 & 0xFFFFu);
                    if (in.x == RXO_ATOMIC) {
                        if (r != RX_FAIL) { pos = r; pc = in.y; run = true; }
                    } else if ((r != RX_FAIL) == (in.x == RXO_LOOK)) {
                        pos = psave; pc = in.y; run = true;
                    }
                }
            }
        }
        if (fin != NOTYET) return fin;
    }
}

// One SIMPLE alternative (regex_device_image: a straight line of one-character / run items in which giving characters back can never
// help -- possessive, or the item's characters cannot be taken by what follows) at p: greedy, item by item, no stack.  Every lane of
// the wavefront walks the same items: this is the code that runs at full width.
template <bool UNI, class C>
__device__ __forceinline__ uint32_t rx_simple_alt(const C& c, const uint32_t* items, uint32_t n_items, uint32_t tail, uint32_t p, uint32_t n, bool on,
                                                  uint32_t& why) {
    const uint32_t* const cs = c.bm + RX_MAX_RUNSETS * C::BMW;
    const bool trunc = n > p + (uint32_t)RX_REACH;
    const uint32_t lim = p + (uint32_t)RX_REACH;
    uint32_t pos = p, rs = p, rk = 0;                       // rs, rk: where the last item began and how many characters it took
    bool ok = on;
    for (uint32_t it = 0; it < n_items; it++) {
        const uint4 item = *reinterpret_cast<const uint4*>(items + 4 * it);       // op, x, min, max
        // (the same for every lane: in scalar registers the tests on them are scalar branches, not exec-mask detours)
        // (UNI: the items are the same for every lane of the wavefront -- k_rx_match's lock step; the walk kernel's lanes each have their own)
        const uint32_t op = UNI ? __builtin_amdgcn_readfirstlane(item.x) : item.x, x = UNI ? __builtin_amdgcn_readfirstlane(item.y) : item.y;
        const uint32_t mn = UNI ? __builtin_amdgcn_readfirstlane(item.z) : item.z, mx = UNI ? __builtin_amdgcn_readfirstlane(item.w) : item.w;
        if (op == RXO_CHAR && mx == 1u && x < 0x80u) {          // an ASCII literal, once or not at all: no branch per lane
            const bool hit = ok && pos < n && c.rd(pos) == x;
            ok = ok && (hit || mn == 0u);
            rs = pos; rk = hit ? 1u : 0u;
            pos += hit ? 1u : 0u;
            continue;
        }
        if (mx == 1u) {                                        // one character, once or not at all: no loop
            bool hit = false;
            uint32_t len = 1;
            if (ok && pos < n) {
                if (trunc && pos + 8 > lim) { why = RXS_REACH; return RX_ABORT; }
                const uint32_t b = c.rd(pos);
                if (b < 0x80) hit = c.one_ascii(op, x, b);
                else { const RxCh ch = c.decode(pos); hit = c.one(op, x, ch); len = ch.len; }
            }
            ok = ok && (hit || mn == 0u);
            rs = pos; rk = hit ? 1u : 0u;
            pos += hit ? len : 0u;
            continue;
        }
        if (!ok) continue;
        uint32_t q = pos, k = 0;
        bool more = true;
        if (op == RXO_CLASS && mx > 8u && pos < n) {
            const uint32_t slot = c.set(x)[9], i0 = pos - c.wb;
            if (slot != 0xFFFFFFFFu && i0 < (uint32_t)C::TAB) {
                const uint32_t ni = n - c.wb;
                const uint32_t tl = ni < (uint32_t)C::TAB ? ni : (uint32_t)C::TAB;
                // the 64 bits from i0 on, without a loop: nearly every run ends inside them (the words behind a bitmap's end that this may
                // read belong to the next bitmap / the padding, and lie beyond tl)
                const uint32_t* const m = c.bm + slot * C::BMW + (i0 >> 5);
                const uint32_t* const s3 = cs + (i0 >> 5);
                const uint32_t sh = i0 & 31u;
                const uint32_t mlo = __builtin_amdgcn_alignbit(m[1], m[0], sh), mhi = __builtin_amdgcn_alignbit(m[2], m[1], sh);
                const uint32_t er = ~mlo ? (uint32_t)__ffs((int)~mlo) - 1u : ~mhi ? 32u + (uint32_t)__ffs((int)~mhi) - 1u : 64u;
                if (er < 64u && i0 + er < tl) {
                    const uint32_t clo = __builtin_amdgcn_alignbit(s3[1], s3[0], sh), chi = __builtin_amdgcn_alignbit(s3[2], s3[1], sh);
                    const uint32_t kk = er >= 32u ? (uint32_t)__popc(clo) + (uint32_t)__popc(chi & ((1u << (er - 32u)) - 1u))
                                                  : (uint32_t)__popc(clo & ((1u << er) - 1u));
                    if (kk <= mx) { k = kk; q = pos + er; more = false; }
                } else {
                    uint32_t e = c.first_zero(c.bm + slot * C::BMW, i0, tl);
                    const bool open_end = e == (uint32_t)C::TAB && e < ni;
                    if (open_end && e > i0) e = c.last_set_below(cs, e);
                    const uint32_t kk = c.count_set(cs, i0, e);
                    if (kk <= mx) { k = kk; q = pos + (e - i0); more = open_end; }
                }
            }
        }
        while (more && k < mx && q < n) {
            if (trunc && q + 8 > lim) { why = RXS_REACH; return RX_ABORT; }
            const uint32_t b = c.rd(q);
            if (b < 0x80) {
                if (!c.one_ascii(op, x, b)) break;
                q++;
            } else {
                const RxCh ch = c.decode(q);
                if (!c.one(op, x, ch)) break;
                q += ch.len;
            }
            k++;
        }
        if (k < mn) ok = false;
        rs = pos; rk = k;
        pos = q;
    }
    if (tail == 0u || !ok) return ok ? pos : RX_FAIL;
    // the tail, at every length of the last run from the longest down to `cut` (the matcher's way back into that run, as a loop)
    const uint4 tl = *reinterpret_cast<const uint4*>(items + 4 * n_items);          // op, x, -, the fewest characters the run may keep (~0: all of them)
    const uint32_t cut = tl.w == 0xFFFFFFFFu ? rk : tl.w;
    for (uint32_t t = rk;; t--) {
        if (trunc && pos + 8 > lim) { why = RXS_REACH; return RX_ABORT; }
        if (tail == 1u || tail == 2u || tail == 3u) {                                 // one character of the tail's item at pos?
            bool takes = false;
            uint32_t len = 1;
            if (pos < n) {
                const uint32_t b = c.rd(pos);
                if (b < 0x80) takes = c.one_ascii(tl.x, tl.y, b);
                else { const RxCh ch = c.decode(pos); takes = c.one(tl.x, tl.y, ch); len = ch.len; }
            }
            if (tail == 1u) { if (takes) return pos + len; }
            else if (takes == (tail == 2u)) return pos;
        } else if (tail == 4u) { if (pos == n || (pos + 1 == n && c.rd(pos) == '\n')) return pos; }
        else if (pos == n) return pos;
        if (t <= cut) return RX_FAIL;
        // one character less: the start of the run's last character
        const uint32_t ei = pos - c.wb;
        if (ei <= (uint32_t)C::TAB) pos = c.wb + c.last_set_below(cs, ei);
        else { pos = rs; for (uint32_t j = 0; j + 1 < t; j++) pos += c.char_len(pos); }
    }
}

// The attempt at every position of the block, one position per lane, the ALTERNATIVES of the pattern in step: all lanes try
// alternative 0, then those it did not match try alternative 1, ... (leftmost-first: the first alternative that matches is the match).
// A simple alternative is evaluated at full width; another one runs the matcher program from its first instruction for the lanes
// that can start it (its first-character filter) -- on GPT-2's pattern that is `\s+(?!\S)` at blanks that no earlier alternative took.
// the matcher gave up at position p: its 256-byte block goes on the list once per call, with the first and the last such position of the
// block (bad_lo / bad_hi: the call's generation above an 8-bit offset, kept up by atomicMax -- entries of earlier calls are smaller, so
// nothing is ever cleared); the host splits the documents that overlap [first, last] of a listed block itself.  A list that overflows
// turns into the whole-batch status.
__device__ __forceinline__ void rx_bad(const RxArgs& a, uint32_t p) {
    const uint32_t blk = p >> 8, o = p & 255u;
    atomicMax(&a.bad_lo[blk], (a.gen << 8) | (255u - o));
    if ((atomicMax(&a.bad_hi[blk], (a.gen << 8) | o) >> 8) == a.gen) return;          // (already listed in this call)
    const uint32_t i = atomicAdd(&a.bad_list[0], 1u);
    if (i < RX_BAD_CAP) a.bad_list[1 + i] = blk;
    else atomicOr(a.status, RXS_STEPS);
}
__device__ __forceinline__ void rx_attempts(RxCtx& c, uint32_t* stk, const RxBlock& bk, const RxArgs& a, uint32_t pl) {
    constexpr int DSW = (RX_BACK + RXB + RX_REACH) / 32 + 2;
    const uint32_t* const cs = c.bm + RX_MAX_RUNSETS * RX_BMW;
    auto ds_bit = [&](uint32_t i) { return (bk.s_ds[i >> 5] >> (i & 31)) & 1u; };
    // the first document start behind window index wi0 (global position; `none` if the window holds none): the rest of wi0's word, then
    // the next word that has one -- by the 64-bit map of such words, no loop
    auto next_start = [&](uint32_t wi0, uint32_t none) -> uint32_t {
        const uint32_t i = wi0 + 1, w = i >> 5;
        uint32_t bits = bk.s_ds[w] & (~0u << (i & 31)), ww = w;
        if (!bits) {
            const unsigned long long rest = w + 1 < 64u ? bk.ds_nz >> (w + 1) : 0ull;
            if (!rest) return none;
            ww = w + 1 + (uint32_t)__builtin_ctzll(rest);
            bits = bk.s_ds[ww];
        }
        const uint32_t e = c.wb + ww * 32 + (uint32_t)__ffs((int)bits) - 1u;
        return e < none ? e : none;
    };
    const uint32_t p = bk.start + pl;
    if (p >= bk.B) bk.s_j[pl] = (uint16_t)(RXJ_EXIT | 0u);
    const uint32_t wi = pl + RX_BACK;
    // (the WHOLE batch has been given up on already -- more than RX_BAD_CAP blocks hold a position the matcher gave up on --: no more
    //  attempts.  This bounds what a pattern that backtracks without end can cost: RX_BAD_CAP blocks' worth of attempts run into RX_STEPS,
    //  everything behind them is skipped)
    const bool given_up = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    // (a byte inside a character: the walk never stands there, its hop is "one on")
    const bool lit = p < bk.B && bk.s_ls && ((bk.s_ls[wi >> 5] >> (wi & 31)) & 1u) != 0u;       // a special-token literal starts here
    bool act = p < bk.B && !given_up && !lit && ((cs[wi >> 5] >> (wi & 31)) & 1u) != 0u;
    RxAt at{p, bk.B, 0u, false};
    uint32_t b0 = 0;
    if (act) {
        at.p_is_start = ds_bit(wi) != 0u;
        at.n = next_start(wi, bk.B);                       // the end of p's document: the next document start behind p (or the end of the corpus)
        if (!at.p_is_start) {                              // bytes of the same document in front of p, up to RX_BACK: the four start bits in front of wi
            const uint32_t i4 = wi - RX_BACK;              // (wi >= RX_BACK)
            const uint32_t prev4 = ((bk.s_ds[i4 >> 5] >> (i4 & 31)) | ((i4 & 31) > 28u ? bk.s_ds[(i4 >> 5) + 1] << (32u - (i4 & 31)) : 0u)) & 0xFu;
            at.back = prev4 ? (uint32_t)__clz((int)(prev4 << 28)) + 1u : (uint32_t)RX_BACK;
            at.back = at.back < p ? at.back : p;
        }
        c.n = at.n;
        b0 = c.rd(p);                                      // (p < n: the attempt begins with a byte)
    }
    const uint32_t* const alts = c.img + __builtin_amdgcn_readfirstlane(c.img[11]);
    const uint32_t* const fsets = c.img + __builtin_amdgcn_readfirstlane(c.img[3]);
    const uint32_t n_alts = __builtin_amdgcn_readfirstlane(alts[0]);
    uint32_t viable = act ? c.img[c.img[8] + (b0 < 0x80 ? b0 : 128u)] : 0u;
#ifndef SPL_RX_PAIRS
#define SPL_RX_PAIRS 1            /* 0: dispatch on the first byte alone (A/B) */
#endif
    {   // two ASCII bytes: the alternatives that can start with that PAIR (regex_device_image: byte classes, a table over pairs of them)
        const uint32_t po = __builtin_amdgcn_readfirstlane(c.img[12]);
        if (SPL_RX_PAIRS && po != 0u && act && b0 < 0x80 && p + 1 < at.n) {
            const uint32_t b1 = c.rd(p + 1);
            if (b1 < 0x80) {
                const uint32_t* const pt = c.img + po;
                const uint32_t c0 = (pt[b0 >> 2] >> ((b0 & 3u) * 8u)) & 0xFFu, c1 = (pt[32 + (b1 >> 2)] >> ((b1 & 3u) * 8u)) & 0xFFu;
                viable = pt[65 + c0 * pt[64] + c1];
            }
        }
    }
    uint32_t e = RX_FAIL, steps = 0;
#ifndef SPL_RX_GROUPS
#define SPL_RX_GROUPS 1           /* 0: every alternative in a sweep of its own (A/B) */
#endif
    for (uint32_t ai = 0; ai < n_alts;) {                  // (uniform)
        // (which alternatives the attempt's first byte can start: one word from the image's first-byte table instead of a filter per alternative)
        uint4 alt = *reinterpret_cast<const uint4*>(alts + 4 + 4 * ai);           // filter, flags, first instruction, items (count | offset << 16)
        alt.x = __builtin_amdgcn_readfirstlane(alt.x); alt.y = __builtin_amdgcn_readfirstlane(alt.y);     // (the same for every lane)
        alt.z = __builtin_amdgcn_readfirstlane(alt.z); alt.w = __builtin_amdgcn_readfirstlane(alt.w);
        const uint32_t glen = SPL_RX_GROUPS && (alt.y & 1u) ? (alt.y >> 16) & 0xFFu : 1u;
        if (glen > 1u) {
            // alternatives ai .. ai + glen - 1 have ONE shape (regex_device_image): a sweep for all of them, every lane with the items of the
            // first of them that its bytes can start -- ` ?\p{L}+`, ` ?\p{N}+`, ` ?[^\s\p{L}\p{N}]+` are one sweep, not three; a lane whose
            // alternative fails takes its next one of the group in the next sweep (leftmost-first is kept: the group is consecutive)
            uint32_t vg = act && e == RX_FAIL ? (viable >> ai) & ((1u << glen) - 1u) : 0u;
            while (__any(vg != 0u)) {
                const bool on = vg != 0u;
                const uint32_t k = ai + (on ? (uint32_t)__ffs((int)vg) - 1u : 0u);
                const uint32_t aw = alts[4 + 4 * k + 3];
                uint32_t why = 0;
                const uint32_t r = rx_simple_alt<false>(c, c.img + (aw >> 16), alt.w & 0xFFFFu, (alt.y >> 8) & 0xFFu, p, at.n, on, why);
                if (on) {
                    if (r == RX_ABORT) { rx_bad(a, p); act = false; vg = 0; }
                    else if (r != RX_FAIL && r > p) { e = r; vg = 0; }
                    else if (r != RX_FAIL) { act = false; vg = 0; }       // (an empty match: find_iter skips the character, as behind no match)
                    else vg &= vg - 1u;
                }
            }
            ai += glen;
            continue;
        }
        bool can = act && e == RX_FAIL && ((viable >> (ai & 31u)) & 1u) != 0u;
        if (n_alts > 32u && can && alt.x != 0xFFFFFFFFu) {
            const uint32_t* fs = fsets + RX_FIRST_WORDS * alt.x;
            can = b0 < 0x80 ? ((fs[b0 >> 5] >> (b0 & 31)) & 1u) != 0 : fs[4] != 0u;
        }
        ai++;
        if (!__any(can)) continue;
        uint32_t r = RX_FAIL, why = 0;
        if (alt.y & 1u) r = rx_simple_alt<true>(c, c.img + (alt.w >> 16), alt.w & 0xFFFFu, (alt.y >> 8) & 0xFFu, p, at.n, can, why);
        else if (can) r = rx_vm(c, stk, a, at, alt.z, steps);
        if (can && r == RX_ABORT) { rx_bad(a, p); act = false; }
        else if (can && r != RX_FAIL && r > p) e = r;
        else if (can && r != RX_FAIL) act = false;         // (an empty match: find_iter skips the character, as behind no match)
    }
    if (p < bk.B) {                                        // the position's hop
        uint32_t nxv = 1u | 0x8000u;
        if (lit) {                                         // the literal: to the next text start (the one behind it), dropped
            nxv = (next_start(wi, bk.B) - p) | 0x8000u;
        } else if (e != RX_FAIL) {
            uint32_t d = e - p;
            if (d > (uint32_t)RX_REACH - 8u) { rx_bad(a, p); d = RX_REACH - 8; }
            nxv = d;
            // a hop over whole blocks: they may not be touched by the walk at all
            for (uint32_t k = (p >> 8) + 1; (k + 1) * (uint32_t)RXB <= p + d; k++) a.bskip[k] = a.gen;
        } else if (!given_up && ((cs[wi >> 5] >> (wi & 31)) & 1u)) {
            nxv = c.char_len(p) | 0x8000u;                 // no match here (or an empty one): the character is skipped
        }
        a.nx[p] = (uint16_t)nxv;
        bk.s_j[pl] = (uint16_t)rx_hop(pl, nxv);
    }
}

// Pointer doubling over one block: j[q] (a local index, or RXJ_EXIT | last-hop-was-a-skip | offset behind the block) becomes
// the place where the walk from q leaves the block.  With `lev`, level k's pointers (before round k) are kept at lev[k * RXB + q].
template <int NTH> __device__ __forceinline__ void rx_double(uint16_t* j, uint16_t* lev, int tid) {
    for (int k = 0; k < 8; k++) {
        uint32_t v2[RXB / NTH];
        for (int e = 0; e < RXB / NTH; e++) {
            const int q = tid + e * NTH;
            const uint32_t v = j[q];
            if (lev) lev[k * RXB + q] = (uint16_t)v;
            v2[e] = (v & RXJ_EXIT) ? v : (uint32_t)j[v];
        }
        __syncthreads();
        for (int e = 0; e < RXB / NTH; e++) j[tid + e * NTH] = (uint16_t)v2[e];
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t rx_hop(uint32_t q, uint32_t nxv) {       // level-0 pointer of local position q
    const uint32_t to = q + (nxv & 0x7FFFu);
    return to < (uint32_t)RXB ? to : (RXJ_EXIT | ((nxv & 0x8000u) ? RXJ_GAP : 0u) | (to - (uint32_t)RXB));
}

__global__ __launch_bounds__(RXT) void k_rx_match(RxArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_img[];         // the program image: image_words (dynamic: what the pattern needs)
    __shared__ uint32_t s_stk[2 * RX_DEPTH * RXT];
    __shared__ __attribute__((aligned(4))) uint8_t s_txt[RX_LDS_TEXT + 4];
    __shared__ uint32_t s_ds[(RX_BACK + RXB + RX_REACH) / 32 + 2];            // document starts of [wbase, wbase + RX_BACK + RXB + RX_REACH)
    __shared__ uint16_t s_j[RXB];
    __shared__ uint32_t s_nz[2];
    __shared__ uint32_t s_ls[(RX_BACK + RXB + RX_REACH) / 32 + 2];            // literal starts of the window (SPL_WITH_SPECIAL)
    __shared__ uint32_t s_bm[(RX_MAX_RUNSETS + 1) * RX_BMW + 2];     // (+ 2: the 64-bit window of a run scan may read two words on)
    const int tid = (int)threadIdx.x;
    const uint32_t B = a.n_bytes;
    const uint32_t start = blockIdx.x * (uint32_t)RXB;
    const int64_t wbase = (int64_t)start - RX_BACK;
    constexpr int DSW = (RX_BACK + RXB + RX_REACH) / 32 + 2;
    for (int i = tid; i < (int)a.image_words; i += RXT) s_img[i] = a.image[i];
    for (int i = tid; i < DSW; i += RXT) s_ds[i] = 0;
    for (int w = tid; w < (RX_LDS_TEXT + 3) / 4; w += RXT) {
        const int64_t g = wbase + 4 * (int64_t)w;
        uint32_t v = 0;
        if (g >= 0 && g + 4 <= (int64_t)B) v = *reinterpret_cast<const uint32_t*>(a.text + g);      // (the text buffer is 16-byte aligned, start a multiple of 256)
        else for (int k = 0; k < 4; k++) if (g + k >= 0 && g + k < (int64_t)B) v |= (uint32_t)a.text[g + k] << (8 * k);
        reinterpret_cast<uint32_t*>(s_txt)[w] = v;
    }
    // documents that start inside the window: RXT-ary search for the first one at or behind its begin, then their bits
    const uint64_t wlo = wbase > 0 ? (uint64_t)wbase : 0ull, whi = (uint64_t)(wbase + RX_BACK + RXB + RX_REACH);
    uint32_t lo = 0, hi = a.n_docs;
    uint64_t p_held = ~0ull;                            // doc_off[d_held] from the first round, if it settled the search
    uint32_t d_held = 0xFFFFFFFFu, d_end = 0;
    if (wlo != 0 && hi > (uint32_t)RXT) {
        // first round by interpolation (as the tile kernel's): with documents of similar size the answer lies within RXT entries of
        // wlo * n_docs / n_bytes, ONE round of loads finds it and already holds the window's documents
        const float gf = (float)wlo * ((float)hi * __builtin_amdgcn_rcpf((float)B));
        const uint32_t g = gf >= (float)hi ? hi : (uint32_t)gf;
        const uint32_t glo = g > (uint32_t)(RXT / 2) ? g - RXT / 2 : 0u;
        const uint32_t ghi = glo + RXT < hi ? glo + RXT : hi;
        const uint32_t idx = glo + (uint32_t)tid;
        const uint64_t p1 = idx < ghi ? a.doc_off[idx] : ~0ull;
        const uint32_t cnt = (uint32_t)__syncthreads_count(idx < ghi && p1 < wlo);
        if (cnt == 0) hi = glo;
        else if (cnt == ghi - glo) lo = ghi;
        else { lo = hi = glo + cnt; p_held = p1; d_held = idx; d_end = ghi; }
    }
    while (wlo != 0 && lo < hi) {
        const uint32_t span = hi - lo, stp = (span + RXT - 1) / RXT;
        const uint64_t idx = (uint64_t)lo + (uint64_t)tid * stp;
        const bool below = idx < hi && a.doc_off[idx] < wlo;
        const uint32_t cnt = (uint32_t)__syncthreads_count(below);
        if (cnt == 0) { hi = lo; break; }
        const uint64_t nhi = (uint64_t)lo + (uint64_t)cnt * stp;
        lo = lo + (cnt - 1) * stp + 1;
        hi = nhi < hi ? (uint32_t)nhi : hi;
    }
    __syncthreads();
    uint32_t base = lo;
    if (d_end > lo) {                                  // the window's documents from the first round's loads
        const bool in = d_held >= lo && d_held < d_end && p_held < whi && p_held < (uint64_t)B;
        if (in) { const uint32_t i = (uint32_t)((int64_t)p_held - wbase); atomicOr(&s_ds[i >> 5], 1u << (i & 31)); }
        base = __syncthreads_or(d_held == d_end - 1u && in) ? d_end : 0xFFFFFFFFu;       // more only if the last entry fetched is still inside
    }
    for (; base != 0xFFFFFFFFu; base += RXT) {
        const uint64_t d = (uint64_t)base + tid;
        uint64_t dp = ~0ull;
        if (d < a.n_docs) dp = a.doc_off[d];
        const bool in = dp < whi && dp < (uint64_t)B;
        if (in) { const uint32_t i = (uint32_t)((int64_t)dp - wbase); atomicOr(&s_ds[i >> 5], 1u << (i & 31)); }
        if (!__syncthreads_or(tid == RXT - 1 && in)) break;
    }
    if (a.sp_tstart) {
        // the window's words of the two bitmaps (the window begins 4 bits in front of a multiple of 256): text starts join the
        // document starts, literal starts END the text in front of them and are kept apart as well
        for (int w = tid; w < DSW; w += RXT) {
            const int64_t g = wbase + 32 * (int64_t)w;
            auto win = [&](const uint32_t* bm) -> uint32_t {
                if (g + 32 <= 0) return 0u;
                if (g < 0) return (0 < (int64_t)a.sp_words ? bm[0] : 0u) << (uint32_t)(-g);
                const uint64_t wi0 = (uint64_t)g >> 5;
                const uint32_t sh = (uint32_t)g & 31u;
                uint32_t v = wi0 < a.sp_words ? bm[wi0] >> sh : 0u;
                if (sh && wi0 + 1 < a.sp_words) v |= bm[wi0 + 1] << (32u - sh);
                return v;
            };
            const uint32_t ls = win(a.sp_tbits);
            s_ls[w] = ls;
            s_ds[w] |= win(a.sp_tstart) | ls;
        }
        __syncthreads();
    }
    static_assert(DSW <= 64, "the map of the window's words with a document start is one 64-bit word");
    if (tid < 64) {
        const unsigned long long nz = __ballot(tid < DSW && s_ds[tid] != 0u);
        if (tid == 0) { s_nz[0] = (uint32_t)nz; s_nz[1] = (uint32_t)(nz >> 32); }
    }
    __syncthreads();
    const unsigned long long ds_nz = (unsigned long long)s_nz[0] | ((unsigned long long)s_nz[1] << 32);
    // ---- the window's characters, tabulated once for all 256 attempts: which bytes start a character (a continuation byte that
    // a lead byte in front of it takes -- same document, as many as it announces -- does not; find_iter only ever stands on the
    // others), and for every class set that a run instruction repeats, the bytes of its member characters
    RxCtx c{s_img, s_txt, (uint32_t)wbase, &a, B, s_bm};
    {
        const uint32_t nrs = __builtin_amdgcn_readfirstlane(s_img[9]);
        const uint32_t* rs = s_img + s_img[10];
        for (int r = 0; r * RXT < RX_BMW * 32; r++) {
            const uint32_t i = (uint32_t)(r * RXT + tid);
            const int64_t q = wbase + (int64_t)i;
            const bool valid = i < (uint32_t)RX_TAB && q >= 0 && q < (int64_t)B;
            uint32_t li = i;
            if (valid && (s_txt[i] & 0xC0u) == 0x80u) {
                for (uint32_t j = 1; j <= 3 && j <= i; j++) {
                    if ((s_ds[(i - j + 1) >> 5] >> ((i - j + 1) & 31)) & 1u) break;       // a document starts between that byte and this one
                    const uint32_t cb = s_txt[i - j];
                    if ((cb & 0xC0u) == 0x80u) continue;
                    if (cb >= 0xC0u && (cb < 0xE0u ? 2u : cb < 0xF0u ? 3u : 4u) > j) li = i - j;
                    break;
                }
            }
            // (an ASCII byte is a character by itself and its membership is a bit of the set's ASCII words: no decode, no class lookup --
            //  only bytes beyond ASCII go through the two-stage table in global memory)
            const uint32_t bv = valid ? (uint32_t)s_txt[li] : 0u;
            const bool asc = bv < 0x80u;
            RxCh ch{bv, 1, 0};
            if (valid && !asc) ch = c.decode_win(li, B, s_ds);
            const unsigned long long mcs = __ballot(valid && li == i);
            const uint32_t w0 = (uint32_t)(r * RXT + (tid & ~63)) >> 5;
            const bool wr = (tid & 63) == 0 && w0 + 1 < (uint32_t)RX_BMW;
            if (wr) { s_bm[RX_MAX_RUNSETS * RX_BMW + w0] = (uint32_t)mcs; s_bm[RX_MAX_RUNSETS * RX_BMW + w0 + 1] = (uint32_t)(mcs >> 32); }
            for (uint32_t k = 0; k < nrs; k++) {
                const uint32_t sx = __builtin_amdgcn_readfirstlane(rs[k]);
                const bool in = asc ? ((c.set(sx)[3 + (bv >> 5)] >> (bv & 31)) & 1u) != 0u : c.in_set(sx, ch);
                const unsigned long long m = __ballot(valid && in);
                if (wr) { s_bm[k * RX_BMW + w0] = (uint32_t)m; s_bm[k * RX_BMW + w0 + 1] = (uint32_t)(m >> 32); }
            }
        }
    }
    __syncthreads();
#ifdef RX_NOVM
    for (int e = 0; e < RXB / RXT; e++) {
        const uint32_t q = (uint32_t)(tid + e * RXT), p = start + q;
        if (p < B) a.nx[p] = (uint16_t)(1u | 0x8000u);
        s_j[q] = (uint16_t)(p < B ? rx_hop(q, 1u | 0x8000u) : (RXJ_EXIT | 0u));
    }
#else
    {
        RxBlock bk{start, B, s_ds, ds_nz, a.sp_tstart ? s_ls : nullptr, s_j};
        rx_attempts(c, s_stk + tid, bk, a, (uint32_t)tid);
    }
#endif
    // the block's own words of the document-start bitmap
    for (int e = 0; e < RXB / RXT; e++) {
        const uint32_t q = (uint32_t)(tid + e * RXT), wi = q + RX_BACK;
        const bool is_start = start + q < B && ((s_ds[wi >> 5] >> (wi & 31)) & 1u) != 0u;
        const unsigned long long m = __ballot(is_start);
        if ((tid & 63) == 0) { a.dstart[(start >> 5) + (q >> 5)] = (uint32_t)m; a.dstart[(start >> 5) + (q >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    rx_double<RXT>(s_j, nullptr, tid);
    const uint32_t g0 = s_j[0];
    bool same = true;
    for (int e = 0; e < RXB / RXT; e++) {
        const uint32_t q = (uint32_t)(tid + e * RXT), p = start + q;
        const uint32_t g = s_j[q];
        if (p < B) a.gx[p] = (uint16_t)(((g & RXJ_GAP) ? 0x8000u : 0u) | (g & RXJ_VAL));
        same = same && (p >= B || g == g0);
    }
    if (__syncthreads_and(same) && tid == 0)
        a.blk[blockIdx.x] = (a.gen << 16) | ((g0 & RXJ_GAP) ? 0x8000u : 0u) | (g0 & RXJ_VAL);
}

__global__ __launch_bounds__(RXB) void k_rx_mark(RxArgs a) {
    __shared__ uint16_t s_j[RXB];
    __shared__ uint16_t s_lev[8 * RXB];
    __shared__ uint8_t s_m[RXB];
    __shared__ unsigned long long s_cm[RXB / 64], s_gm[RXB / 64];
    __shared__ uint32_t s_gb[RXB / 32 + 9];              // dropped bytes of the block's nodes: a character's 4, a special-token literal's 255
    __shared__ uint32_t s_e[2];
    const int tid = (int)threadIdx.x;
    const uint32_t B = a.n_bytes;
    const uint32_t b = blockIdx.x, start = b * (uint32_t)RXB;
    const uint32_t p = start + (uint32_t)tid;
    if (b == 0 && a.bad_host) {
        // the blocks the matcher gave up on, for the host (k_rx_match is done: the list is final); the count re-armed for the next call
        const uint32_t nb = a.bad_list[0] < RX_BAD_CAP ? a.bad_list[0] : RX_BAD_CAP;
        for (uint32_t i = (uint32_t)tid; i < nb; i += RXB) {
            const uint32_t k = a.bad_list[1 + i];
            a.bad_host[1 + 2 * i] = k;
            a.bad_host[2 + 2 * i] = ((255u - (a.bad_lo[k] & 255u)) << 8) | (a.bad_hi[k] & 255u);
        }
        __syncthreads();
        if (tid == 0) {
            if (a.bad_sets_status && a.bad_list[0]) atomicOr(a.status, RXS_REACH);
            a.bad_host[0] = a.bad_list[0]; a.bad_list[0] = 0u;
        }
    } else if (b == 0 && tid == 0) a.bad_list[0] = 0u;
    if (b == 0 && tid == 1) {
        if (a.status_next) *a.status_next = 0u;
        // (k_rx_match is done: what it gave up on is final; what THIS kernel gives up on goes to the host word directly, below)
        if (a.status_host) { const uint32_t st = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (st) *a.status_host = st; }
    }
    if (tid == 0) {
        // where the walk from position 0 enters this block
        int64_t k = (int64_t)b - 1;
        uint32_t v = 0;
        // (nothing to do for a batch that goes to the host anyway; and a stretch of open blocks longer than RX_MAX_OPEN -- megabytes of
        //  digits under \p{N}{1,2} -- is given up on: every block of it would walk back through all of it)
        bool lost = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        while (k >= 0 && !lost) {
            v = a.blk[k];
            if ((v >> 16) == a.gen && a.bskip[k] != a.gen) break;            // closed, and no hop jumps over it
            k--;
            if ((int64_t)b - k > (int64_t)RX_MAX_OPEN) { atomicOr(a.status, RXS_REACH); if (a.status_host) *a.status_host = RXS_REACH; lost = true; }
        }
        if (lost) k = (int64_t)b - 1, v = 0;                    // (any entry will do: the bitmaps are not used)
        uint32_t E = 0, pg = 0;
        if (k >= 0) { E = ((uint32_t)k + 1u) * RXB + (v & 0x7FFFu); pg = (v >> 15) & 1u; }
        for (uint32_t kk = (uint32_t)(k + 1); kk < b && E < B; kk++) {
            if (E >= (kk + 1) * (uint32_t)RXB) continue;           // the walk jumps over block kk
            const uint32_t g = a.gx[E];
            E = (kk + 1) * (uint32_t)RXB + (g & 0x7FFFu);
            pg = g >> 15;
        }
        s_e[0] = E; s_e[1] = pg;
    }
    const uint32_t nxv = p < B ? (uint32_t)a.nx[p] : (1u | 0x8000u);
    s_j[tid] = (uint16_t)(p < B ? rx_hop((uint32_t)tid, nxv) : (RXJ_EXIT | 0u));
    s_m[tid] = 0;
    if (tid < RXB / 32 + 9) s_gb[tid] = 0;
    __syncthreads();
    rx_double<RXB>(s_j, s_lev, tid);
    const uint32_t E = s_e[0], pg_in = s_e[1];
    // The block writes its own eight words of either bitmap PLAINLY (no fill in front of the kernels, no atomics): the dropped bytes of a
    // character -- or a special-token literal -- that reaches over from the block in front are this block's to set (the walk enters at E;
    // if the hop that got there was a skip, [start, E) is the rest of it: a skip is shorter than a block), what its own nodes drop
    // beyond its end is the next block's.  The last block also clears the bitmaps' closing words.
    const uint32_t wb0 = start >> 5;
    auto put = [&](uint32_t* bm, uint32_t w, uint32_t v) { if (w < a.bm_words) bm[w] = v; };
    if (b == gridDim.x - 1)
        for (uint32_t w = wb0 + RXB / 32 + (uint32_t)tid; w < a.bm_words; w += RXB) { a.starts[w] = 0u; a.gaps[w] = 0u; }
    if (E >= start + (uint32_t)RXB || E >= B) {                    // the walk does not touch this block (uniform): nothing starts, nothing is dropped
        if (tid < RXB / 32) { put(a.starts, wb0 + (uint32_t)tid, 0u); put(a.gaps, wb0 + (uint32_t)tid, 0u); }
        return;
    }
    if ((uint32_t)tid == E - start) s_m[tid] = 1;
    if (pg_in && (uint32_t)tid < E - start) atomicOr(&s_gb[tid >> 5], 1u << (tid & 31));
    __syncthreads();
    for (int k = 7; k >= 0; k--) {
        const uint32_t v = s_lev[k * RXB + tid];
        if (s_m[tid] && !(v & RXJ_EXIT)) s_m[v] = 1;
        __syncthreads();
    }
    const bool chain = s_m[tid] != 0 && p < B;
    const bool gapn = chain && (nxv & 0x8000u) != 0u;
    {
        const unsigned long long cm = __ballot(chain), gm = __ballot(gapn);
        if ((tid & 63) == 0) { s_cm[tid >> 6] = cm; s_gm[tid >> 6] = gm; }
    }
    if (gapn) {
        const uint32_t len = nxv & 0x7FFFu;
        for (uint32_t k = 0; k < len; k++) { const uint32_t i = (uint32_t)tid + k; atomicOr(&s_gb[i >> 5], 1u << (i & 31)); }
    }
    __syncthreads();
    bool st = chain;
    if (gapn) {
        // a stretch of dropped bytes has ONE start bit: where it begins (behind a match, or at the start of a document)
        uint32_t pgap = pg_in;
        int w = tid >> 6;
        unsigned long long below = s_cm[w] & ((1ull << (tid & 63)) - 1ull);
        while (!below && --w >= 0) below = s_cm[w];
        if (below) { const int l = 63 - __builtin_clzll(below); pgap = (uint32_t)((s_gm[w] >> l) & 1ull); }
        const bool ds = ((a.dstart[p >> 5] >> (p & 31)) & 1u) != 0u;
        st = !pgap || ds;
    }
    const unsigned long long sm = __ballot(st);
    if ((tid & 63) == 0) {
        const uint32_t w0 = wb0 + (uint32_t)(tid >> 5);
        put(a.starts, w0, (uint32_t)sm);
        put(a.starts, w0 + 1, (uint32_t)(sm >> 32));
    }
    if (tid < RXB / 32) put(a.gaps, wb0 + (uint32_t)tid, s_gb[tid]);
}

// The stretches of the two bitmaps that belong to documents split on the host (per-document fallback): patch = per document four words
// {first bitmap word, words, mask of the first word, mask of the last word} followed by `words` start words and `words` gap words; one
// workgroup per document.  Bits outside the document's own byte range (the masks) keep what the device splitter left.
__global__ __launch_bounds__(256) void k_rx_patch(uint32_t* starts, uint32_t* gaps, const uint32_t* patch, const uint32_t* at) {
    const uint32_t* h = patch + at[blockIdx.x];
    const uint32_t w0 = h[0], n = h[1], m_first = h[2], m_last = h[3];
    const uint32_t* ps = h + 4;
    const uint32_t* pg = ps + n;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        uint32_t m = 0xFFFFFFFFu;
        if (k == 0) m &= m_first;
        if (k == n - 1u) m &= m_last;
        if (m == 0xFFFFFFFFu) { starts[w0 + k] = ps[k]; gaps[w0 + k] = pg[k]; }
        else {
            // an edge word: the neighbouring document may be patched by another workgroup at the same time (disjoint masks: atomics, in any order)
            atomicAnd(&starts[w0 + k], ~m); atomicOr(&starts[w0 + k], ps[k] & m);
            atomicAnd(&gaps[w0 + k], ~m); atomicOr(&gaps[w0 + k], pg[k] & m);
        }
    }
}

}  // namespace spl
