// spl_regex.h -- the host splitter for split patterns OTHER than the three the GPU scanner implements.
//
// Tokenizer::new / with_full_options (reference src/core/tokenizer.rs:410-456) compile ANY pattern with a
// third-party regex engine (regexr, or PCRE2 with UTF | UCP: tokenizer.rs:470-488) and encode() walks
// its non-overlapping leftmost-first matches (find_iter, tokenizer.rs:244-257, 729-808: bytes no match covers
// are dropped).  There is no regex engine on the GPU; for such a pattern the split runs here, on the host
// cores, and the chunk boundaries go to the same probe / merge kernels as two bitmaps (chunk starts, gaps).
//
// This is a small backtracking matcher of its own (product code, not shared with the test infrastructure), over the
// code-point classes of splintr_amd/data/unicode_classes.bin -- the table the GPU scanner classifies with.
// Supported: literals, `.`, escapes (\r \n \t \f \v \e \0 \xHH \x{H..} \uHHHH and escaped punctuation),
// \s \S \d \D \w \W, \p{..} / \P{..} for every GENERAL CATEGORY (L Lu Ll Lt Lm Lo L& M Mn Mc Me N Nd Nl No P Pc Pd Ps Pe
// Pi Pf Po S Sm Sc Sk So Z Zs Zl Zp C Cc Cf Cs Co Cn; the ones beyond L* / M / N from the general-category part of a
// version-2 class table), bracket classes with ranges, negation and those escapes inside (caseless under (?i): ASCII
// case pairs, U+017F, U+212A), groups (capturing groups group only), (?: ) (?i: ) (?i) (?> ), alternation, the
// quantifiers ? * + {m} {m,} {m,n} greedy, lazy or POSSESSIVE, the look-aheads (?= ) (?! ), and the assertions
// ^ $ \A \Z \z \b \B (no multi-line mode: a text is one subject).  Semantics are PCRE2's with UTF | UCP (10.39: \d =
// \p{Nd}, \w = [\p{L}\p{N}_]); tests/test_host_regex.py compares thirteen patterns -- upstream tiktoken's cl100k_base
// and o200k_base strings, Qwen2's, GPT-2's among them -- against that engine.  Anything else -- scripts and binary
// properties, back-references, look-behind, \G \K \R \X, \p{Lu} under (?i), non-ASCII ranges under (?i), a pattern that
// can match the empty string -- is refused at construction with the construct named, never approximated.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "spl_tables.h"

namespace spl {

struct RegexProg;
struct RegexDeleter { void operator()(RegexProg* p) const; };
using RegexPtr = std::unique_ptr<RegexProg, RegexDeleter>;

// nullptr on failure, `err` then names the construct and its byte offset in the pattern.
RegexPtr regex_compile(const std::string& pattern, const HostTables& ht, std::string& err);

// The matches of ONE text, text[0, n): for every non-empty match a chunk-start bit at base + start; for every
// stretch no match covers (and behind a match that is followed by one) a start bit where it begins, and gap
// bits over the bytes of stretches that are dropped.  Bits are OR-ed atomically (texts of one batch share
// bitmap words).  Returns false if the step budget of a match attempt ran out (a pathological pattern).
bool regex_split_bits(const RegexProg& prog, const uint8_t* text, size_t n, uint64_t base, uint32_t* start_bits, uint32_t* gap_bits);

// The program as a flat image of 32-bit words for the DEVICE splitter (spl_rx_split.h: the same matcher, one text position per
// lane).  Layout: RX_HDR_WORDS header words -- [0] instructions, [1] word offset of the class sets, [2] sets, [3] offset of the
// first-character filters, [4] filters, [5] offset of the ranges, [6] ranges, [7] 1 if a set tests general categories, [8] offset
// of the first-byte dispatch, [9] class sets with a run bitmap, [10] offset of their list, [11] offset of the alternatives, [12] offset of the two-byte dispatch (0: none) --; from word RX_HDR_WORDS the
// instructions, RX_INST_WORDS words each (op, x, y, f); class sets of RX_SET_WORDS words (class-code bits, general-category bits,
// negated, four words of ASCII membership, first range, ranges, run-bitmap slot or ~0); filters of RX_FIRST_WORDS words (four
// words of ASCII membership, "anything beyond ASCII"); ranges as (first, last) pairs; the first-byte table: for each ASCII byte and (entry
// 128) for any other, a bit per alternative of the top-level alternation whose first-character filter lets the byte in; the two-byte dispatch (byte classes of the first and the second byte, 128 bytes each; the number of second-byte classes; for each pair of classes the alternatives an attempt that begins with such bytes can start); the list of the (at most RX_MAX_RUNSETS) class sets
// that a run instruction repeats: the device matcher tabulates their membership over its text window once per block and takes a
// run as a bit scan; the alternatives of the top-level alternation (count, three words of padding, then four words each: first-
// character filter or ~0, 1 if SIMPLE | its tail kind << 8, first instruction, items | word offset of the items << 16; a simple alternative's items are
// four words each: one-character op, its operand, fewest, most repeats -- see regex_device_image).  Returns false if the program is larger than the device matcher keeps in LDS (RX_IMAGE_MAX_WORDS): such a
// pattern is split on the host.
constexpr uint32_t RX_HDR_WORDS = 16, RX_INST_WORDS = 4, RX_SET_WORDS = 10, RX_FIRST_WORDS = 5, RX_IMAGE_MAX_WORDS = 3072, RX_MAX_RUNSETS = 12;
bool regex_device_image(const RegexProg& prog, std::vector<uint32_t>& words);

// The same as a list of (start, end) pairs (tests).
bool regex_split_spans(const RegexProg& prog, const uint8_t* text, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& out);

}  // namespace spl
