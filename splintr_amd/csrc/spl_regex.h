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
// \s \S, \p{L} \p{Lu} \p{Ll} \p{Lt} \p{Lm} \p{Lo} \p{M} \p{N} and \P{..}, bracket classes with ranges,
// negation and those escapes inside, groups (capturing groups group only), (?: ) (?i: ) (?i), alternation,
// the quantifiers ? * + {m} {m,} {m,n} greedy or lazy, and the look-aheads (?= ) (?! ).  Anything else --
// anchors, \b, \d, \w, other Unicode properties (the class table does not split N or the "other" characters
// further), back-references, look-behind, possessive quantifiers, atomic groups -- is refused at
// construction with the construct named, never approximated.
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

// The same as a list of (start, end) pairs (tests).
bool regex_split_spans(const RegexProg& prog, const uint8_t* text, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& out);

}  // namespace spl
