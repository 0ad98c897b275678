// spl_regex.cpp -- restricted regex compiler + backtracking matcher of the host splitter (spl_regex.h).
#include "spl_regex.h"

#include <array>

#include <algorithm>
#include <cctype>
#include <cstring>

namespace spl {
namespace {

constexpr uint32_t CP_INVALID = 0xFFFFFFFFu;     // a byte sequence that is no character: class "other", equals no literal

// general categories (tools/gen_unicode_tables.py GC_NAMES; class table version 2)
enum : uint32_t { GC_Cn = 0, GC_Lu, GC_Ll, GC_Lt, GC_Lm, GC_Lo, GC_Mn, GC_Mc, GC_Me, GC_Nd, GC_Nl, GC_No, GC_Pc, GC_Pd, GC_Ps, GC_Pe, GC_Pi, GC_Pf,
                  GC_Po, GC_Sm, GC_Sc, GC_Sk, GC_So, GC_Zs, GC_Zl, GC_Zp, GC_Cc, GC_Cf, GC_Cs, GC_Co, GC_COUNT };
constexpr uint32_t GC_ALL = (1u << GC_COUNT) - 1u;

struct ClassSet {
    uint32_t codes = 0;                                   // bit c: every code point of class code c (spl_common.h C_*)
    uint32_t gcs = 0;                                     // bit g: every code point of general category g (properties the class codes do not single out)
    std::vector<std::pair<uint32_t, uint32_t>> ranges;    // inclusive code-point ranges
    bool neg = false;
    uint64_t ascii[2] = {0, 0};                           // membership of U+0000..U+007F, precomputed (regex_compile)
};

enum Op : uint8_t { OP_CHAR, OP_CHAR_FOLD, OP_CLASS, OP_ANY, OP_SPLIT, OP_JMP, OP_MATCH, OP_LOOK, OP_NLOOK, OP_REP1,
                    OP_ATOMIC,      // (?>...) and the possessive quantifiers: sub-program at pc + 1 (ends in MATCH), its FIRST match is final; x = continuation
                    OP_ASSERT };    // x = AS_*: a position test that consumes nothing
enum : uint32_t { AS_BOL, AS_EOL, AS_EOT, AS_WORDB, AS_NWORDB };     // ^ \A | $ \Z | \z | \b | \B
// (the device splitter reads these values from the program image: spl_rx_split.h RXO_* / RXA_*)
static_assert(OP_CHAR == 0 && OP_CHAR_FOLD == 1 && OP_CLASS == 2 && OP_ANY == 3 && OP_SPLIT == 4 && OP_JMP == 5 && OP_MATCH == 6 && OP_LOOK == 7 &&
              OP_NLOOK == 8 && OP_REP1 == 9 && OP_ATOMIC == 10 && OP_ASSERT == 11, "spl_rx_split.h RXO_*");
static_assert(AS_BOL == 0 && AS_EOL == 1 && AS_EOT == 2 && AS_WORDB == 3 && AS_NWORDB == 4, "spl_rx_split.h RXA_*");
struct Inst { Op op; uint32_t x, y; uint32_t f = 0xFFFFFFFFu; };   // f (SPLIT of an alternation): first-character filter of branch x
// which characters can START a match of an alternative: 128 bits for ASCII, one flag for everything else (conservative)
struct FirstSet { uint64_t ascii[2] = {0, 0}; bool other = false; };                  // CHAR: x = cp; CLASS: x = set; SPLIT: x first, y second; JMP: x;
                                                          // LOOK / NLOOK: sub-program at pc + 1 (ends in MATCH), x = continuation
                                                          // REP1: greedy x..y (y == UINT32_MAX: unbounded) repeats of the ONE-character
                                                          //       instruction at pc + 1; goes on at pc + 2

struct Node {
    enum Kind { CHAR, CLASS, ANY, CAT, ALT, REP, LOOK, EMPTY, ATOMIC, ASSERT } kind = EMPTY;
    uint32_t as = 0;                                     // ASSERT: AS_*
    bool possessive = false;                             // REP of a one-character item: no way back into the run
    uint32_t cp = 0; bool fold = false;                  // CHAR
    uint32_t set = 0;                                    // CLASS
    std::vector<std::unique_ptr<Node>> kids;             // CAT / ALT; REP, LOOK: one
    uint32_t lo = 0, hi = 0; bool lazy = false;          // REP: hi == UINT32_MAX: unbounded
    bool neg = false;                                    // LOOK
};

}  // namespace

struct RegexProg {
    std::vector<Inst> code;
    std::vector<ClassSet> sets;
    std::vector<FirstSet> firsts;
    const HostTables* ht = nullptr;
};
void RegexDeleter::operator()(RegexProg* p) const { delete p; }

namespace {

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    RegexProg& prog;
    Parser(const std::string& p, RegexProg& pr) : s(p), prog(pr) {}

    bool fail(const std::string& what, size_t at) {
        if (err.empty()) err = "unsupported or malformed construct in the split pattern at byte " + std::to_string(at) + ": " + what;
        return false;
    }
    bool eof() const { return i >= s.size(); }

    // one code point of the PATTERN text (which is valid UTF-8: it comes from a str)
    bool next_cp(uint32_t& cp) {
        const uint8_t b = (uint8_t)s[i];
        uint32_t len = b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
        if (b >= 0x80 && b < 0xC0) return fail("stray UTF-8 continuation byte in the pattern", i);
        if (i + len > s.size()) return fail("truncated UTF-8 in the pattern", i);
        cp = len == 1 ? b : b & (0xFFu >> (len + 1));
        for (uint32_t k = 1; k < len; k++) cp = (cp << 6) | ((uint8_t)s[i + k] & 0x3Fu);
        i += len;
        return true;
    }

    bool property(ClassSet& cs, bool negate_item, size_t at) {            // after \p or \P: {Name} or a single letter
        std::string name;
        if (!eof() && s[i] == '{') {
            const size_t e = s.find('}', i);
            if (e == std::string::npos) return fail("\\p{ without }", at);
            name = s.substr(i + 1, e - i - 1);
            i = e + 1;
        } else if (!eof()) name = std::string(1, s[i++]);
        bool inner_neg = false;
        if (!name.empty() && name[0] == '^') { inner_neg = true; name = name.substr(1); }
        uint32_t m = 0, g = 0;
        bool by_code = true;
        if (name == "L" || name == "Letter") m = M_L;
        else if (name == "Lu") m = SPL_BIT(C_LU);
        else if (name == "Ll") m = SPL_BIT(C_LL);
        else if (name == "Lt") m = SPL_BIT(C_LT);
        else if (name == "Lm") m = SPL_BIT(C_LM);
        else if (name == "Lo") m = SPL_BIT(C_LO);
        else if (name == "L&" || name == "Lc") m = SPL_BIT(C_LU) | SPL_BIT(C_LL) | SPL_BIT(C_LT);
        else if (name == "M" || name == "Mark") m = SPL_BIT(C_M);
        else if (name == "N" || name == "Number") m = SPL_BIT(C_N);
        else {
            // properties the class codes do not single out: by general category (class table version 2)
            by_code = false;
            static const struct { const char* n; uint32_t g; } GCS[] = {
                {"Mn", 1u << GC_Mn}, {"Mc", 1u << GC_Mc}, {"Me", 1u << GC_Me}, {"Nd", 1u << GC_Nd}, {"Nl", 1u << GC_Nl}, {"No", 1u << GC_No},
                {"Pc", 1u << GC_Pc}, {"Pd", 1u << GC_Pd}, {"Ps", 1u << GC_Ps}, {"Pe", 1u << GC_Pe}, {"Pi", 1u << GC_Pi}, {"Pf", 1u << GC_Pf},
                {"Po", 1u << GC_Po}, {"P", (1u << GC_Pc) | (1u << GC_Pd) | (1u << GC_Ps) | (1u << GC_Pe) | (1u << GC_Pi) | (1u << GC_Pf) | (1u << GC_Po)},
                {"Sm", 1u << GC_Sm}, {"Sc", 1u << GC_Sc}, {"Sk", 1u << GC_Sk}, {"So", 1u << GC_So},
                {"S", (1u << GC_Sm) | (1u << GC_Sc) | (1u << GC_Sk) | (1u << GC_So)},
                {"Zs", 1u << GC_Zs}, {"Zl", 1u << GC_Zl}, {"Zp", 1u << GC_Zp}, {"Z", (1u << GC_Zs) | (1u << GC_Zl) | (1u << GC_Zp)},
                {"Cc", 1u << GC_Cc}, {"Cf", 1u << GC_Cf}, {"Cs", 1u << GC_Cs}, {"Co", 1u << GC_Co}, {"Cn", 1u << GC_Cn},
                {"C", (1u << GC_Cc) | (1u << GC_Cf) | (1u << GC_Cs) | (1u << GC_Co) | (1u << GC_Cn)}};
            for (const auto& e : GCS) if (name == e.n) g = e.g;
            if (!g) {
                // a SCRIPT (\p{Han}, \p{Hiragana}, \p{Latin}, \p{Script=Cyrillic}, \p{sc=Greek} ...): the ranges the class table carries for it
                // (version 3: probed from the engine the table stands for), added to the set -- complemented for \P{..} / \p{^..}
                std::string sn = name;
                for (const char* pre : {"Script=", "script=", "sc=", "Is"})
                    if (sn.compare(0, strlen(pre), pre) == 0 && sn.size() > strlen(pre)) { sn = sn.substr(strlen(pre)); break; }
                auto norm = [](const std::string& x) {
                    std::string r;
                    for (char ch : x) if (ch != '_' && ch != ' ' && ch != '-') r += (char)std::tolower((unsigned char)ch);
                    return r;
                };
                const HostTables::Script* found = nullptr;
                for (const auto& sc : prog.ht->scripts) if (norm(sc.name) == norm(sn)) found = &sc;
                if (found) {
                    if (negate_item != inner_neg) {
                        uint32_t next = 0;
                        for (const auto& rg : found->ranges) {                  // (ascending, disjoint)
                            if (rg.first > next) cs.ranges.emplace_back(next, rg.first - 1u);
                            next = rg.second + 1u;
                        }
                        if (next <= 0x10FFFFu) cs.ranges.emplace_back(next, 0x10FFFFu);
                    } else for (const auto& rg : found->ranges) cs.ranges.push_back(rg);
                    return true;
                }
                return fail("\\p{" + name + "}: general categories (L, Lu .. Lo, L&, M, Mn .. Me, N, Nd .. No, P, Pc .. Po, S, Sm .. So, Z, Zs .. Zp, C, Cc .. Cn) and "
                                    + (prog.ht->scripts.empty() ? std::string("-- with a version-3 class table -- ") : std::string()) +
                                    "scripts (Han, Hiragana, Katakana, Hangul, Latin, Cyrillic ...) are the properties implemented; binary properties and "
                                    "script extensions are not", at);
            }
            if (prog.ht->gc_stage1.empty())
                return fail("\\p{" + name + "} needs the general-category table (a version-2 class table: splintr_amd/data/unicode_classes.bin)", at);
        }
        if (by_code) {
            if (negate_item != inner_neg) m = ~m & ((1u << C_EOT) - 1u);
            cs.codes |= m;
        } else {
            if (negate_item != inner_neg) g = ~g & GC_ALL;
            cs.gcs |= g;
        }
        return true;
    }

    // escape after the backslash.  in_class: inside [...].  Either a single code point (is_cp) or a class item
    // added to `cs`.
    bool escape(bool in_class, bool& is_cp, uint32_t& cp, ClassSet& cs, size_t at) {
        if (eof()) return fail("pattern ends in a backslash", at);
        const char c = s[i++];
        is_cp = true;
        switch (c) {
            case 'r': cp = '\r'; return true;
            case 'n': cp = '\n'; return true;
            case 't': cp = '\t'; return true;
            case 'f': cp = '\f'; return true;
            case 'v': cp = 0x0B; return true;
            case 'e': cp = 0x1B; return true;
            case 'a': cp = 0x07; return true;
            case '0': cp = 0; return true;
            case 'x': case 'u': {
                uint32_t v = 0; int nd = 0;
                if (!eof() && s[i] == '{') {
                    const size_t e = s.find('}', i);
                    if (e == std::string::npos) return fail("\\x{ without }", at);
                    for (size_t k = i + 1; k < e; k++, nd++) {
                        const int h = std::isxdigit((unsigned char)s[k]) ? (std::isdigit((unsigned char)s[k]) ? s[k] - '0' : (s[k] | 0x20) - 'a' + 10) : -1;
                        if (h < 0) return fail("bad hex digit in \\x{...}", at);
                        v = v * 16 + (uint32_t)h;
                    }
                    i = e + 1;
                } else {
                    const int want = c == 'x' ? 2 : 4;
                    for (; nd < want && !eof() && std::isxdigit((unsigned char)s[i]); nd++, i++)
                        v = v * 16 + (uint32_t)(std::isdigit((unsigned char)s[i]) ? s[i] - '0' : (s[i] | 0x20) - 'a' + 10);
                    if (nd != want) return fail("\\x / \\u need 2 / 4 hex digits", at);
                }
                if (nd == 0 || v > 0x10FFFF) return fail("code point out of range", at);
                cp = v;
                return true;
            }
            case 'd': case 'D': {                                        // \d = \p{Nd} under UCP
                is_cp = false;
                if (prog.ht->gc_stage1.empty()) return fail("\\d needs the general-category table (a version-2 class table)", at);
                cs.gcs |= c == 'd' ? (1u << GC_Nd) : (~(1u << GC_Nd) & GC_ALL);
                return true;
            }
            case 'w':                                                    // \w = [\p{L}\p{N}_] (PCRE2 10.39 with UCP)
                is_cp = false; cs.codes |= M_L | SPL_BIT(C_N); cs.ranges.emplace_back('_', '_'); return true;
            case 'W':
                if (in_class) return fail("\\W inside a bracket class", at);
                is_cp = false; cs.codes |= M_L | SPL_BIT(C_N); cs.ranges.emplace_back('_', '_'); cs.neg = true; return true;
            case 'b': if (in_class) { cp = 0x08; return true; } break;  // (outside a class: the word boundary, atom())
            case 's': is_cp = false; cs.codes |= M_S; return true;
            case 'S': is_cp = false; cs.codes |= ~M_S & ((1u << C_EOT) - 1u); return true;
            case 'p': is_cp = false; return property(cs, false, at);
            case 'P': is_cp = false; return property(cs, true, at);
            default: break;
        }
        if (std::isalnum((unsigned char)c))
            return fail(std::string("\\") + c + " (back-references, \\G, \\K, \\R, \\X, \\h, \\N and the other letter escapes are not implemented)", at);
        if ((uint8_t)c >= 0x80) { i--; return next_cp(cp); }            // an escaped non-ASCII character: itself
        cp = (uint8_t)c;                                                // escaped punctuation
        (void)in_class;
        return true;
    }

    std::unique_ptr<Node> char_node(uint32_t cp, bool fold) {
        auto n = std::make_unique<Node>();
        n->kind = Node::CHAR; n->cp = cp;
        n->fold = fold && cp < 0x80 && std::isalpha((int)cp);
        return n;
    }
    std::unique_ptr<Node> set_node(ClassSet&& cs) {
        auto n = std::make_unique<Node>();
        n->kind = Node::CLASS; n->set = (uint32_t)prog.sets.size();
        prog.sets.push_back(std::move(cs));
        return n;
    }

    std::unique_ptr<Node> bracket(bool fold, size_t at) {                // after '['
        ClassSet cs;
        if (!eof() && s[i] == '^') { cs.neg = true; i++; }
        bool first = true;
        for (;;) {
            if (eof()) { fail("[ without ]", at); return nullptr; }
            if (s[i] == ']' && !first) { i++; break; }
            first = false;
            if (s[i] == '[' && i + 1 < s.size() && s[i + 1] == ':') { fail("POSIX classes [:name:]", i); return nullptr; }
            uint32_t lo;
            bool is_cp = true;
            const size_t item_at = i;
            if (s[i] == '\\') { i++; if (!escape(true, is_cp, lo, cs, item_at)) return nullptr; }
            else if (!next_cp(lo)) return nullptr;
            if (!is_cp) continue;
            uint32_t hi = lo;
            if (i + 1 < s.size() && s[i] == '-' && s[i + 1] != ']') {
                i++;
                bool hi_cp = true;
                ClassSet dummy;
                if (s[i] == '\\') { i++; if (!escape(true, hi_cp, hi, dummy, item_at)) return nullptr; if (!hi_cp) { fail("a class escape as the end of a range", item_at); return nullptr; } }
                else if (!next_cp(hi)) return nullptr;
                if (hi < lo) { fail("range out of order", item_at); return nullptr; }
            }
            if (fold) {
                // caseless: the other case of every ASCII letter of the range, U+017F for s, U+212A for k (what PCRE2's
                // UTF | UCP caseless matching adds for ASCII letters; ranges beyond ASCII would need the full case folding)
                if (hi >= 0x80) { fail("non-ASCII characters in a bracket class under (?i)", item_at); return nullptr; }
                for (uint32_t c = lo; c <= hi; c++)
                    if (std::isalpha((int)c)) {
                        cs.ranges.emplace_back(c ^ 0x20u, c ^ 0x20u);
                        if ((c | 0x20u) == 's') cs.ranges.emplace_back(0x17F, 0x17F);
                        if ((c | 0x20u) == 'k') cs.ranges.emplace_back(0x212A, 0x212A);
                    }
            }
            cs.ranges.emplace_back(lo, hi);
        }
        if (!case_props_ok(cs, fold, at)) return nullptr;
        return set_node(std::move(cs));
    }
    // \p{Lu} \p{Ll} \p{Lt} under (?i): what caseless matching does to them differs between engines and versions -- refused
    bool case_props_ok(const ClassSet& cs, bool fold, size_t at) {
        const uint32_t k = cs.codes & (SPL_BIT(C_LU) | SPL_BIT(C_LL) | SPL_BIT(C_LT));
        if (fold && k != 0 && k != (SPL_BIT(C_LU) | SPL_BIT(C_LL) | SPL_BIT(C_LT))) return fail("\\p{Lu} / \\p{Ll} / \\p{Lt} under (?i)", at);
        return true;
    }
    std::unique_ptr<Node> assert_node(uint32_t as) {
        auto n = std::make_unique<Node>();
        n->kind = Node::ASSERT; n->as = as;
        return n;
    }

    std::unique_ptr<Node> atom(bool& fold) {
        const size_t at = i;
        const char c = s[i];
        if (c == '(') {
            i++;
            bool sub_fold = fold;
            bool look = false, neg = false;
            if (!eof() && s[i] == '?') {
                i++;
                if (eof()) { fail("(? at the end", at); return nullptr; }
                if (s[i] == ':') i++;
                else if (s[i] == '=') { i++; look = true; }
                else if (s[i] == '!') { i++; look = true; neg = true; }
                else if (s[i] == 'i' && i + 1 < s.size() && s[i + 1] == ':') { i += 2; sub_fold = true; }
                else if (s[i] == 'i' && i + 1 < s.size() && s[i + 1] == ')') {     // (?i): the rest of the enclosing group
                    i += 2;
                    fold = true;
                    auto n = std::make_unique<Node>();
                    n->kind = Node::EMPTY;
                    return n;
                }
                else if (s[i] == '<' && i + 1 < s.size() && (s[i + 1] == '=' || s[i + 1] == '!')) { fail("look-behind", at); return nullptr; }
                else if (s[i] == '>') {                                            // atomic group: its first match is final
                    i++;
                    auto inner = alternation(sub_fold);
                    if (!inner) return nullptr;
                    if (eof() || s[i] != ')') { fail("( without )", at); return nullptr; }
                    i++;
                    auto n = std::make_unique<Node>();
                    n->kind = Node::ATOMIC;
                    n->kids.push_back(std::move(inner));
                    return n;
                }
                else if (s[i] == '<' || s[i] == 'P' || s[i] == '\'') {             // named group: groups only
                    const char close = s[i] == '\'' ? '\'' : '>';
                    const size_t e = s.find(close, i + 1);
                    if (e == std::string::npos) { fail("unterminated group name", at); return nullptr; }
                    i = e + 1;
                }
                else { fail(std::string("group option (?") + s[i], at); return nullptr; }
            }
            auto inner = alternation(sub_fold);
            if (!inner) return nullptr;
            if (eof() || s[i] != ')') { fail("( without )", at); return nullptr; }
            i++;
            if (look) {
                auto n = std::make_unique<Node>();
                n->kind = Node::LOOK; n->neg = neg;
                n->kids.push_back(std::move(inner));
                return n;
            }
            return inner;
        }
        if (c == '[') { i++; return bracket(fold, at); }
        if (c == '.') {
            i++;
            auto n = std::make_unique<Node>();
            n->kind = Node::ANY;
            return n;
        }
        if (c == '^') { i++; return assert_node(AS_BOL); }                // (no multi-line mode: the start / end of the text)
        if (c == '$') { i++; return assert_node(AS_EOL); }
        if (c == '*' || c == '+' || c == '?' || c == '{') { fail(std::string("quantifier ") + c + " without an operand", at); return nullptr; }
        if (c == '\\') {
            i++;
            if (!eof()) {
                const char e = s[i];
                const int as = e == 'b' ? (int)AS_WORDB : e == 'B' ? (int)AS_NWORDB : e == 'A' ? (int)AS_BOL : e == 'Z' ? (int)AS_EOL : e == 'z' ? (int)AS_EOT : -1;
                if (as >= 0) { i++; return assert_node((uint32_t)as); }
            }
            bool is_cp;
            uint32_t cp = 0;
            ClassSet cs;
            if (!escape(false, is_cp, cp, cs, at)) return nullptr;
            if (is_cp) return char_node(cp, fold);
            if (!case_props_ok(cs, fold, at)) return nullptr;
            return set_node(std::move(cs));
        }
        uint32_t cp;
        if (!next_cp(cp)) return nullptr;
        if (fold && cp >= 0x80) { fail("non-ASCII literal under (?i) (only ASCII case pairs, U+017F and U+212A are folded)", at); return nullptr; }
        return char_node(cp, fold);
    }

    static bool nullable(const Node& n) {
        switch (n.kind) {
            case Node::CHAR: case Node::CLASS: case Node::ANY: return false;
            case Node::EMPTY: case Node::LOOK: case Node::ASSERT: return true;
            case Node::ATOMIC: return nullable(*n.kids[0]);
            case Node::CAT: for (auto& k : n.kids) if (!nullable(*k)) return false; return true;
            case Node::ALT: for (auto& k : n.kids) if (nullable(*k)) return true; return false;
            case Node::REP: return n.lo == 0 || nullable(*n.kids[0]);
        }
        return true;
    }

    std::unique_ptr<Node> repeat(bool& fold) {
        auto a = atom(fold);
        if (!a) return nullptr;
        while (!eof()) {
            const size_t at = i;
            uint32_t lo, hi;
            const char c = s[i];
            if (c == '?') { lo = 0; hi = 1; i++; }
            else if (c == '*') { lo = 0; hi = UINT32_MAX; i++; }
            else if (c == '+') { lo = 1; hi = UINT32_MAX; i++; }
            else if (c == '{') {
                size_t k = i + 1;
                auto num = [&](uint32_t& v) { size_t k0 = k; v = 0; while (k < s.size() && std::isdigit((unsigned char)s[k]) && v < 100000) v = v * 10 + (uint32_t)(s[k++] - '0'); return k > k0; };
                if (!num(lo)) break;                                     // a literal '{' (as in PCRE2): handled as a character below
                hi = lo;
                if (k < s.size() && s[k] == ',') { k++; if (!num(hi)) hi = UINT32_MAX; }
                if (k >= s.size() || s[k] != '}') break;
                i = k + 1;
                if (hi < lo) { fail("{m,n} with n < m", at); return nullptr; }
                if (lo > 1000 || (hi != UINT32_MAX && hi > 1000)) { fail("counted repeat beyond 1000", at); return nullptr; }
            } else break;
            bool lazy = false, possessive = false;
            if (!eof() && s[i] == '?') { lazy = true; i++; }
            else if (!eof() && s[i] == '+') { possessive = true; i++; }   // X*+ == (?>X*): what the run took stays taken
            if (a->kind == Node::EMPTY || a->kind == Node::LOOK || a->kind == Node::ASSERT) { fail("a quantifier on an assertion", at); return nullptr; }
            if (hi == UINT32_MAX && nullable(*a)) { fail("an unbounded quantifier over an expression that can match the empty string", at); return nullptr; }
            const bool one_char = a->kind == Node::CHAR || a->kind == Node::CLASS || a->kind == Node::ANY;
            auto r = std::make_unique<Node>();
            r->kind = Node::REP; r->lo = lo; r->hi = hi; r->lazy = lazy;
            r->possessive = possessive && one_char;
            r->kids.push_back(std::move(a));
            a = std::move(r);
            if (possessive && !one_char) {
                auto g = std::make_unique<Node>();
                g->kind = Node::ATOMIC;
                g->kids.push_back(std::move(a));
                a = std::move(g);
            }
        }
        if (!eof() && s[i] == '{') {                                     // not a quantifier: the caller's next atom reads it as a literal
        }
        return a;
    }

    std::unique_ptr<Node> concat(bool& fold) {
        auto n = std::make_unique<Node>();
        n->kind = Node::CAT;
        while (!eof() && s[i] != '|' && s[i] != ')') {
            std::unique_ptr<Node> r;
            if (s[i] == '{') {                                           // literal brace (no valid quantifier follows an atom here)
                i++;
                r = char_node('{', fold);
            } else r = repeat(fold);
            if (!r) return nullptr;
            n->kids.push_back(std::move(r));
        }
        return n;
    }

    std::unique_ptr<Node> alternation(bool fold) {                       // (by value: an inline (?i) lasts to the end of ITS group)
        auto first = concat(fold);
        if (!first) return nullptr;
        if (eof() || s[i] != '|') return first;
        auto n = std::make_unique<Node>();
        n->kind = Node::ALT;
        n->kids.push_back(std::move(first));
        while (!eof() && s[i] == '|') {
            i++;
            auto k = concat(fold);
            if (!k) return nullptr;
            n->kids.push_back(std::move(k));
        }
        return n;
    }
};

struct Emitter {
    RegexProg& p;
    bool too_big = false;
    uint32_t emit(Op op, uint32_t x = 0, uint32_t y = 0) {
        if (p.code.size() > 200000) too_big = true;
        p.code.push_back(Inst{op, x, y});
        return (uint32_t)p.code.size() - 1;
    }
    // The characters a match of `n` can start with, OR-ed into fs; returns whether n can match without consuming one
    // (then whatever follows it contributes too).  Exact for ASCII, one conservative flag for everything beyond.
    bool first_of(const Node& n, FirstSet& fs) const {
        auto add = [&](uint32_t c) { fs.ascii[c >> 6] |= 1ull << (c & 63); };
        switch (n.kind) {
            case Node::EMPTY: case Node::LOOK: case Node::ASSERT: return true;
            case Node::ATOMIC: return first_of(*n.kids[0], fs);
            case Node::CHAR:
                if (n.cp < 0x80) { add(n.cp); if (n.fold) { add(n.cp ^ 0x20u); fs.other = true; } }      // (fold: U+017F, U+212A)
                else fs.other = true;
                return false;
            case Node::ANY:
                fs.ascii[0] = ~0ull & ~(1ull << '\n'); fs.ascii[1] = ~0ull; fs.other = true;
                return false;
            case Node::CLASS: {
                const ClassSet& cs = p.sets[n.set];
                for (uint32_t c = 0; c < 128; c++) {
                    const uint32_t cls = p.ht->ucls_stage2[((uint32_t)p.ht->ucls_stage1[0] << p.ht->ucls_shift) | c];
                    bool in = ((cs.codes >> cls) & 1u) != 0;
                    if (!in && cs.gcs) in = ((cs.gcs >> host_cp_category(*p.ht, c)) & 1u) != 0;
                    for (const auto& r : cs.ranges) in = in || (c >= r.first && c <= r.second);
                    if (in != cs.neg) add(c);
                }
                fs.other = true;
                return false;
            }
            case Node::CAT:
                for (const auto& k : n.kids) if (!first_of(*k, fs)) return false;
                return true;
            case Node::ALT: {
                bool nul = false;
                for (const auto& k : n.kids) nul = first_of(*k, fs) || nul;
                return nul;
            }
            case Node::REP: {
                const bool nul = first_of(*n.kids[0], fs);
                return nul || n.lo == 0;
            }
        }
        return true;
    }
    void gen(const Node& n) {
        if (too_big) return;
        switch (n.kind) {
            case Node::EMPTY: break;
            case Node::CHAR: emit(n.fold ? OP_CHAR_FOLD : OP_CHAR, n.cp); break;
            case Node::CLASS: emit(OP_CLASS, n.set); break;
            case Node::ANY: emit(OP_ANY); break;
            case Node::CAT: for (auto& k : n.kids) gen(*k); break;
            case Node::ALT: {
                std::vector<uint32_t> jumps;
                for (size_t k = 0; k < n.kids.size(); k++) {
                    if (k + 1 < n.kids.size()) {
                        const uint32_t sp = emit(OP_SPLIT);
                        p.code[sp].x = sp + 1;
                        FirstSet fs;
                        if (!first_of(*n.kids[k], fs)) {                // (an alternative that can match the empty string is always tried)
                            p.code[sp].f = (uint32_t)p.firsts.size();
                            p.firsts.push_back(fs);
                        }
                        gen(*n.kids[k]);
                        jumps.push_back(emit(OP_JMP));
                        p.code[sp].y = (uint32_t)p.code.size();
                    } else gen(*n.kids[k]);
                }
                for (uint32_t j : jumps) p.code[j].x = (uint32_t)p.code.size();
                break;
            }
            case Node::REP: {
                const Node& e = *n.kids[0];
                if (!n.lazy && (e.kind == Node::CHAR || e.kind == Node::CLASS || e.kind == Node::ANY)) {
                    // a greedy run of ONE-character items (\p{L}+, \s*, [\r\n]*, ' ?'): taken in one go with ONE way back
                    // on the stack (give one character back, re-counted from the run's start) instead of one per character;
                    // possessive (f == 1): no way back at all
                    const uint32_t r1 = emit(OP_REP1, n.lo, n.hi);
                    if (n.possessive) p.code[r1].f = 1u;
                    gen(e);
                    break;
                }
                for (uint32_t k = 0; k < n.lo; k++) gen(e);
                if (n.hi == UINT32_MAX) {                               // e*: L0: SPLIT L1, end; L1: e; JMP L0
                    const uint32_t l0 = emit(OP_SPLIT);
                    gen(e);
                    emit(OP_JMP, l0);
                    const uint32_t end = (uint32_t)p.code.size();
                    p.code[l0].x = n.lazy ? end : l0 + 1;
                    p.code[l0].y = n.lazy ? l0 + 1 : end;
                } else {                                                // (e (e (e)?)?)?: every SPLIT's way out is the common end
                    std::vector<uint32_t> splits;
                    for (uint32_t k = n.lo; k < n.hi; k++) { splits.push_back(emit(OP_SPLIT)); gen(e); }
                    const uint32_t end = (uint32_t)p.code.size();
                    for (uint32_t sp : splits) {
                        p.code[sp].x = n.lazy ? end : sp + 1;
                        p.code[sp].y = n.lazy ? sp + 1 : end;
                    }
                }
                break;
            }
            case Node::ASSERT: emit(OP_ASSERT, n.as); break;
            case Node::ATOMIC: {
                const uint32_t l = emit(OP_ATOMIC);
                gen(*n.kids[0]);
                emit(OP_MATCH);
                p.code[l].x = (uint32_t)p.code.size();
                break;
            }
            case Node::LOOK: {
                const uint32_t l = emit(n.neg ? OP_NLOOK : OP_LOOK);
                gen(*n.kids[0]);
                emit(OP_MATCH);
                p.code[l].x = (uint32_t)p.code.size();
                break;
            }
        }
    }
};

// ---- matching -------------------------------------------------------------------------------------------
struct Ch { uint32_t cp, len, cls; };

// One character of the TEXT (raw bytes; the kernels' policy for bytes that are no UTF-8, include/splintr_hip.h:
// a lead byte takes the continuation bytes actually present, at most as many as it announces -- complete: the
// decoded value, otherwise ONE character of class "other"; a stray continuation byte is such a character itself).
inline Ch decode(const HostTables& ht, const uint8_t* t, size_t pos, size_t n) {
    const uint32_t b = t[pos];
    if (b < 0x80) return Ch{b, 1, (uint32_t)ht.ucls_stage2[((uint32_t)ht.ucls_stage1[0] << ht.ucls_shift) | b]};
    if (b < 0xC0) return Ch{CP_INVALID, 1, C_P};
    const uint32_t want = b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4;
    uint32_t len = 1;
    while (len < want && pos + len < n && (t[pos + len] & 0xC0u) == 0x80u) len++;
    if (len != want) return Ch{CP_INVALID, len, C_P};
    uint32_t cp = b & (0xFFu >> (want + 1));
    for (uint32_t k = 1; k < want; k++) cp = (cp << 6) | (t[pos + k] & 0x3Fu);
    return Ch{cp, want, host_cp_class(ht, cp)};
}

inline bool in_set(const HostTables& ht, const ClassSet& cs, const Ch& c) {
    bool in = ((cs.codes >> c.cls) & 1u) != 0;
    // (bytes that are no character: general category Cn, as they are class "other")
    if (!in && cs.gcs) in = ((cs.gcs >> (c.cp == CP_INVALID ? (uint32_t)GC_Cn : host_cp_category(ht, c.cp))) & 1u) != 0;
    if (!in && c.cp != CP_INVALID)
        for (const auto& r : cs.ranges) if (c.cp >= r.first && c.cp <= r.second) { in = true; break; }
    return in != cs.neg;
}

// caseless partners of an ASCII letter under Unicode simple case folding: the other case, U+017F for s, U+212A for k
inline bool fold_eq(uint32_t pat, uint32_t cp) {
    const uint32_t lo = pat | 0x20u;
    if (cp < 0x80) return (cp | 0x20u) == lo && std::isalpha((int)cp);
    return (lo == 's' && cp == 0x17F) || (lo == 'k' && cp == 0x212A);
}

struct Matcher {
    const RegexProg& p;
    const uint8_t* t;
    size_t n;
    uint64_t steps = 0;
    uint64_t step_max = 0;                                               // per match attempt
    struct Frame { uint32_t pc, k; size_t pos; };                        // k == NO_K: go on at pc; else: REP1 at pc, retry with k characters
    static constexpr uint32_t NO_K = 0xFFFFFFFFu;
    std::vector<Frame> stack;

    bool one_ascii(const Inst& in, uint32_t b) const {                   // ... an ASCII byte (no decoding, no class lookup)
        if (in.op == OP_CHAR) return b == in.x;
        if (in.op == OP_CHAR_FOLD) return (b | 0x20u) == (in.x | 0x20u) && ((b | 0x20u) - 'a') < 26u;
        if (in.op == OP_ANY) return b != '\n';
        return ((p.sets[in.x].ascii[b >> 6] >> (b & 63)) & 1ull) != 0;
    }
    bool one(const Inst& in, const Ch& c) const {                        // does the one-character instruction take c?
        if (in.op == OP_CHAR) return c.cp == in.x;
        if (in.op == OP_CHAR_FOLD) return c.cp != CP_INVALID && fold_eq(in.x, c.cp);
        if (in.op == OP_ANY) return c.cp != '\n';
        return in_set(*p.ht, p.sets[in.x], c);
    }
    bool is_word(size_t q) const {                                       // \w at q: a letter, a number or '_'
        const Ch c = decode(*p.ht, t, q, n);
        return c.cp == '_' || ((SPL_BIT(c.cls) & (M_L | SPL_BIT(C_N))) != 0 && c.cp != CP_INVALID);
    }
    // longest-by-priority match of the program that starts at `pc0`, anchored at `pos`; SIZE_MAX: no match
    size_t run(uint32_t pc0, size_t pos0) {
        const size_t floor = stack.size();
        stack.push_back(Frame{pc0, NO_K, pos0});
        while (stack.size() > floor) {
            uint32_t pc = stack.back().pc;
            size_t pos = stack.back().pos;
            const uint32_t retry_k = stack.back().k;
            stack.pop_back();
            if (retry_k != NO_K) {                                        // a REP1 gives a character back: k of them from the run's start
                const Inst& in = p.code[pc];
                if (retry_k > in.x) stack.push_back(Frame{pc, retry_k - 1, pos});
                for (uint32_t k = 0; k < retry_k; k++) pos += t[pos] < 0x80 ? 1u : decode(*p.ht, t, pos, n).len;
                steps += retry_k;
                pc += 2;
            }
            for (;;) {
                if (++steps > step_max) { stack.resize(floor); return SIZE_MAX - 1; }
                const Inst& in = p.code[pc];
                if (in.op == OP_MATCH) { stack.resize(floor); return pos; }
                if (in.op == OP_JMP) { pc = in.x; continue; }
                if (in.op == OP_SPLIT) {
                    if (in.f != 0xFFFFFFFFu) {                            // an alternative that cannot start with this character: the next one
                        bool can = false;
                        if (pos < n) {
                            const uint32_t b = t[pos];
                            const FirstSet& fs = p.firsts[in.f];
                            can = b < 0x80 ? ((fs.ascii[b >> 6] >> (b & 63)) & 1ull) != 0 : fs.other;
                        }
                        if (!can) { pc = in.y; continue; }
                    }
                    stack.push_back(Frame{in.y, NO_K, pos});
                    pc = in.x;
                    continue;
                }
                if (in.op == OP_LOOK || in.op == OP_NLOOK) {
                    const size_t r = run(pc + 1, pos);
                    if (r == SIZE_MAX - 1) { stack.resize(floor); return r; }
                    if ((r != SIZE_MAX) == (in.op == OP_LOOK)) { pc = in.x; continue; }
                    break;
                }
                if (in.op == OP_ATOMIC) {                                  // the sub-program's FIRST match, and no way back into it
                    const size_t r = run(pc + 1, pos);
                    if (r == SIZE_MAX - 1) { stack.resize(floor); return r; }
                    if (r == SIZE_MAX) break;
                    pos = r; pc = in.x;
                    continue;
                }
                if (in.op == OP_ASSERT) {
                    bool ok;
                    if (in.x == AS_BOL) ok = pos == 0;
                    else if (in.x == AS_EOT) ok = pos == n;
                    else if (in.x == AS_EOL) ok = pos == n || (pos + 1 == n && t[pos] == '\n');
                    else {
                        const bool after = pos < n && is_word(pos);
                        bool before = false;
                        if (pos > 0) {
                            size_t q = pos - 1;
                            while (q > 0 && pos - q < 4 && (t[q] & 0xC0u) == 0x80u) q--;
                            if (q + decode(*p.ht, t, q, n).len == pos) before = is_word(q);     // (else: a stray byte, no word character)
                        }
                        ok = (before != after) == (in.x == AS_WORDB);
                    }
                    if (!ok) break;
                    pc++;
                    continue;
                }
                if (in.op == OP_REP1) {
                    const Inst& a = p.code[pc + 1];
                    size_t q = pos;
                    uint32_t k = 0;
                    while (k < in.y && q < n) {
                        if (t[q] < 0x80) {
                            if (!one_ascii(a, t[q])) break;
                            q++;
                        } else {
                            const Ch c = decode(*p.ht, t, q, n);
                            if (!one(a, c)) break;
                            q += c.len;
                        }
                        k++;
                    }
                    steps += k;
                    if (k < in.x) break;
                    if (k > in.x && in.f != 1u) stack.push_back(Frame{pc, k - 1, pos});
                    pos = q;
                    pc += 2;
                    continue;
                }
                if (pos >= n) break;
                if (t[pos] < 0x80) {
                    if (!one_ascii(in, t[pos])) break;
                    pos++;
                } else {
                    const Ch c = decode(*p.ht, t, pos, n);
                    if (!one(in, c)) break;
                    pos += c.len;
                }
                pc++;
            }
        }
        return SIZE_MAX;
    }
};

template <class F> bool split_text(const RegexProg& prog, const uint8_t* text, size_t n, F on_span) {
    Matcher m{prog, text, n};
    m.step_max = 64ull * n + 1'000'000ull;                                // (linear for sane patterns; a safety net for the others)
    size_t pos = 0;
    while (pos < n) {
        m.steps = 0;
        const size_t e = m.run(0, pos);
        if (e == SIZE_MAX - 1) return false;
        if (e == SIZE_MAX || e == pos) {                                 // no match here / an empty one: the character is skipped
            pos += decode(*prog.ht, text, pos, n).len;
            continue;
        }
        on_span(pos, e);
        pos = e;
    }
    return true;
}

inline void or_bit(uint32_t* bm, uint64_t pos) { __atomic_fetch_or(&bm[pos >> 5], 1u << (pos & 31), __ATOMIC_RELAXED); }

}  // namespace

RegexPtr regex_compile(const std::string& pattern, const HostTables& ht, std::string& err) {
    RegexPtr prog(new RegexProg());
    prog->ht = &ht;
    Parser ps(pattern, *prog);
    auto ast = ps.alternation(false);
    if (ast && !ps.eof()) { ps.fail("unbalanced )", ps.i); ast = nullptr; }
    if (ast && Parser::nullable(*ast))
        ps.fail("the pattern can match the empty string (what find_iter does behind an empty match differs between the reference's regex back ends)", 0), ast = nullptr;
    if (!ast) { err = ps.err.empty() ? "malformed split pattern" : ps.err; return nullptr; }
    for (ClassSet& cs : prog->sets)                                       // ASCII membership of every class, once
        for (uint32_t c = 0; c < 128; c++) {
            const uint32_t cls = ht.ucls_stage2[((uint32_t)ht.ucls_stage1[0] << ht.ucls_shift) | c];
            bool in = ((cs.codes >> cls) & 1u) != 0;
            if (!in && cs.gcs) in = ((cs.gcs >> host_cp_category(ht, c)) & 1u) != 0;
            for (const auto& r : cs.ranges) in = in || (c >= r.first && c <= r.second);
            if (in != cs.neg) cs.ascii[c >> 6] |= 1ull << (c & 63);
        }
    Emitter em{*prog};
    em.gen(*ast);
    em.emit(OP_MATCH);
    if (em.too_big) { err = "the split pattern expands to more than 200 000 matcher instructions"; return nullptr; }
    return prog;
}

bool regex_device_image(const RegexProg& prog, std::vector<uint32_t>& w) {
    w.assign(RX_HDR_WORDS, 0u);
    if (prog.code.size() > 0xFFFFu) return false;                         // (a program counter is 16 bits of a stack word on the device)
    w[0] = (uint32_t)prog.code.size();
    for (const Inst& in0 : prog.code) {
        Inst in = in0;
        // (a jump to a jump, or to the end of the run, is that instruction itself: one matcher step less per match)
        for (int hops = 0; in.op == OP_JMP && hops < 8; hops++) {
            const Inst& to = prog.code[in.x];
            if (to.op == OP_JMP || to.op == OP_MATCH) in = to; else break;
        }
        w.push_back((uint32_t)in.op); w.push_back(in.x); w.push_back(in.y); w.push_back(in.f);
    }
    // class sets that a run instruction repeats: a bitmap slot each (the first RX_MAX_RUNSETS of them)
    std::vector<uint32_t> slot_of(prog.sets.size(), 0xFFFFFFFFu), run_sets;
    for (size_t pc = 0; pc + 1 < prog.code.size(); pc++)
        if (prog.code[pc].op == OP_REP1 && prog.code[pc + 1].op == OP_CLASS) {
            const uint32_t sx = prog.code[pc + 1].x;
            if (slot_of[sx] == 0xFFFFFFFFu && run_sets.size() < RX_MAX_RUNSETS) { slot_of[sx] = (uint32_t)run_sets.size(); run_sets.push_back(sx); }
        }
    std::vector<uint32_t> ranges;
    w[1] = (uint32_t)w.size(); w[2] = (uint32_t)prog.sets.size();
    for (size_t si = 0; si < prog.sets.size(); si++) {
        const ClassSet& cs = prog.sets[si];
        w.push_back(cs.codes); w.push_back(cs.gcs); w.push_back(cs.neg ? 1u : 0u);
        w.push_back((uint32_t)cs.ascii[0]); w.push_back((uint32_t)(cs.ascii[0] >> 32));
        w.push_back((uint32_t)cs.ascii[1]); w.push_back((uint32_t)(cs.ascii[1] >> 32));
        w.push_back((uint32_t)(ranges.size() / 2)); w.push_back((uint32_t)cs.ranges.size());
        w.push_back(slot_of[si]);
        for (const auto& r : cs.ranges) { ranges.push_back(r.first); ranges.push_back(r.second); }
        if (cs.gcs) w[7] = 1u;
    }
    w[3] = (uint32_t)w.size(); w[4] = (uint32_t)prog.firsts.size();
    for (const FirstSet& fs : prog.firsts) {
        w.push_back((uint32_t)fs.ascii[0]); w.push_back((uint32_t)(fs.ascii[0] >> 32));
        w.push_back((uint32_t)fs.ascii[1]); w.push_back((uint32_t)(fs.ascii[1] >> 32));
        w.push_back(fs.other ? 1u : 0u);
    }
    w[5] = (uint32_t)w.size(); w[6] = (uint32_t)(ranges.size() / 2);
    w.insert(w.end(), ranges.begin(), ranges.end());
    // (w[8]: the first-byte table, filled in below once the alternatives are known)
    w[9] = (uint32_t)run_sets.size(); w[10] = (uint32_t)w.size();
    w.insert(w.end(), run_sets.begin(), run_sets.end());
    while (w.size() % 4) w.push_back(0u);
    // ---- the alternatives of the top-level alternation, for the device matcher's lock-step evaluation ------------------
    // An alternative is SIMPLE if it is a straight line of one-character / run items up to the end of the pattern (jumps followed:
    // (a|b)c is the alternatives ac, bc -- the order of exploration is the same) in which giving characters back can never help:
    // every run that could give some back is possessive, or its characters can be taken by none of the items behind it up to and
    // including the first one that must take a character.  Greedy item by item is then exactly what the backtracking matcher finds.
    struct Item { uint32_t op, x, mn, mx; bool gives_back; };
    const HostTables& ht = *prog.ht;
    auto takes = [&](const Item& it, const Ch& c) {
        if (it.op == OP_CHAR) return c.cp == it.x;
        if (it.op == OP_CHAR_FOLD) return c.cp != CP_INVALID && fold_eq(it.x, c.cp);
        if (it.op == OP_ANY) return c.cp != '\n';
        return in_set(ht, prog.sets[it.x], c);
    };
    std::vector<std::pair<std::array<uint32_t, 4>, bool>> seen;       // (pairs already compared: a pattern repeats its few sets)
    auto disjoint = [&](const Item& a, const Item& b) {     // no character that both items take (every code point, and the byte that is none)
        const std::array<uint32_t, 4> key{a.op, a.x, b.op, b.x};
        for (const auto& kv : seen) if (kv.first == key) return kv.second;
        bool dj = !(takes(a, Ch{CP_INVALID, 1, C_P}) && takes(b, Ch{CP_INVALID, 1, C_P}));
        for (uint32_t cp = 0; cp < 0x110000u && dj; cp++) {
            const Ch c{cp, 1, host_cp_class(ht, cp)};
            if (takes(a, c) && takes(b, c)) dj = false;
        }
        seen.emplace_back(key, dj);
        return dj;
    };
    // A simple alternative may END in a tail that is tried at every length of its last run, longest first (the one place where the
    // backtracking matcher's way back is kept, as a loop): ONE one-character item whose characters the run may take too (\s*[\r\n]), a
    // look-ahead of one one-character item (\s+(?!\S)), or the assertion $ / \z.  In front of a look-ahead or assertion tail no other item
    // may give anything back (what it gave back could change where the tail is tested).
    enum : uint32_t { TAIL_NONE = 0, TAIL_ITEM, TAIL_LOOK, TAIL_NLOOK, TAIL_EOL, TAIL_EOT };
    struct Alt { uint32_t f, start; bool simple; std::vector<Item> items; uint32_t tail; Item tail_item; };
    std::vector<Alt> alts;
    {
        uint32_t pc = 0;
        while (prog.code[pc].op == OP_SPLIT && prog.code[pc].f != 0xFFFFFFFFu) { alts.push_back(Alt{prog.code[pc].f, prog.code[pc].x, false, {}, TAIL_NONE, Item{}}); pc = prog.code[pc].y; }
        alts.push_back(Alt{0xFFFFFFFFu, pc, false, {}, TAIL_NONE, Item{}});
    }
    auto one_char = [](const Inst& in) { return in.op == OP_CHAR || in.op == OP_CHAR_FOLD || in.op == OP_CLASS || in.op == OP_ANY; };
    for (Alt& al : alts) {
        uint32_t pc = al.start;
        bool ok = true;
        for (int guard = 0; guard < 64 && ok; guard++) {
            const Inst& in = prog.code[pc];
            if (in.op == OP_MATCH) break;
            if (in.op == OP_JMP) { pc = in.x; continue; }
            if (al.tail != TAIL_NONE) { ok = false; break; }            // (a tail is the last thing of its alternative)
            if (in.op == OP_REP1) {
                const Inst& a1 = prog.code[pc + 1];
                al.items.push_back(Item{(uint32_t)a1.op, a1.x, in.x, in.y, in.f != 1u && in.y > in.x});
                pc += 2;
            } else if (one_char(in)) {
                al.items.push_back(Item{(uint32_t)in.op, in.x, 1u, 1u, false});
                pc++;
            } else if ((in.op == OP_LOOK || in.op == OP_NLOOK) && one_char(prog.code[pc + 1]) && prog.code[pc + 2].op == OP_MATCH) {
                al.tail = in.op == OP_LOOK ? TAIL_LOOK : TAIL_NLOOK;
                al.tail_item = Item{(uint32_t)prog.code[pc + 1].op, prog.code[pc + 1].x, 1u, 1u, false};
                pc = in.x;
            } else if (in.op == OP_ASSERT && (in.x == AS_EOL || in.x == AS_EOT)) {
                al.tail = in.x == AS_EOL ? TAIL_EOL : TAIL_EOT;
                pc++;
            } else ok = false;
            if (al.items.size() > 8) ok = false;
        }
        if (ok && prog.code[pc].op != OP_MATCH) ok = false;
        if (ok && pc + 1 != prog.code.size()) ok = false;              // (the end of the PATTERN, not of a look-ahead's sub-program)
        if (ok && al.items.empty()) ok = false;
        // a last one-character item behind a run that gives back and shares characters with it: the tail of that run
        if (ok && al.tail == TAIL_NONE && al.items.size() >= 2) {
            const Item& last = al.items.back();
            const Item& run = al.items[al.items.size() - 2];
            if (last.mn == 1 && last.mx == 1 && run.gives_back && !disjoint(run, last)) {
                al.tail = TAIL_ITEM; al.tail_item = last;
                al.items.pop_back();
            }
        }
        const size_t n_plain = al.tail == TAIL_NONE ? al.items.size() : al.items.size() - 1;      // items in front of the tail's run
        for (size_t j = 0; j < n_plain && ok; j++) {
            if (!al.items[j].gives_back) continue;
            if (al.tail >= TAIL_LOOK) { ok = false; break; }
            for (size_t k = j + 1; k < al.items.size() && ok; k++) {
                if (!disjoint(al.items[j], al.items[k])) ok = false;
                if (al.items[k].mn >= 1) break;
                if (k + 1 == al.items.size() && al.tail == TAIL_ITEM && !disjoint(al.items[j], al.tail_item)) ok = false;
            }
        }
        al.simple = ok;
    }
    // for each ASCII byte and (entry 128) any other: which alternatives an attempt that begins with it can start -- bit i: alternative i's
    // first-character filter lets it in (all bits if the pattern has more than 32 alternatives: the filters are then tested one by one)
    w[8] = (uint32_t)w.size();
    for (uint32_t b = 0; b <= 128; b++) {
        uint32_t mask = 0;
        for (size_t i = 0; i < alts.size() && i < 32; i++) {
            bool can = true;
            if (alts[i].f != 0xFFFFFFFFu) { const FirstSet& fs = prog.firsts[alts[i].f]; can = b < 128 ? ((fs.ascii[b >> 6] >> (b & 63)) & 1ull) != 0 : fs.other; }
            if (can) mask |= 1u << i;
        }
        w.push_back(alts.size() > 32 ? 0xFFFFFFFFu : mask);
    }
    while (w.size() % 4) w.push_back(0u);
    // ---- the two-byte dispatch: for an attempt that begins with two ASCII bytes, the alternatives that can start with THAT pair -- a blank
    // in front of a digit starts ` ?\p{N}+`, not the letter run, the punctuation run and the two blank runs its first byte alone lets in.
    // A simple alternative is walked over the two bytes exactly as the device walks it (greedy, item by item); whatever two bytes do not
    // decide -- an alternative that runs the matcher program, a tail that walks back into its run, items beyond the second byte -- counts
    // as "can".  Bytes that behave alike share a class (rows / columns of the 128 x 128 table that are equal): cls0[128], cls1[128] as bytes,
    // the number of column classes, then the table.  w[12] = 0: none (more than 32 alternatives, or too many classes).
    if (alts.size() <= 32) {
        auto first_ok = [&](const Alt& al, uint32_t b) {
            if (al.f == 0xFFFFFFFFu) return true;
            const FirstSet& fs = prog.firsts[al.f];
            return ((fs.ascii[b >> 6] >> (b & 63)) & 1ull) != 0;
        };
        auto can2 = [&](const Alt& al, uint32_t b0, uint32_t b1) {
            if (!first_ok(al, b0)) return false;
            if (!al.simple) return true;
            const uint32_t ch[2] = {b0, b1};
            int ci = 0;
            for (size_t j = 0; j < al.items.size(); j++) {
                const Item& it = al.items[j];
                if (j + 1 == al.items.size() && al.tail != TAIL_NONE) return true;
                uint32_t k = 0;
                while (k < it.mx && ci < 2 && takes(it, Ch{ch[ci], 1, host_cp_class(ht, ch[ci])})) { k++; ci++; }
                if (ci == 2) return true;
                if (k < it.mn) return false;
            }
            return true;
        };
        std::vector<uint32_t> full(128 * 128, 0u);
        for (uint32_t b0 = 0; b0 < 128; b0++)
            for (uint32_t b1 = 0; b1 < 128; b1++) {
                uint32_t mask = 0;
                for (size_t i = 0; i < alts.size(); i++) if (can2(alts[i], b0, b1)) mask |= 1u << i;
                full[b0 * 128 + b1] = mask;
            }
        std::vector<uint32_t> cls0(128), cls1(128), rep0, rep1;
        for (uint32_t b = 0; b < 128; b++) {
            size_t k = 0;
            for (; k < rep0.size(); k++) if (std::equal(full.begin() + b * 128, full.begin() + b * 128 + 128, full.begin() + rep0[k] * 128)) break;
            if (k == rep0.size()) rep0.push_back(b);
            cls0[b] = (uint32_t)k;
        }
        for (uint32_t b = 0; b < 128; b++) {
            size_t k = 0;
            for (; k < rep1.size(); k++) {
                bool eq = true;
                for (uint32_t r = 0; r < 128 && eq; r++) eq = full[r * 128 + b] == full[r * 128 + rep1[k]];
                if (eq) break;
            }
            if (k == rep1.size()) rep1.push_back(b);
            cls1[b] = (uint32_t)k;
        }
        if (rep0.size() * rep1.size() <= 512) {
            w[12] = (uint32_t)w.size();
            for (uint32_t q = 0; q < 32; q++) w.push_back(cls0[4 * q] | cls0[4 * q + 1] << 8 | cls0[4 * q + 2] << 16 | cls0[4 * q + 3] << 24);
            for (uint32_t q = 0; q < 32; q++) w.push_back(cls1[4 * q] | cls1[4 * q + 1] << 8 | cls1[4 * q + 2] << 16 | cls1[4 * q + 3] << 24);
            w.push_back((uint32_t)rep1.size());
            for (uint32_t r : rep0) for (uint32_t c : rep1) w.push_back(full[r * 128 + c]);
            while (w.size() % 4) w.push_back(0u);
        }
    }
    w[11] = (uint32_t)w.size();
    w.push_back((uint32_t)alts.size()); w.push_back(0u); w.push_back(0u); w.push_back(0u);
    const size_t ent = w.size();
    w.resize(ent + 4 * alts.size(), 0u);
    for (size_t i = 0; i < alts.size(); i++) {
        const Alt& al = alts[i];
        const uint32_t off = (uint32_t)w.size();
        if (al.simple) {
            for (const Item& it : al.items) { w.push_back(it.op); w.push_back(it.x); w.push_back(it.mn); w.push_back(it.mx); }
            // (the tail's item behind them; its last word: the fewest characters the run in front of it may be cut back to)
            const Item& run = al.items.back();
            w.push_back(al.tail_item.op); w.push_back(al.tail_item.x); w.push_back(0u); w.push_back(run.gives_back ? run.mn : 0xFFFFFFFFu);
        }
        // (bits 16..23 of the flags: how many alternatives from this one on have its SHAPE -- simple, as many items, single characters and runs
        //  at the same places, the same kind of tail: the device evaluates them in ONE sweep, every lane with the items of its own alternative)
        uint32_t glen = 1;
        auto same_shape = [](const Alt& x, const Alt& y) {
            if (!x.simple || !y.simple || x.items.size() != y.items.size() || x.tail != y.tail) return false;
            for (size_t j = 0; j < x.items.size(); j++) if ((x.items[j].mx == 1u) != (y.items[j].mx == 1u)) return false;
            return true;
        };
        if (alts.size() <= 32) while (i + glen < alts.size() && glen < 16 && same_shape(al, alts[i + glen])) glen++;
        w[ent + 4 * i] = al.f; w[ent + 4 * i + 1] = al.simple ? (1u | (al.tail << 8) | (glen << 16)) : 0u; w[ent + 4 * i + 2] = al.start;
        w[ent + 4 * i + 3] = (al.simple ? (uint32_t)al.items.size() : 0u) | (off << 16);
    }
    return w.size() <= RX_IMAGE_MAX_WORDS;
}

bool regex_split_spans(const RegexProg& prog, const uint8_t* text, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& out) {
    return split_text(prog, text, n, [&](size_t a, size_t e) { out.emplace_back((uint32_t)a, (uint32_t)e); });
}

bool regex_split_bits(const RegexProg& prog, const uint8_t* text, size_t n, uint64_t base, uint32_t* start_bits, uint32_t* gap_bits) {
    size_t covered = 0;                                                  // end of the previous match
    // start bits are collected per bitmap word and OR-ed in when the word is left (positions only grow): one atomic
    // per 32 bytes of text instead of one per chunk
    uint64_t cur_w = ~0ull;
    uint32_t cur_bits = 0;
    auto flush = [&] { if (cur_bits) __atomic_fetch_or(&start_bits[cur_w], cur_bits, __ATOMIC_RELAXED); cur_bits = 0; };
    auto start = [&](uint64_t pos) {
        if ((pos >> 5) != cur_w) { flush(); cur_w = pos >> 5; }
        cur_bits |= 1u << (pos & 31);
    };
    auto gap = [&](size_t a, size_t e) {                                 // bytes no match covers: dropped (tokenizer.rs:729-808 only walks the matches)
        start(base + a);
        for (size_t q = a; q < e; q++) or_bit(gap_bits, base + q);
    };
    const bool ok = split_text(prog, text, n, [&](size_t a, size_t e) {
        if (a > covered) gap(covered, a);
        start(base + a);
        covered = e;
    });
    if (ok && covered < n) gap(covered, n);
    flush();
    return ok;
}

}  // namespace spl
