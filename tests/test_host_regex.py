"""SURVEY 8f rank 2, "arbitrary pattern": the product's host splitter (splintr_amd/csrc/spl_regex.cpp) against
PCRE2 with UTF | UCP -- the back end the reference itself declares equivalent to its default one
(src/core/tokenizer.rs:470-488, python/tests/test_cl100k.py:436-454) -- on adversarial fuzz, for the three
patterns the GPU scanner implements AND patterns it does not (GPT-2's, variants of the reference's own).  What
the splitter cannot express is refused at construction with the construct named.
Reference: Tokenizer::new compiles any pattern (src/core/tokenizer.rs:410-456)."""
import pytest

from fuzzgen import cased_corpus, fuzz_corpus, latin_corpus

GPT2_PATTERN = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
# a Tekken-like variant: single digits, no contractions, '/' joins the newline tail, {1,2} counted repeat
VARIANT_A = r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+|[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*|\p{N}{1,2}| ?[^\s\p{L}\p{N}]+[\r\n/]*|\s*[\r\n]+|\s+(?!\S)|\s+"
# cl100k with the contraction group spelt (?i) ... and numbers of up to four, lazy whitespace tail
VARIANT_B = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,4}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+?(?=\S)|\s+"
# a pattern that does NOT tile the text: only letter runs and digit runs are chunks, everything else is dropped
SPARSE = r"\p{L}+|[0-9]+"
# literals, escapes, \x{..}, a class with a range and an escaped bracket, '.' (no newline), alternation in a group
MIXED = r"(?:https?://|www\.)[^\s]+|[A-Za-z_][A-Za-z0-9_]*|\x{4f60}\x{597d}|[\[\]{}()]|.|\n"


# round 4: what upstream tokenizers actually ship.  tiktoken's cl100k_base pattern string (possessive quantifiers, a caseless
# bracket class, the `$` anchor), Qwen2's, a DeepSeek-style pattern with \p{P} \p{S} and an explicit CJK range, and a
# pattern of \d \w \b atomic-group pieces
TIKTOKEN_CL100K = r"'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|\s+(?!\S)|\s"
TIKTOKEN_O200K = "|".join([
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?",
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*(?i:'s|'t|'re|'ve|'m|'ll|'d)?",
    r"\p{N}{1,3}", r" ?[^\s\p{L}\p{N}]+[\r\n/]*", r"\s*[\r\n]+", r"\s+(?!\S)", r"\s+"])
QWEN2 = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
DEEPSEEK_LIKE = (r"\p{N}{1,3}|[\x{4e00}-\x{9fa5}\x{3040}-\x{309f}\x{30a0}-\x{30ff}]+|[!-/:-@\[-`{-~][A-Za-z]+|[^\r\n\p{L}\p{P}\p{S}]?[\p{L}\p{M}]+|"
                 r" ?[\p{P}\p{S}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
WORDS_DIGITS = r"\b\w+\b|\d++|(?>\s+)(?=\S)|\p{Zs}+|\p{Sc}\d*|\p{Pd}+|\A\W|\W\z|[^\w\s]"
# round 5: SCRIPT properties, as CJK-aware tokenizers use them (Kimi / GLM style: Han runs apart from the other letters, kana and
# hangul runs of their own), and their complements (\P{..}, \p{^..}, a script inside a negated class)
SCRIPTS = (r"\p{Han}+|[\p{Hiragana}\p{Katakana}\x{30fc}]+|\p{Hangul}+|[^\r\n\p{L}\p{N}]?[\p{Latin}\p{M}]+(?i:'s|'t|'re|'ve|'m|'ll|'d)?|[^\r\n\p{L}\p{N}]?\p{L}+|"
           r"\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
SCRIPTS_NEG = r"[^\P{Cyrillic}\x{0451}]+|\p{^Latin}{1,5}|\p{Latin}+[\P{Greek}&]?"


def _patterns():
    from splintr_amd import CL100K_BASE_PATTERN, O200K_BASE_PATTERN, MISTRAL_V3_PATTERN
    return {"cl100k": CL100K_BASE_PATTERN, "o200k": O200K_BASE_PATTERN, "mistral_v3": MISTRAL_V3_PATTERN, "gpt2": GPT2_PATTERN,
            "variant_a": VARIANT_A, "variant_b": VARIANT_B, "sparse": SPARSE, "mixed": MIXED, "tiktoken_cl100k": TIKTOKEN_CL100K,
            "tiktoken_o200k": TIKTOKEN_O200K, "qwen2": QWEN2, "deepseek_like": DEEPSEEK_LIKE, "words_digits": WORDS_DIGITS,
            "scripts": SCRIPTS, "scripts_neg": SCRIPTS_NEG}


@pytest.fixture(scope="module")
def sim():
    from hostsim import HostSim
    return HostSim("cl100k_base")


@pytest.mark.parametrize("key", ["cl100k", "o200k", "mistral_v3", "gpt2", "variant_a", "variant_b", "sparse", "mixed", "tiktoken_cl100k",
                                 "tiktoken_o200k", "qwen2", "deepseek_like", "words_digits", "scripts", "scripts_neg"])
def test_host_splitter_equals_pcre2(sim, key):
    from hostsim import HostRegex
    from oracle import pyoracle as O
    if not O.pcre2_available():
        pytest.skip("libpcre2-8 not present")
    pat = _patterns()[key]
    hr = HostRegex(pat, sim)
    pc = O.Pcre2Pattern(pat)
    texts = fuzz_corpus(4242, 2500, 40) + latin_corpus(7, 400, 80) + cased_corpus(9, 400, 60)
    texts += ["", " ", "\n", "a", "'", "'s", "x's'S'ſ'K'K", "http://a.b/c?d=e www.x.y z", "你好你好 你 好", "a\nb\r\nc", "{[()]}", "12345678901",
              "a  ", "a \n", "  \n", "x  \n\n", "it'S 'LL 'ſ 'Ve", "$12 €3 £ -- — ―", "٣٤٥ ⅷ ² 12", "foo_bar1 baz", "ひらがな カタカナ 漢字 한글", "a b c",
              "«quoted» “x” ‘y’", "±×÷ ^ ` ~", "́x ⃝", "end\n", "end  \n",
              "漢字かなカナー한글 mixed латиница ελληνικά it's 漢's", "ёЁжук ё", "々〆〇 㐀 𠀀 ｶﾅ ㍿"]
    bad = 0
    for t in texts:
        b = t.encode("utf-8")
        want = [(a, e) for a, e in pc.find_iter(b) if e > a]
        got = hr.split(b)
        if got != want:
            bad += 1
            if bad <= 3:
                print(repr(t), got[:12], want[:12])
    assert bad == 0


def test_split_bits_mark_chunks_and_gaps(sim):
    from hostsim import HostRegex
    import numpy as np
    hr = HostRegex(SPARSE, sim)
    data = "  ab, 12x!".encode()
    st, gp = hr.split_bits(data, base=37)
    bit = lambda bm, p: (int(bm[(37 + p) >> 5]) >> ((37 + p) & 31)) & 1
    # chunks: "ab" [2,4) "12" [6,8) "x" [8,9); gaps [0,2) [4,6) [9,10)
    assert [p for p in range(len(data)) if bit(st, p)] == [0, 2, 4, 6, 8, 9]
    assert [p for p in range(len(data)) if bit(gp, p)] == [0, 1, 4, 5, 9]
    assert int(np.sum(st[:1])) == 0                     # nothing before the base


@pytest.mark.parametrize("pattern, what", [
    (r"(?<=a)b", "look-behind"), (r"(a)\1", r"\1"), (r"\p{Alphabetic}+", "Alphabetic"), (r"\p{Emoji}", "Emoji"), (r"\p{scx=Han}", "scx=Han"), (r"[[:alpha:]]", "POSIX"), (r"\Gx", r"\G"),
    (r"a\Kb", r"\K"), (r"[\W]", r"\W inside"), (r"(?i:\p{Lu}+)", "under (?i)"), (r"(?i:[à-ÿ])", "non-ASCII"),
    (r"(a", "without )"), (r"a)", "unbalanced"), (r"[a", "without ]"), (r"a{5,2}", "n < m"), (r"(?i:é)", "non-ASCII literal"),
    (r"a*", "empty string"), (r"(a|b*)c?", "empty string"), (r"(a*)*", "empty string"), (r"x{2000}", "beyond 1000"), (r"^", "empty string"),
    (r"a|$", "empty string"), (r"\b+", "on an assertion"),
])
def test_unsupported_constructs_are_named(sim, pattern, what):
    from hostsim import HostRegex
    with pytest.raises(ValueError) as e:
        HostRegex(pattern, sim)
    assert what in str(e.value), str(e.value)


def test_random_patterns_equal_pcre2(sim):
    """round 4: 400 random patterns from tests/patgen.py (alternations of short item sequences, every quantifier flavour, groups, look-ahead and
    assertion tails; class sets that overlap) on texts that make them backtrack: the host splitter == PCRE2, pattern by pattern"""
    import random
    from hostsim import HostRegex
    from oracle import pyoracle as O
    from patgen import random_pattern, random_texts
    if not O.pcre2_available():
        pytest.skip("libpcre2-8 not present")
    rng = random.Random(20260930)
    done = 0
    for _ in range(400):
        pat = random_pattern(rng)
        try:
            hr = HostRegex(pat, sim)
        except ValueError:
            continue                      # (can match the empty string, or a construct in a place the compiler refuses)
        pc = O.Pcre2Pattern(pat)
        texts = random_texts(rng, 40)
        for t in texts + ["".join(texts)]:
            b = t.encode("utf-8")
            want = [(a, e) for a, e in pc.find_iter(b) if e > a]
            assert hr.split(b) == want, (pat, t)
        done += 1
    assert done >= 150
