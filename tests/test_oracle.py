"""The oracle against everything that pins it (CPU only).

1. the 18 exact-id vectors the reference's own tests/docs hold (tests/golden/reference_vectors.json)
2. the toy-vocabulary known answers of the reference's bpe.rs unit tests (src/core/bpe.rs:203-250)
3. the three split engines against each other: libpcre2-8 (UTF|UCP; the reference's optional
   backend), Python `regex`, and the C restatement -- on adversarial fuzz strings
4. structural pins: tiling, encode_batch[i] == encode(text_i), empty input, vocab_size values,
   the ValueError text, committed SHA-256 fixtures of whole synthetic corpora
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, VOCABS
from fuzzgen import fuzz_corpus
from oracle import pyoracle as O


@pytest.mark.parametrize("name", VOCABS)
@pytest.mark.parametrize("engine", ["pcre2", "regex"])
def test_python_oracle_reference_vectors(golden, name, engine):
    if engine == "pcre2" and not O.pcre2_available():
        pytest.skip("libpcre2-8 not loadable")
    t = O.Oracle.from_pretrained(name, engine=engine)
    for text, ids in golden[name]:
        assert t.encode(text) == ids, (name, text)


@pytest.mark.parametrize("name", VOCABS)
def test_c_oracle_reference_vectors(golden, coracle, name):
    c = coracle(name)
    for text, ids in golden[name]:
        assert c.encode(text) == ids, (name, text)


def test_mistral_v3_special_ids_held_by_the_reference(golden_all, coracle):
    # tests/mistral_v3.rs:27-150
    py = O.Oracle.from_pretrained("mistral_v3", engine="pcre2" if O.pcre2_available() else "regex")
    c = coracle("mistral_v3")
    for text, ids in golden_all["_mistral_v3_with_special"]:
        assert py.encode_with_special(text) == ids and c.encode_with_special(text) == ids, text
    assert py.vocab_size == 131126                       # tests/mistral_v3.rs:75-79


@pytest.mark.parametrize("name", VOCABS)
def test_reference_test_strings(coracle, name):
    """The strings the reference's Python tests run through regexr, PCRE2 and regexr-without-JIT and
    require equal tokens for (python/tests/test_cl100k.py:436-570): the committed ids (PCRE2 oracle)
    are reproduced by the independent `regex` engine and by the C oracle, and round-trip."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_strings.json"), encoding="utf-8") as f:
        fx = json.load(f)
    py = O.Oracle.from_pretrained(name, engine="regex")
    c = coracle(name)
    for s, ids in zip(fx["plain"], fx["ids"][name]):
        assert py.encode(s) == ids and c.encode(s) == ids, (name, s)
        assert py.decode_bytes(ids).decode("utf-8") == s
    for s, ids in zip(fx["special"], fx["ids_with_special"][name]):
        assert py.encode_with_special(s) == ids and c.encode_with_special(s) == ids, (name, s)
    batch = c.encode_batch(fx["large_batch_base"] * 100, threads=8)          # the 700-text batch
    assert batch == [fx["ids"][name][fx["plain"].index(s)] for s in fx["large_batch_base"]] * 100


def test_bpe_toy_vocab_known_answers():
    # src/core/bpe.rs:203-250
    enc = {b"a": 0, b"b": 1, b"c": 2, b"ab": 3, b"bc": 4, b"abc": 5}
    assert O.byte_pair_encode(b"a", enc) == [0]
    assert O.byte_pair_encode(b"ab", enc) == [3]
    assert O.byte_pair_encode(b"abc", enc) == [5]
    assert O.byte_pair_encode(b"", enc) == []
    assert O.byte_pair_encode(b"ac", enc) == [0, 2]
    # whole-piece hit is checked before merging; unknown bytes are dropped (bpe.rs:73-75, 182-191)
    assert O.byte_pair_encode(b"z", enc) == []
    assert O.byte_pair_encode(b"az", enc) == [0]
    # leftmost minimum wins on ties: "abab" -> ab ab
    assert O.byte_pair_encode(b"abab", enc) == [3, 3]
    assert O.byte_pair_encode(b"bcbc", enc) == [4, 4]
    assert O.byte_pair_encode(b"abcabc", {**enc, b"abcabc": 9}) == [9]


def test_byte_level_table():
    # src/core/byte_level.rs:46-74 and its unit tests: space -> U+0120, bijection, identity ranges
    assert O.BYTE_TO_CHAR[0x20] == "Ġ"
    assert O.BYTE_TO_CHAR[ord("A")] == "A"
    assert O.BYTE_TO_CHAR[0] == "Ā"
    assert len(set(O.BYTE_TO_CHAR)) == 256
    assert O.byte_level_encode("你".encode()) == "ä½ł".encode()
    assert O.byte_level_decode_bytes(O.byte_level_encode(bytes(range(256)))) == bytes(range(256))


@pytest.mark.parametrize("pattern_name", ["cl100k", "o200k", "mistral_v3"])
def test_split_engines_agree_on_fuzz(coracle, pattern_name):
    if not O.pcre2_available():
        pytest.skip("libpcre2-8 not loadable")
    pat = {"cl100k": O.CL100K_BASE_PATTERN, "o200k": O.O200K_BASE_PATTERN, "mistral_v3": O.MISTRAL_V3_PATTERN}[pattern_name]
    c = coracle({"cl100k": "cl100k_base", "o200k": "o200k_base", "mistral_v3": "mistral_v3"}[pattern_name])
    slashy = ["a!\n/b", "x.\r\n//\n/y", "?/\n", " ;\n\n/ z", "/\n/\n1", "12 3", "a1b22c333", "it's DON'T"]
    for s in fuzz_corpus(20260928, 6000) + slashy:
        b = s.encode("utf-8")
        p = O.split_pcre2(pat, b)
        # tiling: matches cover the text with no gaps
        assert [a for a, _ in p] == [0] * bool(b) + [e for _, e in p[:-1]], s
        assert (p[-1][1] if p else 0) == len(b)
        if "\u180e" not in s:      # U+180E is \s in PCRE2 10.39 only (Unicode 14 vs `regex`): SURVEY 8c
            assert O.split_regex(pat, s) == p, s
        assert c.split_bytes(b) == [a for a, _ in p], s


@pytest.mark.parametrize("name", VOCABS)
def test_c_oracle_equals_python_oracle(coracle, name):
    py = O.Oracle.from_pretrained(name, engine="pcre2" if O.pcre2_available() else "regex")
    c = coracle(name)
    for s in fuzz_corpus(77, 1500):
        assert c.encode(s) == py.encode(s), (name, s)
        assert c.encode_with_special(s) == py.encode_with_special(s), (name, s)


@pytest.mark.parametrize("name", VOCABS)
def test_structural_pins(coracle, name):
    c = coracle(name)
    texts = ["Hello, world!", "The quick brown fox jumps over the lazy dog.", "", "   \n\t  ",
             "Multi-line\ntext\nwith\nnewlines", "Unicode: こんにちは 世界 🦀", "don't — “quoted” it’s"]
    batch = c.encode_batch(texts, threads=4)                 # tests/cl100k.rs:191-214
    assert batch == [c.encode(t) for t in texts]
    assert c.encode("") == [] and c.encode_batch([]) == []
    py = O.Oracle.from_pretrained(name, engine="regex")
    for t in texts:                                           # round trip (python/tests/test_cl100k.py:56-71)
        assert py.decode_bytes(c.encode(t)).decode("utf-8") == t


def test_vocab_size_and_error_text():
    sizes = {"cl100k_base": 100331, "o200k_base": 200073, "llama3": 128354, "deepseek_v3": 128954}
    for n, v in sizes.items():
        assert O.Oracle.from_pretrained(n, engine="regex").vocab_size == v   # tokenizer.rs:964-972
    with pytest.raises(ValueError, match="Unknown pretrained model: nope. See from_pretrained docstring"):
        O.Oracle.from_pretrained("nope")                                     # bindings.rs:161-164


def test_special_literals_cannot_overlap():
    # what makes "every occurrence is a match" equal to Aho-Corasick Standard/non-overlapping
    with open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8") as f:
        tab = json.load(f)
    for name, lits in tab.items():
        ks = [k.encode("utf-8") for k in lits]
        for a in ks:
            assert len(a) <= 32
            for b in ks:
                if a is not b:
                    assert a not in b
                for k in range(1, min(len(a), len(b))):
                    if not (a is b and k == len(a)):
                        assert a[-k:] != b[:k], (name, a, b)


def test_overlapping_special_sets_two_restatements_agree():
    """Literal sets whose occurrences can overlap (user maps): the Python and the C restatement of the
    reference's matcher (earliest end from the previous match's end, longest on a tie) agree."""
    import random
    from oracle import coracle as C
    extra = {"<|a|>": 100300, "<|a|>x": 100301, "a|><": 100302, "|>": 100303, "ab": 100304, "abc": 100305,
             "bcd": 100306, "aa": 100307, "aaa": 100308, "\n\n": 100310}
    c = C.COracle("cl100k_base")
    for lit, tid in extra.items():
        b = lit.encode()
        C.lib().orc_add_special(c._h, b, len(b), tid)
    py = O.Oracle.from_pretrained("cl100k_base", engine="regex")
    py.special_tokens.update(extra)
    rng = random.Random(5)
    atoms = list(extra) + ["a", "b", "c", "d", "x", "<", "|", ">", " ", "\n", "hello ", "<|endoftext|>", "<|a", "aaaa"]
    for _ in range(800):
        s = "".join(rng.choice(atoms) for _ in range(rng.randint(0, 30)))
        assert c.encode_with_special(s) == py.encode_with_special(s), s


def test_special_tokens(coracle):
    c = coracle("cl100k_base")                         # tests/cl100k.rs:104-186
    t = c.encode_with_special("Hello<|endoftext|>World")
    assert 100257 in t and c.encode("Hello") + [100257] + c.encode("World") == t
    assert 100257 not in c.encode("Hello<|endoftext|>World")
    assert c.encode_with_special("<|fim_suffix|>") == [100260]
    d = coracle("deepseek_v3")
    assert d.encode_with_special("<think>a</think>") == [128798] + d.encode("a") + [128799]


def test_corpus_fixtures(coracle):
    """SHA-256 of the id streams of the synthetic corpora (made by tools/make_fixtures.py with the
    PCRE2-backed Python oracle): the C oracle must reproduce them."""
    from splintr_amd import corpus
    with open(os.path.join(ROOT, "tests", "golden", "corpus_fixtures.json")) as f:
        fx = json.load(f)
    for key, ent in fx.items():
        texts = getattr(corpus, ent["generator"])(ent["n"], **ent.get("kwargs", {}))
        ids = coracle(ent["vocab"]).encode_batch(texts, threads=8)
        h = hashlib.sha256()
        for row in ids:
            h.update(np.asarray(row, dtype=np.uint32).tobytes())
            h.update(b"|")
        assert h.hexdigest() == ent["sha256"], key
        assert sum(map(len, ids)) == ent["n_tokens"], key
