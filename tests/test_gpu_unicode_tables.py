"""-m gpu: the code-point class table is an argument of spl_create (uclass_tab) -- VERDICT r03 #8.  Two tables ship:
the default, probed from PCRE2 10.39 (Unicode 14.0: the engine the reference's own tests declare equivalent to its
default one, python/tests/test_cl100k.py:436-454), and one probed from the Python `regex` module (a newer Unicode).
The reference's default engine is regexr (Cargo.toml:41) with tables of an unknown version: nothing in the reference
pins which is right, so the choice is the caller's (`Tokenizer(..., unicode_tables="regex")`).  Here the HIP path
with either table is checked bit-exact against the Python oracle running THAT engine -- including the 14 186 code
points the two tables class differently (assigned or re-categorised after Unicode 14, U+180E)."""
import pytest

from test_hostsim import _post14_corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base", "deepseek_v3"])
def test_fuzz_parity_on_both_class_tables(name):
    pytest.importorskip("regex")
    from fuzzgen import fuzz_corpus
    from oracle import pyoracle as O
    from splintr_amd import Tokenizer
    texts = _post14_corpus(7, 500) + fuzz_corpus(31, 300, 30)
    texts = [t for t in texts if "᠎" not in t] + ["a᠎b", " ᠎ ", "x ᠎᠎y"]
    tok_re = Tokenizer.from_pretrained(name, unicode_tables="regex")
    orc_re = O.Oracle.from_pretrained(name, engine="regex")
    got = tok_re.encode_batch(texts)
    want_re = [orc_re.encode(t) for t in texts]
    for t, g, w in zip(texts, got, want_re):
        assert g == w, (t, g[:12], w[:12])
    if O.pcre2_available():
        tok_pc = Tokenizer.from_pretrained(name)
        orc_pc = O.Oracle.from_pretrained(name, engine="pcre2")
        got_pc = tok_pc.encode_batch(texts)
        want_pc = [orc_pc.encode(t) for t in texts]
        for t, g, w in zip(texts, got_pc, want_pc):
            assert g == w, (t, g[:12], w[:12])
        assert sum(a != b for a, b in zip(want_pc, want_re)) > 20      # the two tables do tokenize this corpus differently
