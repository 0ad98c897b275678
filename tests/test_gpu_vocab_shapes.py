"""GPU tests (-m gpu): vocabularies the reference accepts and that round 5 refused (missing #4).
(a) a vocabulary that LACKS single bytes -- byte_pair_encode ranks pairs by their concatenated bytes and drops a node whose bytes are
    no token (/root/reference/src/core/bpe.rs:73-75, 99-111, 182-191); the toy vocabulary of its own unit tests (bpe.rs:203-250) has
    three bytes in all;
(b) a vocabulary in which thousands of short keys share one two-byte prefix -- FxHashMap (tokenizer.rs:302) has no limit; the single-slot
    tables of this build grow until every key has a slot.
Both against oracle/pyoracle.py (the literal restatement) on the same inputs."""
import base64
import random

import pytest

pytestmark = pytest.mark.gpu

PATTERN = r"[a-z]+|\s+|[^a-z\s]+"        # (a custom pattern: split on the device, merged by the tile kernel)


def _tiktoken(enc):
    return b"".join(base64.b64encode(k) + b" " + str(v).encode() + b"\n" for k, v in sorted(enc.items(), key=lambda kv: kv[1]))


def _check(enc, texts, pattern=PATTERN):
    from splintr_amd import Tokenizer
    from oracle.pyoracle import Oracle
    t = Tokenizer.from_bytes(_tiktoken(enc), pattern)
    orc = Oracle(enc, pattern, False)
    want = [orc.encode(x) for x in texts]
    for _ in range(3):                                  # (cold, while the chunk memo fills, warm)
        assert t.encode_batch(texts) == want
    for x, w in list(zip(texts, want))[:40]:
        assert t.encode(x) == w
    return t, want


def test_the_reference_s_toy_vocabulary():
    enc = {b"a": 0, b"b": 1, b"c": 2, b"ab": 3, b"bc": 4, b"abc": 5}           # bpe.rs:203-215
    rng = random.Random(1)
    texts = ["a", "ab", "abc", "", "ac", "abcabc", "cab", "xyz", "a b c", "abz abc", "zzzz", "bca" * 50]
    texts += ["".join(rng.choice("abc abcx\n") for _ in range(rng.randrange(1, 300))) for _ in range(400)]
    t, want = _check(enc, texts)
    # the unit tests' own answers (bpe.rs:217-250)
    assert want[0] == [0] and want[1] == [3] and want[2] == [5] and want[3] == [] and want[4] == [0, 2]
    assert want[7] == []                                  # bytes the vocabulary lacks: nothing (bpe.rs:182-191)
    assert t.vocab_size == 6


def test_pairs_merge_through_a_byte_the_vocabulary_lacks():
    """"xy" and "axy" are tokens, "x" and "y" are not: the pair is ranked by its bytes (bpe.rs:99-111), so they merge; a lone x or y is
    dropped.  Long runs take the pair-table loops (spans beyond 8 bytes), short ones the tabulated ranks."""
    enc = {bytes([c]): i for i, c in enumerate(b"abcdefgh \n")}
    nxt = len(enc)
    for k in (b"xy", b"axy", b"xya", b"ab", b"abc", b"abcd", b"xyxy", b"abxy", b"abcdefgh", b"abcdefghxy", b"xyabcdefgh", b"  ", b"y ", b"hx"):
        enc[k] = nxt
        nxt += 1
    rng = random.Random(2)
    words = ["xy", "x", "y", "axy", "xya", "yx", "abxy", "xyxyxy", "abcdefghxy", "xyabcdefghxy", "hxy", "y y", "abcdefghxyabcdefghxyabcd", "qq", "xqy"]
    texts = [" ".join(rng.choice(words) for _ in range(rng.randrange(1, 60))) for _ in range(500)]
    texts += ["".join(rng.choice("abcdefghxy") for _ in range(n)) for n in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 300, 700, 1500, 5000)]
    _check(enc, texts)


def test_four_thousand_short_keys_under_one_prefix():
    """4 000 keys of 3 and 4 bytes that all begin with "ab" (+ the 256 single bytes): one group of the tiny table's hash-and-displace
    build, far beyond what a 16-bit salt separates in a table of the default size -- the table grows (spl_tables.cpp, displace)."""
    enc = {bytes([b]): b for b in range(256)}
    rng = random.Random(3)
    keys = set()
    while len(keys) < 4000:
        keys.add(b"ab" + bytes(rng.randrange(33, 127) for _ in range(rng.choice((1, 2)))))
    for k in sorted(keys):
        enc[k] = len(enc)
    enc[b"ab"] = len(enc)
    sample = sorted(keys)
    texts = [" ".join((rng.choice(sample).decode("latin-1") if rng.random() < 0.7 else "ab" + "".join(rng.choice("abcxyz!?") for _ in range(rng.randrange(0, 6))))
                      for _ in range(rng.randrange(1, 80))) for _ in range(400)]
    t, _ = _check(enc, texts, pattern=r"\S+|\s+")
    assert t.vocab_size == len(enc)
