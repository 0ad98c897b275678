"""GPU tests (-m gpu) of decode: spl_decode_batch behind Tokenizer.decode_bytes /
decode / decode_lossy / decode_batch against the oracle's restatement of Tokenizer::decode_bytes
(reference src/core/tokenizer.rs:877-958, src/core/byte_level.rs:125-146): vocabulary ids, special
ids, ids that are in neither map, ByteLevel keys that are not ByteLevel text, invalid UTF-8."""
import random

import numpy as np
import pytest

from conftest import VOCABS
from fuzzgen import fuzz_corpus

pytestmark = pytest.mark.gpu

_cache = {}


def pair(name):
    from splintr_amd import Tokenizer
    from oracle import pyoracle as O
    if name not in _cache:
        _cache[name] = (Tokenizer.from_pretrained(name), O.Oracle.from_pretrained(name, engine="regex"))
    return _cache[name]


@pytest.mark.parametrize("name", VOCABS)
def test_decode_bytes_random_id_streams(name):
    t, o = pair(name)
    rng = random.Random(4242)
    vs = t.vocab_size
    special_ids = sorted(o.special_tokens.values())
    holes = [i for i in range(max(o.encoder.values()) + 1, vs) if i not in set(special_ids)][:50]
    streams = []
    for _ in range(200):
        n = rng.randint(0, 80)
        s = []
        for _ in range(n):
            r = rng.random()
            if r < 0.70:
                s.append(rng.randrange(vs))                       # anything below vocab_size
            elif r < 0.80:
                s.append(rng.choice(special_ids))
            elif r < 0.88 and holes:
                s.append(rng.choice(holes))                       # an id of neither map: contributes nothing
            elif r < 0.94:
                s.append(rng.choice([vs, vs + 1, 2 ** 21, 2 ** 31 - 1, 2 ** 32 - 1]))   # beyond every table
            else:
                s.append(rng.randrange(min(300, vs)))             # deepseek ids 0..2, byte tokens, mistral control tokens
        streams.append(s)
    want = [o.decode_bytes(s) for s in streams]
    assert t._decode_batch_bytes(streams) == want
    for s, w in list(zip(streams, want))[:40]:
        assert t.decode_bytes(s) == w
    assert t._decode_batch_bytes([]) == [] and t._decode_batch_bytes([[], []]) == [b"", b""]


@pytest.mark.parametrize("name", VOCABS)
def test_decode_inverts_encode(name):
    t, o = pair(name)
    texts = ["Hello, world!", "   \n\t  ", "Unicode: こんにちは 世界 🦀", "don't — “quoted” it’s", ""] + fuzz_corpus(5, 300)
    enc = t.encode_batch(texts)
    assert t.decode_batch(enc) == texts
    enc_s = t.encode_batch_with_special(texts)
    assert t.decode_batch(enc_s) == texts
    for lit, tid in list(o.special_tokens.items())[:20]:
        assert tid in t.encode_with_special("a" + lit + "b")
        assert t.decode([tid]) == lit


def test_decode_errors_and_lossy():
    t, o = pair("cl100k_base")
    ids = t.encode("你好世界")                      # [57668, 53901, 3574, 244, 98220]: 世 is split over two tokens
    assert t.decode(ids) == "你好世界"
    half = ids[:3]                                  # ends inside a character
    with pytest.raises(ValueError):
        t.decode(half)                              # TokenizerError::Utf8Error -> ValueError (bindings.rs:300-304)
    with pytest.raises(ValueError):
        t.decode_batch([ids, half])
    assert t.decode_lossy(half) == o.decode_bytes(half).decode("utf-8", "replace")
    assert t.decode_batch_lossy([ids, half]) == ["你好世界", t.decode_lossy(half)]
    assert t.decode_bytes(half) == o.decode_bytes(half)


def test_decode_a_large_batch_twice_without_regrowing():
    t, o = pair("o200k_base")
    from splintr_amd import corpus
    texts = corpus.c3(400, seed=8)
    enc = t.encode_batch(texts)
    assert t.decode_batch(enc) == texts
    assert t.decode_batch(enc[:100]) == texts[:100]          # smaller call: scratch is reused
    assert sum(len(e) for e in enc) > 100000


def test_decode_pipeline_equals_one_piece_and_the_oracle():
    """Large batches are decoded in chunks of whole documents through two slots (ids of chunk k + 1 in and measured while chunk k's bytes
    are gathered and leave).  A small chunk size forces that path on a modest batch: empty documents (whole chunks of them), a document
    larger than a chunk, ids of neither map, special ids; the result must equal the one-piece path's and the oracle's, and the result
    buffer's growth path (bytes per token far above the first guess) must hold."""
    from splintr_amd import Tokenizer, corpus, _ffi
    L = _ffi.lib()
    t, o = pair("o200k_base")
    rng = random.Random(99)
    texts = corpus.c3(300, seed=21)
    enc = t.encode_batch(texts)
    streams = []
    for e in enc:
        streams.append(e)
        if rng.random() < 0.3:
            streams.extend([[]] * rng.randint(1, 40))                        # runs of empty documents: chunks without an id
    streams.append([x for e in enc[:60] for x in e])                         # one document of ~70 000 ids: larger than a chunk
    streams.append([t.vocab_size + 5, 2 ** 31 - 1] + enc[0] + sorted(o.special_tokens.values())[:3])
    want = [o.decode_bytes(s) for s in streams]
    tp = Tokenizer.from_pretrained("o200k_base")
    assert L.spl_set_option(tp.handle, b"decode_chunk_ids", 4096) == 0       # ~90 chunks
    got = tp._decode_batch_bytes(streams)
    assert got == want
    assert got == t._decode_batch_bytes(streams)                              # the one-piece path (chunks of 2^20 ids: below three chunks)
    assert tp._decode_batch_bytes(streams[:7]) == want[:7]                   # a small call on the same handle afterwards
    # far more bytes per token than the first guess of the result's size: ids of long tokens only
    longest = sorted(range(min(t.vocab_size, 200000)), key=lambda i: -len(o.decode_bytes([i])))[:50]
    big = [[rng.choice(longest) for _ in range(3000)] for _ in range(12)]
    assert tp._decode_batch_bytes(big) == [o.decode_bytes(s) for s in big]


def test_deepseek_ids_that_are_not_byte_level_text():
    t, o = pair("deepseek_v3")
    # ids 0..2 hold text with characters outside the ByteLevel alphabet: decode_bytes gives the key itself
    for i in range(3):
        assert t.decode_bytes([i]) == o.decode_bytes([i]) and len(t.decode_bytes([i])) > 3
    assert t.decode([0]) == "<｜begin▁of▁sentence｜>"


def test_special_ids_far_beyond_the_vocabulary_cost_nothing():
    """spl_add_special takes any id below 2**31: the decode table stays O(vocabulary) -- such ids live in a small
    sorted side table (a dense table up to the largest special id was tens of GB)."""
    from splintr_amd import Tokenizer, CL100K_BASE_PATTERN
    import os
    from conftest import ROOT
    with open(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), "rb") as f:
        blob = f.read()
    sp = {"<|far|>": 2 ** 31 - 1, "<|mid|>": 5_000_000, "<|near|>": 100300, "<|dup|>": 5_000_000}
    t = Tokenizer.from_bytes(blob, CL100K_BASE_PATTERN, sp)
    assert t.vocab_size == 2 ** 31
    ids = t.encode_with_special("a<|far|>b<|near|>c<|dup|>")
    assert ids == t.encode("a") + [2 ** 31 - 1] + t.encode("b") + [100300] + t.encode("c") + [5_000_000]
    assert t.decode_bytes(ids) == b"a<|far|>b<|near|>c<|dup|>"          # the later literal of a shared id wins, as a map insert
    assert t.decode_bytes([4_999_999, 5_000_001, 2 ** 31 - 2]) == b""
    assert t.decode_batch([[2 ** 31 - 1], [], [100300, 5_000_000]]) == ["<|far|>", "", "<|near|><|dup|>"]
