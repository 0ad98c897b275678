"""GPU tests (-m gpu) of the chunk memo (csrc/spl_k_memo.h): what the reference's LRU of encoded chunks is to its CPU path
(/root/reference/src/core/tokenizer.rs:707-722) -- result-transparent there and here.  The memo is filled BETWEEN launches from the
chunks the tiles logged, so every test encodes the same data several times and checks every pass against the oracle: cold (nothing
held), while it fills, warm; with a table so small that chunks keep losing their slots; with chunks of every token count up to and
beyond what an entry holds; off."""
import ctypes
import os
import random

import numpy as np
import pytest

from conftest import VOCABS

pytestmark = pytest.mark.gpu


def _csr(orc, texts, special=False):
    bs = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    return orc.encode_packed(np.frombuffer(b"".join(bs), dtype=np.uint8), off, special, threads=os.cpu_count() or 8)


def _stats(t):
    from splintr_amd import _ffi
    L = _ffi.lib()
    L.spl_memo_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    o = (ctypes.c_uint64 * 4)()
    assert L.spl_memo_stats(t.handle, o) == 0
    return list(o)


def _opt(t, k, v):
    from splintr_amd import _ffi
    assert _ffi.lib().spl_set_option(t.handle, k.encode(), int(v)) == 0, _ffi.last_error()


def _passes(t, orc, texts, n, special=False):
    want = _csr(orc, texts, special)
    for k in range(n):
        ids, off = t.encode_batch_csr(texts, with_special=special)
        assert np.array_equal(off, want[1]) and np.array_equal(ids, want[0]), f"pass {k}"


@pytest.mark.parametrize("name", VOCABS)
def test_every_pass_equals_the_oracle_while_the_memo_fills(coracle, name):
    from splintr_amd import Tokenizer, corpus
    t = Tokenizer.from_pretrained(name)
    texts = corpus.c2_wide(300, seed=77) + corpus.c3(40, seed=78) + corpus.c4(200, seed=79)
    _passes(t, coracle(name), texts, 6)
    st = _stats(t)
    assert st[3] > 0 and st[0] >= 1 and st[1] > 0, st                  # the table exists, a fill ran, chunks were put in
    # other data through the warm memo, then the first again
    other = corpus.c2_wide(200, seed=5) + corpus.c2(100, seed=6)
    _passes(t, coracle(name), other, 3)
    _passes(t, coracle(name), texts, 2)


def test_a_table_of_sixteen_entries_keeps_losing_its_chunks_and_stays_exact(coracle):
    from splintr_amd import Tokenizer, corpus
    t = Tokenizer.from_pretrained("cl100k_base")
    _opt(t, "memo_bits", 4)
    _opt(t, "memo_long_bits", 4)
    _opt(t, "memo_log_cap", 8)
    texts = corpus.c2_wide(400, seed=11)
    _passes(t, coracle("cl100k_base"), texts, 8)
    assert _stats(t)[3] == 16 + 16 and _stats(t)[0] >= 2


def test_chunks_of_every_token_count(coracle):
    """Chunks the vocabulary lacks with 2 .. 20+ tokens and 2 .. 64 bytes: up to six tokens live in an entry, seven to fourteen in the slot's
    second line, more are remembered as beyond it (and merged every time); chunks of 33 .. 64 bytes live in the second table."""
    from splintr_amd import Tokenizer
    rng = random.Random(3)
    orc = coracle("cl100k_base")
    words = []
    for n in range(2, 65):
        for _ in range(24):
            w = "".join(rng.choice("qxzjkvwQXZJKVW") for _ in range(n))
            words.append(w)
            words.append(" " + w[: max(1, n - 1)])
    texts = [" ".join(rng.sample(words, 60)) for _ in range(300)]
    counts = {len(orc.encode_batch([w])[0]) for w in words}
    assert min(counts) <= 2 and max(counts) > 14, counts               # the data does hold chunks on both sides of every limit
    t = Tokenizer.from_pretrained("cl100k_base")
    _passes(t, orc, texts, 6)
    st = _stats(t)
    assert st[1] > 0 and st[2] > 0, st                                 # chunks put in, and chunks found to be beyond an entry
    _passes(t, orc, list(reversed(texts)), 2)


def test_special_tokens_and_a_custom_pattern_through_the_warm_memo(coracle):
    from splintr_amd import Tokenizer, corpus
    t = Tokenizer.from_pretrained("cl100k_base")
    base = corpus.c2_wide(150, seed=21)
    texts = [x[: len(x) // 2] + "<|endoftext|>" + x[len(x) // 2:] for x in base]
    for sp in (False, True, False, True, True):
        _passes(t, coracle("cl100k_base"), texts, 1, special=sp)
    assert _stats(t)[1] > 0


def test_long_identifiers_with_and_without_the_second_table(coracle):
    """Chunks of 33 .. 64 bytes (snake_case / camelCase identifiers, URLs): held by the second table; with it off ("memo_long_bits" 0) and
    with sixteen entries of it, the same ids."""
    from splintr_amd import Tokenizer
    rng = random.Random(9)
    parts = ["user", "account", "manager", "factory", "Handler", "Config", "Service", "request", "response", "buffer", "Stream", "http", "www", "index", "Value"]
    idents = ["_".join(rng.sample(parts, rng.randrange(4, 9))) for _ in range(150)] + ["".join(rng.sample(parts, rng.randrange(5, 10))) for _ in range(150)]
    idents += ["https://" + ".".join(rng.sample(parts, 3)) + "/" + "/".join(rng.sample(parts, 3)) for _ in range(60)]
    assert any(33 <= len(x) <= 64 for x in idents)
    texts = [" ".join(rng.choice(idents) for _ in range(rng.randrange(3, 40))) + "\n" for _ in range(400)]
    orc = coracle("cl100k_base")
    for bits in (None, 0, 4):
        t = Tokenizer.from_pretrained("cl100k_base")
        if bits is not None:
            _opt(t, "memo_long_bits", bits)
        _passes(t, orc, texts, 6)
        _passes(t, orc, list(reversed(texts)), 2)
        if bits != 0:
            assert _stats(t)[1] > 0


def test_memo_off_and_on_in_one_handle(coracle):
    from splintr_amd import Tokenizer, corpus
    t = Tokenizer.from_pretrained("o200k_base")
    texts = corpus.c3(60, seed=31) + corpus.c2_wide(200, seed=32)
    _opt(t, "memo", 0)
    _passes(t, coracle("o200k_base"), texts, 2)
    assert _stats(t)[0] == 0
    _opt(t, "memo", 1)
    _passes(t, coracle("o200k_base"), texts, 4)
    assert _stats(t)[1] > 0
    _opt(t, "memo", 0)
    _passes(t, coracle("o200k_base"), texts, 1)


def test_the_latency_path_and_device_batches_share_the_memo(coracle):
    import torch
    from splintr_amd import Tokenizer, corpus
    from splintr_amd.device import DeviceBatch, encode_device, result_csr
    t = Tokenizer.from_pretrained("cl100k_base")
    orc = coracle("cl100k_base")
    texts = corpus.c2_wide(1000, seed=41)
    b = DeviceBatch(texts, torch.device("cuda", 0))
    want = _csr(orc, texts)
    for _ in range(4):
        encode_device(t, b)
        torch.cuda.synchronize()
        ids, off = result_csr(b)
        assert np.array_equal(ids, want[0]) and np.array_equal(off, want[1])
    for x in texts[:200]:                                              # single texts (the latency path) through the memo the batches filled
        assert t.encode(x) == orc.encode_batch([x])[0]


def test_cache_len_and_clear_cache_are_the_memo_s(coracle):
    """Tokenizer::cache_len / clear_cache (/root/reference/src/core/tokenizer.rs:995-1005; its tests :1111-1129)."""
    from splintr_amd import Tokenizer, corpus
    t = Tokenizer.from_pretrained("cl100k_base")
    assert t.cache_len == 0
    texts = corpus.c2_wide(400, seed=51)
    _passes(t, coracle("cl100k_base"), texts, 4)
    assert t.cache_len > 0
    t.clear_cache()
    assert t.cache_len == 0
    _passes(t, coracle("cl100k_base"), texts, 4)
    assert t.cache_len > 0
