"""GPU tests (-m gpu) of the host entry point a user calls -- spl_encode_batch behind
Tokenizer.encode_batch (reference src/python/bindings.rs:337-339): the chunked pinned pipeline,
several pipelines at once (spl_set_devices, here all on GPU 0), sub-document cuts, the result
buffer's growth path, pinned and pageable input.  Every result bit-exact against the oracle."""
import ctypes
import random

import numpy as np
import pytest

from test_gpu_parity import oracle_csr

pytestmark = pytest.mark.gpu


def fresh(name, **opts):
    from splintr_amd import Tokenizer, _ffi
    t = Tokenizer.from_pretrained(name)
    for k, v in opts.items():
        assert _ffi.lib().spl_set_option(t.handle, k.encode(), int(v)) == 0, _ffi.last_error()
    return t


def check(t, name, texts, coracle, special=False):
    ids, off = t.encode_batch_csr(texts, with_special=special)
    o_ids, o_off = oracle_csr(coracle(name), texts, special)
    assert np.array_equal(off, o_off), name
    assert np.array_equal(ids, o_ids), name
    return ids, off


def mixed_docs(seed, n, lo=0, hi=3000):
    from splintr_amd import corpus
    rng = random.Random(seed)
    base = corpus.c3(64, seed=seed)
    out = []
    for i in range(n):
        d = base[i % len(base)]
        k = rng.randint(lo, hi)
        out.append(d[:k] if rng.random() < 0.8 else "")
    return out


@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base", "deepseek_v3"])
@pytest.mark.parametrize("chunk", [1 << 12, 1 << 16, 1 << 20])
def test_chunked_pipeline_equals_oracle(coracle, name, chunk):
    """Tiny chunks force every batch through the multi-chunk path (producer thread, three slots,
    per-chunk placement) including chunks of empty documents and documents larger than a chunk."""
    t = fresh(name, chunk_bytes=chunk)
    texts = mixed_docs(7 + chunk, 700) + ["", "", "x" * 70000, ""]
    check(t, name, texts, coracle)
    check(t, name, [""] * 50, coracle)
    check(t, name, [], coracle)
    check(t, name, ["only one"], coracle)
    check(t, name, [""] * 9 + ["a b c"] + [""] * 9 + ["x" * 9000] + [""] * 9, coracle)
    check(t, name, texts[:40], coracle, special=True)


def test_result_buffer_growth(coracle):
    """A far too small first guess of the token count: the result moves to a larger pinned buffer
    in both the single-chunk and the multi-chunk path (digits: one token per 1..3 bytes)."""
    texts = ["7 8 9 1 2 3 4 5 6 " * 400 for _ in range(200)]
    check(fresh("cl100k_base", result_estimate_div=64, direct_write=0), "cl100k_base", texts, coracle)
    check(fresh("cl100k_base", result_estimate_div=64, chunk_bytes=1 << 16), "cl100k_base", texts, coracle)
    # one-chunk batches by default have the last kernel write the ids straight into the pinned result
    check(fresh("cl100k_base"), "cl100k_base", texts, coracle)
    check(fresh("cl100k_base", direct_write=0), "cl100k_base", texts[:7], coracle)


@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_several_pipelines_one_result(coracle, ndev):
    """spl_set_devices with GPU 0 listed several times: independent contexts (tables, workspace,
    streams) shard the batch by bytes and place their parts into one pinned CSR."""
    from splintr_amd import corpus
    t = fresh("o200k_base", chunk_bytes=1 << 20).set_devices([0] * ndev)
    texts = corpus.c3(3000, seed=99)                    # ~12 MB: every lane gets >= 1 MiB
    check(t, "o200k_base", texts, coracle)
    check(t, "o200k_base", texts[:3], coracle)          # too small to shard: one lane
    check(t, "o200k_base", texts, coracle, special=True)


def test_sub_document_cuts(coracle):
    """Two huge documents over four pipelines: lanes are cut INSIDE a document, behind a newline that
    is followed by a letter or digit (a context-free match boundary); ids and offsets as the oracle's."""
    from splintr_amd import corpus, _ffi
    docs = corpus.c5(2, seed=5, doc_bytes=5 << 20)
    t = fresh("deepseek_v3").set_devices([0, 0, 0, 0])
    ids, off = check(t, "deepseek_v3", docs, coracle)
    assert len(off) == 3
    # the same with the cuts switched off (whole documents per lane): same result
    assert _ffi.lib().spl_set_option(t.handle, b"subdoc_split", 0) == 0
    ids2, off2 = t.encode_batch_csr(docs)
    assert np.array_equal(ids, ids2) and np.array_equal(off, off2)
    # a document without any cut point (no newline) stays whole
    t2 = fresh("cl100k_base").set_devices([0, 0])
    check(t2, "cl100k_base", ["word " * (600 << 10), "b " * (1 << 20)], coracle)


def test_pinned_and_pageable_input_and_result_lifetime(coracle):
    from splintr_amd import Tokenizer, corpus, _ffi
    L = _ffi.lib()
    texts = corpus.c2(300, seed=3)
    bs = [x.encode() for x in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    blob = b"".join(bs)
    t = Tokenizer.from_pretrained("cl100k_base")
    want_ids, want_off = oracle_csr(coracle("cl100k_base"), texts)
    ids, oo = t.encode_packed(blob, off)                               # pageable input
    assert np.array_equal(ids, want_ids) and np.array_equal(oo, want_off)
    p = L.spl_host_alloc(len(blob) + 64)
    assert p
    ctypes.memmove(p, blob, len(blob))
    ids, oo = t._encode_packed(p, off.ctypes.data, len(bs), 0)        # pinned input: DMA straight from it
    assert np.array_equal(ids, want_ids) and np.array_equal(oo, want_off)
    # a batch of several chunks from pageable memory: the chunks are copied into pinned staging by several threads, the kernels of
    # consecutive chunks run on two streams (the context and its twin); the same batch from pinned memory; both against the oracle
    big = corpus.c3(3200, seed=5)                                      # ~13 MB: three chunks of <= 5 MiB
    bb = [x.encode() for x in big]
    boff = np.zeros(len(bb) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bb], out=boff[1:])
    bblob = b"".join(bb)
    t3 = Tokenizer.from_pretrained("o200k_base")
    big_ids, big_off = oracle_csr(coracle("o200k_base"), big)
    for opts in ({}, {"copy_threads": 1}, {"twin_streams": 0}, {"chunk_ramp": 1}):
        for k, v in opts.items():
            assert L.spl_set_option(t3.handle, k.encode(), v) == 0
        ids, oo = t3.encode_packed(bblob, boff)
        assert np.array_equal(ids, big_ids) and np.array_equal(oo, big_off), opts
    pb = L.spl_host_alloc(len(bblob) + 64)
    ctypes.memmove(pb, bblob, len(bblob))
    ids, oo = t3._encode_packed(pb, boff.ctypes.data, len(bb), 0)
    assert np.array_equal(ids, big_ids) and np.array_equal(oo, big_off)
    L.spl_host_free(pb)
    del t3
    # two results alive at once, one of them outliving the handle
    r1, r2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.spl_encode_batch(t.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r1)) == 0
    assert L.spl_encode_batch(t.handle, blob, off.ctypes.data, len(bs), 0, ctypes.byref(r2)) == 0
    del t
    import gc
    gc.collect()
    for r in (r1, r2):
        n = L.spl_result_n_tokens(r)
        assert np.array_equal(np.ctypeslib.as_array(L.spl_result_tokens(r), shape=(n,)), want_ids)
        L.spl_result_free(r)
    L.spl_host_free(p)


def test_python_surface_types_and_errors():
    from splintr_amd import Tokenizer
    t = Tokenizer.from_pretrained("cl100k_base")
    assert t.encode_batch(["Hello, world!", "", "你好世界"]) == [[9906, 11, 1917, 0], [], [57668, 53901, 3574, 244, 98220]]
    assert t.encode_batch(("Hello world",)) == [[9906, 1917]]
    assert t.encode_batch([]) == []
    out = t.encode_batch(["a"] * 3)
    assert all(type(x) is int for l in out for x in l)
    with pytest.raises(TypeError):
        t.encode_batch("a bare string")
    with pytest.raises(TypeError):
        t.encode_batch([b"bytes"])
    with pytest.raises(TypeError):
        t.encode(5)
    with pytest.raises(UnicodeEncodeError):
        t.encode_batch(["ok", "\ud800"])
    # latin-1, UCS-2 and UCS-4 strings take three different packing paths
    for s in ["café naïve", "你好，世界", "Hello \U0001F30D World!", "é你\U0001F30D" * 50]:
        assert t.encode_batch([s, s])[1] == t.encode(s)
        assert t.decode(t.encode(s)) == s


def test_option_and_device_errors():
    from splintr_amd import Tokenizer, _ffi
    L = _ffi.lib()
    t = Tokenizer.from_pretrained("cl100k_base")
    assert L.spl_set_option(t.handle, b"no_such_option", 1) != 0 and "unknown option" in _ffi.last_error()
    assert L.spl_set_option(t.handle, b"chunk_bytes", 0) != 0
    with pytest.raises(ValueError):
        t.set_devices([L.spl_device_count()])            # one past the last device
    with pytest.raises(ValueError):
        t.set_devices([])
    assert L.spl_n_devices(t.handle) == 1
    t.set_devices([0, 0])
    assert L.spl_n_devices(t.handle) == 2 and t.encode("Hello, world!") == [9906, 11, 1917, 0]


def test_round_trip_property_on_arbitrary_unicode():
    """decode(encode(t)) == t, and a document's ids do not depend on its neighbours, for batches drawn by
    hypothesis from all of Unicode (no surrogates) -- the structural pins the reference's tests apply to a
    handful of strings (python/tests/test_cl100k.py:56-71, tests/cl100k.rs:191-214), at scale and for every
    vocabulary.  Derandomised: the same examples every run."""
    from hypothesis import given, settings, strategies as st
    from conftest import VOCABS
    from splintr_amd import Tokenizer
    toks = {name: Tokenizer.from_pretrained(name) for name in VOCABS}
    text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=120)

    @settings(max_examples=150, derandomize=True, deadline=None)
    @given(st.lists(text, max_size=24), st.sampled_from(VOCABS))
    def prop(texts, name):
        t = toks[name]
        enc = t.encode_batch(texts)
        assert t.decode_batch(enc) == texts
        assert t.encode_batch(texts[::-1]) == enc[::-1]
        if texts:
            assert t.encode(texts[0]) == enc[0]

    prop()


def test_latency_path_equals_the_pipeline_and_the_oracle(coracle):
    """Batches of at most 4 KB / 256 documents take encode_small (csrc/spl_api.hip: text read from pinned host memory, completion by a
    word the host spins on): the same ids as the chunk pipeline and as the oracle, at every size up to and across the limit, for
    several documents, empty ones, special tokens, multi-byte text cut by the limit; and the counter says which calls took it.
    Reference: Tokenizer::encode, src/core/tokenizer.rs:729-808."""
    import random
    from splintr_amd import Tokenizer, corpus, _ffi
    from test_gpu_parity import oracle_csr
    L = _ffi.lib()
    rng = random.Random(31)
    base = "".join(corpus.c2(8, seed=5)) + "".join(corpus.c3(2, seed=6))
    for name in ("cl100k_base", "o200k_base", "deepseek_v3"):
        t = Tokenizer.from_pretrained(name)
        orc = coracle(name)
        cases = [[base[:n]] for n in (1, 2, 13, 100, 799, 800, 801, 1600, 2500, 4000)]
        cases += [[base[k:k + 4096].encode("utf-8")[:4096].decode("utf-8", "ignore")] for k in (0, 3000, 9000)]      # at the limit
        cases += [[base[:5000]], [base[10000:10000 + 4200]]]                                                  # beyond it: the pipeline
        cases += [["", "a", "", base[:300], "", "\n", base[300:900]], [base[i * 10:i * 10 + 9] for i in range(256)], ["x"] * 256, ["x"] * 257]
        cases += [["<|endoftext|>" + base[:200] + "<|endoftext|>", "a<|endoftext|>"]]
        for texts in cases:
            for special in (False, True):
                n_bytes = sum(len(x.encode("utf-8")) for x in texts)
                want_ids, want_off = oracle_csr(orc, texts, special)
                got = {}
                for on, fuse in ((1, 1), (1, 0), (0, 1)):          # (fuse: ONE launch, the tiles place their part of the CSR themselves; 0: k_pretok + k_tile_out)
                    assert L.spl_set_option(t.handle, b"small_path", on) == 0 and L.spl_set_option(t.handle, b"fuse", fuse) == 0
                    before = L.spl_small_path_calls(t.handle)
                    ids, off = t.encode_batch_csr(texts, with_special=special)
                    took = L.spl_small_path_calls(t.handle) - before
                    assert took == (1 if on and 0 < n_bytes <= 4096 and len(texts) <= 256 else 0), (n_bytes, len(texts), on, took)
                    assert np.array_equal(off, want_off) and np.array_equal(ids, want_ids), (name, n_bytes, len(texts), special, on)
        L.spl_set_option(t.handle, b"small_path", 1)
        L.spl_set_option(t.handle, b"fuse", 1)
        for _ in range(300):                                     # many calls in a row (the completion word, the 256-call synchronisation)
            x = base[rng.randrange(0, 20000):][:rng.randrange(1, 1200)]
            assert t.encode(x) == orc.encode_batch([x])[0]
