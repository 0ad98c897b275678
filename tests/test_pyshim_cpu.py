"""The CPython front end's two halves without a GPU (csrc/spl_pyshim.c): packing list[str] into UTF-8 + offsets
(what PyO3's Vec<String> extraction is to the reference, src/python/bindings.rs:337-339) and building
list[list[int]] from a CSR (PyO3's Vec<Vec<u32>> -> list conversion)."""
import gc
import threading

import numpy as np
import pytest


def _shim():
    from splintr_amd import _ffi
    return _ffi.shim()


def _check(texts):
    b, o = _shim().pack_bytes(texts)
    ref = [t.encode("utf-8") for t in texts]
    assert b == b"".join(ref)
    off = np.frombuffer(o, dtype=np.uint64)
    assert len(off) == len(texts) + 1 and off[0] == 0
    assert np.array_equal(np.diff(off), np.array([len(r) for r in ref], dtype=np.uint64))


def test_pack_small_and_every_str_kind():
    _check([])
    _check([""])
    _check(["ascii", "", "latin1 \xe9\xff", "bmp 你好 €", "astral \U0001f600\U00020000", "mix a\xe9你\U0001f600"])
    _check(("tuple", "works"))


def test_pack_large_batches_take_the_threaded_path():
    from splintr_amd import corpus
    texts = corpus.c4(30000) + ["", "\xe9" * 5, "\U0001f600x", "ab"] + corpus.c3(200)
    _check(texts)                                   # > 8192 strings and > 2 MB: sizes and bytes by helper threads
    _check(["x" * 300] * 9000)                      # the same object many times
    _check(["你" * 100000] * 8 + ["a"] * 10)    # few large strings: byte-balanced ranges


@pytest.mark.parametrize("n", [3, 9000])
def test_pack_errors_match_pyo3_extraction(n):
    sh = _shim()
    pad = ["a"] * n
    with pytest.raises(TypeError, match="cannot be converted to 'PyString'"):
        sh.pack_bytes(pad + [1])
    with pytest.raises(TypeError, match="cannot be converted to 'PyString'"):
        sh.pack_bytes(pad + [b"bytes"])
    with pytest.raises(UnicodeEncodeError):
        sh.pack_bytes(pad + ["lone \ud800 surrogate"])
    with pytest.raises(TypeError, match="Can't extract `str` to `Vec`"):
        sh.pack_bytes("a bare str")
    with pytest.raises(TypeError):
        sh.pack_bytes(5)
    _check(pad)                                     # the staging buffers survive a failed call


def test_lists_from_csr_exact_ints_and_shapes():
    sh = _shim()
    ids = np.array([0, 1, 255, 256, 65535, 65536, 199999, 2 ** 21 - 1, 7, 7, 7], dtype=np.uint32)
    off = np.array([0, 3, 3, 8, 11], dtype=np.uint64)
    r = sh.lists_from_csr(ids.tobytes(), off.tobytes())
    assert r == [[0, 1, 255], [], [256, 65535, 65536, 199999, 2 ** 21 - 1], [7, 7, 7]]
    assert all(type(x) is int for row in r for x in row) and all(type(row) is list for row in r)
    assert sh.lists_from_csr(b"", np.zeros(1, dtype=np.uint64).tobytes()) == []
    with pytest.raises(ValueError):
        sh.lists_from_csr(ids.tobytes(), np.array([0, 5], dtype=np.uint64).tobytes())
    with pytest.raises(ValueError):
        sh.lists_from_csr(ids.tobytes(), np.array([0, 8, 4, 11], dtype=np.uint64).tobytes())


def test_lists_from_csr_leaves_the_collector_as_it_found_it():
    sh = _shim()
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 100000, size=400000, dtype=np.uint32)
    off = np.arange(0, 400001, 40, dtype=np.uint64)
    for on in (True, False):
        gc.enable() if on else gc.disable()
        try:
            r = sh.lists_from_csr(ids.tobytes(), off.tobytes())
            assert gc.isenabled() is on
            assert len(r) == 10000 and r[123] == ids[123 * 40:124 * 40].tolist()
        finally:
            gc.enable()


def test_calls_from_several_threads_do_not_mix_their_batches():
    """the staging buffers are process-wide; every entry point stages AND consumes them inside one
    call that never releases the GIL, so concurrent callers serialise instead of overwriting each other."""
    sh = _shim()
    batches = [[f"thread {k} text {i} " * (1 + (i + k) % 7) for i in range(3000)] for k in range(4)]
    want = [b"".join(t.encode() for t in bt) for bt in batches]
    bad = []

    def work(k):
        for _ in range(20):
            b, _o = sh.pack_bytes(batches[k])
            if b != want[k]:
                bad.append(k)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad
