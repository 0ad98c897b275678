"""Randomized parity stress in the driver-run suite (-m gpu): fixed seeds of the two generators that found
round 2's window-edge bugs (tests/stressgen.py), every vocabulary, the execution modes a BASELINE config can
reach (0 the size decides / 4 queue mode / 5 tile-owned geometry B) and every forced geometry that is compiled
in.  Bounded: every test stops drawing new seeds after its time budget, but never before its required seeds
(the regression seeds among them) are through."""
import time

import pytest

import test_gpu_parity as tg
from stressgen import custom_batch, edge_batch, random_batch

pytestmark = pytest.mark.gpu

REACHABLE = [0, 0, 0, 4, 5, 5]            # modes a BASELINE config reaches (and their forced forms)
BUDGET_S = 25.0


def _compiled(modes):
    """the forced geometries this build knows (2 and 3 were the multi-pass pipeline, removed in round 4)"""
    import ctypes
    from splintr_amd import _ffi
    st = (ctypes.c_uint64 * 16)()
    h = tg.tok("cl100k_base").handle
    ok = [m for m in modes if _ffi.lib().spl_debug_phases(h, m << 1, st) == 0]
    _ffi.lib().spl_debug_phases(h, 0, st)
    return ok


def _run(gen, seeds, required, coracle, modes):
    modes = _compiled(modes)
    t0 = time.time()
    done = 0
    for k, seed in enumerate(seeds):
        if k >= required and time.time() - t0 > BUDGET_S:
            break
        name, geom, special, texts = gen(seed, modes)
        tg._force_tiles(name, geom)
        try:
            tg.assert_batch_equal(name, texts, coracle, special=special)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} ({gen.__name__}, {name}, mode {geom}, special {special}): {e}") from None
        finally:
            tg._force_tiles(name, 0)
        done += 1
    assert done >= required


def test_random_stress_fixed_seeds(coracle):
    # 22739: the chunk that straddles the window's end (commit 32e0a5b), in the mode the size picks and forced
    for mode in (0, 4, 5):
        _run(random_batch, [22739], 1, coracle, [mode])
    _run(random_batch, list(range(1000, 1400)), 10, coracle, REACHABLE)


def test_random_stress_every_compiled_mode(coracle):
    _run(random_batch, list(range(5000, 5400)), 10, coracle, [0, 1, 2, 3, 4, 5])


def test_edge_sweep_fixed_seeds(coracle):
    # special tokens and empty texts right behind the window's end are part of the generator (commit 6a09bab)
    _run(edge_batch, list(range(1, 400)), 20, coracle, REACHABLE)


def test_edge_sweep_every_compiled_mode(coracle):
    _run(edge_batch, list(range(7000, 7400)), 12, coracle, [0, 1, 3, 4, 5])


_custom = {}


def check_custom(seed):
    """One batch of stressgen.custom_batch through the HIP path and through the Python oracle (same pattern, PCRE2);
    returns None or a description of the first difference."""
    import os
    from conftest import ROOT
    from splintr_amd import Tokenizer, _ffi
    from oracle import pyoracle as O
    vocab, bl, pat, sp, special, texts, (chunk_bytes, single_max) = custom_batch(seed)
    key = (vocab, pat, tuple(sorted(sp.items())))
    if key not in _custom:
        if len(_custom) > 12:
            _custom.clear()
        path = os.path.join(ROOT, "splintr_amd", "data", vocab + ".splv")
        with open(path, "rb") as f:
            blob = f.read()
        enc, _ = O.load_splv(path)
        t = (Tokenizer.from_bytes_byte_level if bl else Tokenizer.from_bytes)(blob, pat, sp)
        _custom[key] = (t, O.Oracle(enc, pat, bl, sp, "pcre2"))
    t, orc = _custom[key]
    L = _ffi.lib()
    assert L.spl_set_option(t.handle, b"chunk_bytes", chunk_bytes) == 0 and L.spl_set_option(t.handle, b"single_chunk_max_bytes", single_max) == 0
    got = t.encode_batch_with_special(texts) if special else t.encode_batch(texts)
    for i, x in enumerate(texts):
        want = orc.encode_with_special(x) if special else orc.encode(x)
        if got[i] != want:
            return f"{vocab} special {special} pattern {pat[:40]!a} text #{i} {x[:60]!a}: {got[i][:12]} vs {want[:12]}"
    return None


def test_custom_pattern_stress_fixed_seeds():
    """2736: a gap (dropped bytes) that outgrows the window was tokenised (round 3); then a fixed block of seeds."""
    from oracle import pyoracle as O
    if not O.pcre2_available():
        pytest.skip("libpcre2-8 not present")
    t0 = time.time()
    for k, seed in enumerate([2736, 2839, 2862] + list(range(1, 400))):
        if k >= 40 and time.time() - t0 > BUDGET_S:
            break
        err = check_custom(seed)
        assert err is None, f"seed {seed}: {err}"
