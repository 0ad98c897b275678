"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle on the
same inputs -- bit-exact token ids.  Sizes are chosen so the oracle finishes in seconds; the
full BASELINE.json sizes are covered through size-independent properties at the end."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from conftest import ROOT, VOCABS
from fuzzgen import cased_corpus, fuzz_corpus, invalid_utf8_corpus, latin_corpus

pytestmark = pytest.mark.gpu

_toks = {}


def tok(name):
    from splintr_amd import Tokenizer
    if name not in _toks:
        _toks[name] = Tokenizer.from_pretrained(name)
    return _toks[name]


def oracle_csr(orc, texts, special=False):
    bs = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        np.cumsum([len(b) for b in bs], out=off[1:])
    return orc.encode_packed(np.frombuffer(b"".join(bs), dtype=np.uint8), off, special, threads=os.cpu_count() or 8)


def assert_batch_equal(name, texts, coracle, special=False):
    ids, off = tok(name).encode_batch_csr(texts, with_special=special)
    o_ids, o_off = oracle_csr(coracle(name), texts, special)
    if not np.array_equal(off, o_off) or not np.array_equal(ids, o_ids):
        bad = [i for i in range(len(texts))
               if ids[int(off[i]):int(off[i + 1])].tolist() != o_ids[int(o_off[i]):int(o_off[i + 1])].tolist()]
        i = bad[0]
        raise AssertionError(f"{name}: {len(bad)} of {len(texts)} docs differ; first #{i} {texts[i][:120]!r}: "
                             f"{ids[int(off[i]):int(off[i + 1])][:24].tolist()} vs "
                             f"{o_ids[int(o_off[i]):int(o_off[i + 1])][:24].tolist()}")


def test_extension_is_native_and_loaded():
    from splintr_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH)
    assert _ffi.lib().spl_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libsplintr_hip.so" in f.read()


@pytest.mark.parametrize("name", VOCABS)
def test_reference_vectors(golden, name):
    t = tok(name)
    for text, ids in golden[name]:
        assert t.encode(text) == ids, (name, text)
        assert t.encode_rayon(text) == ids
    assert t.encode_batch([x for x, _ in golden[name]]) == [y for _, y in golden[name]]


@pytest.mark.parametrize("name", VOCABS)
def test_reference_test_strings(name):
    """tests/golden/reference_test_strings.json through the HIP path: single, batch of 700, with special,
    and the backend switches the reference's tests flip (all map to the one scanner)."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_strings.json"), encoding="utf-8") as f:
        fx = json.load(f)
    t = tok(name)
    assert t.encode_batch(fx["plain"]) == fx["ids"][name]
    for s, ids in zip(fx["plain"], fx["ids"][name]):
        assert t.encode(s) == ids and t.pcre2(True).encode(s) == ids and t.jit(False).encode(s) == ids
        assert t.decode(ids) == s
    assert t.encode_batch_with_special(fx["special"]) == fx["ids_with_special"][name]
    want = [fx["ids"][name][fx["plain"].index(s)] for s in fx["large_batch_base"]] * 100
    assert t.encode_batch(fx["large_batch_base"] * 100) == want


def test_mistral_v3_special_ids_held_by_the_reference(golden_all):
    t = tok("mistral_v3")                                     # tests/mistral_v3.rs:27-150
    for text, ids in golden_all["_mistral_v3_with_special"]:
        assert t.encode_with_special(text) == ids, text
        assert t.decode(ids) == text


@pytest.mark.parametrize("name", VOCABS)
@pytest.mark.parametrize("geom", [0, 4, 5])
def test_invalid_utf8_policy(coracle, name, geom):
    """include/splintr_hip.h, "Text that is not valid UTF-8": the C ABI takes raw bytes; stray continuation
    bytes, truncated / over-long sequences and impossible lead bytes -- also right at document boundaries
    (a document ending in a truncated character followed by one that starts with continuation bytes) --
    give the ids of the oracle's restatement of the policy, in every execution mode, and decode back to
    the input bytes."""
    from splintr_amd import Tokenizer, _ffi
    t = Tokenizer.from_pretrained(name)
    assert _ffi.lib().spl_debug_phases(t.handle, geom << 1, None) == 0
    docs = invalid_utf8_corpus(31415 + geom, 3000)
    docs += [b"abc\xe4\xb8", b"\x96\x96 def", b"\xf0\x9f", b"\x8c\x8d", b"", b"\x80", b"\xe4", b"\xb8\x96"]
    docs += [b"\x80" * 3000, b"\xe4\xb8" * 1500, (b"ab\xc3" * 400) + b"\n" + b"\xbf" * 700]
    off = np.zeros(len(docs) + 1, dtype=np.uint64)
    np.cumsum([len(d) for d in docs], out=off[1:])
    blob = b"".join(docs)
    ids, oo = t.encode_packed(blob, off)
    o_ids, o_off = coracle(name).encode_packed(np.frombuffer(blob, dtype=np.uint8), off, threads=os.cpu_count() or 8)
    assert np.array_equal(oo, o_off)
    assert np.array_equal(ids, o_ids)
    got = t._decode_batch_bytes([ids[int(oo[i]):int(oo[i + 1])].tolist() for i in range(len(docs))])
    assert got == docs


@pytest.mark.parametrize("geom", [0, 4])
def test_special_token_sets_whose_occurrences_overlap(geom):
    """A user-supplied special-token map is arbitrary (src/core/tokenizer.rs:304, 429-434): literals that
    contain one another, that chain (a suffix of one is a prefix of another or of itself), and literals
    longer than the fast scan's 32 bytes take the general matcher (k_special_ends / k_special_select) and
    give what Aho-Corasick's Standard, non-overlapping find_iter gives: the earliest-ending occurrence from
    the end of the previous match, the longest one on a tie (restated in oracle/pyoracle.py)."""
    from splintr_amd import Tokenizer, _ffi, CL100K_BASE_PATTERN
    from oracle import pyoracle as O
    special = {"<|a|>": 100300, "<|a|>x": 100301, "a|><": 100302, "|>": 100303, "ab": 100304, "abc": 100305,
               "bcd": 100306, "aa": 100307, "aaa": 100308, "<|" + "long" * 20 + "|>": 100309, "\n\n": 100310, "é": 100311}
    with open(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), "rb") as f:
        blob = f.read()
    t = Tokenizer.from_bytes(blob, CL100K_BASE_PATTERN, special)
    _ffi.lib().spl_debug_phases(t.handle, geom << 1, None)
    enc, _ = O.load_splv(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"))
    o = O.Oracle(enc, O.CL100K_BASE_PATTERN, False, special, engine="regex")
    rng = random.Random(99 + geom)
    atoms = list(special) + ["a", "b", "c", "d", "x", "<", "|", ">", " ", "\n", "hello ", "é", "世界", "<|a", "|>x", "aaaa", "abcd", "<|a|"]
    texts = ["".join(rng.choice(atoms) for _ in range(rng.randint(0, 40))) for _ in range(1500)]
    texts += ["<|a|>x<|a|>", "abcd", "aaaaa", "aaaaaa", "<|a|><|a|>x|>", "a|><|a|>", "<|" + "long" * 20 + "|>!", "", "|>"]
    got = t.encode_batch_with_special(texts)
    for s_, g in zip(texts, got):
        assert g == o.encode_with_special(s_), s_
    # without the flag the literals are plain text, and decode gives the literal back for a special id
    assert t.encode_batch(texts[:50]) == [o.encode(s_) for s_ in texts[:50]]
    assert t.decode([100309]) == "<|" + "long" * 20 + "|>"
    # a literal given twice keeps the later id (a map)
    assert _ffi.lib().spl_add_special(t.handle, b"ab", 2, 100399) == 0
    assert t.encode_with_special("ab") == [100399]


@pytest.mark.parametrize("name", VOCABS)
def test_surface(name):
    t = tok(name)
    sizes = {"cl100k_base": 100331, "o200k_base": 200073, "llama3": 128354, "deepseek_v3": 128954, "mistral_v3": 131126}
    assert t.vocab_size == sizes[name]
    assert repr(t) == f"Tokenizer(vocab_size={sizes[name]})"
    assert t.encode("") == [] and t.encode_batch([]) == [] and t.encode_batch(["", ""]) == [[], []]
    with pytest.raises(TypeError):
        t.encode_batch("not a list")
    with pytest.raises(TypeError):
        t.encode(b"bytes")
    assert t.pcre2(True).encode("Hello") == t.jit(False).encode("Hello") == t.encode("Hello")


@pytest.mark.parametrize("name", VOCABS)
def test_fuzz_batch(coracle, name):
    assert_batch_equal(name, fuzz_corpus(101, 20000, 60), coracle)


@pytest.mark.parametrize("name", VOCABS)
def test_bitvector_starts_latin_text(coracle, name):
    """Text whose multi-byte characters are all letters: the tiles take their match starts from the
    bit-vector computation (spl_scan_starts.h) -- whitespace runs with and without newlines, contractions
    in either case and with U+017F (cl100k: a match of their own; o200k family: a suffix of the letters,
    also back to back), case changes inside a word, newlines and '/' behind "other" runs, number runs of
    every length mod 3, documents from empty to several tiles, so that text starts and ends fall
    everywhere in the windows."""
    rng = random.Random(7)
    texts = latin_corpus(23, 12000, 60) + latin_corpus(24, 300, 1500) + cased_corpus(25, 6000) + cased_corpus(26, 200, 2000)
    texts += ["".join(rng.choice(texts[:2000]) for _ in range(40)) for _ in range(50)]
    assert_batch_equal(name, texts, coracle)
    assert_batch_equal(name, ["".join(texts[:3000])], coracle)          # one document over many tiles


def _long_word_texts(seed, n_docs):
    """Words of 9..64 letters that are no tokens (and some that are), alone and in pairs within a tile."""
    rng = random.Random(seed)
    stems = ["international", "counter", "intuitive", "thermo", "dynamics", "mis", "understand", "ing", "ization", "electro",
             "encephalo", "graphy", "pseudo", "hypo", "para", "thyroid", "ism", "anti", "dis", "establishment", "arian",
             "über", "straße", "ación", "xqzj", "Qwrtp", "_snake_case_", "CamelCase", "0x", "deadbeef", "==", "--"]
    out = []
    for _ in range(n_docs):
        words = []
        for _ in range(rng.randint(1, 60)):
            w = "".join(rng.choice(stems) for _ in range(rng.randint(1, 5)))
            words.append(w[:rng.randint(9, 64)] if len(w) > 9 else w)
            if rng.random() < 0.5:
                words.append(rng.choice(["the", "of", "a", "and", "\n", "to"]))
        out.append(rng.choice(["", " "]) + " ".join(words))
    return out


@pytest.mark.parametrize("geom", [0, 4, 5])
@pytest.mark.parametrize("name", VOCABS)
def test_long_words_merge_two_to_a_wavefront(coracle, name, geom):
    """Chunks of 17..32 bytes share a wavefront two by two (32 lanes each), longer ones take one alone, spans
    of more than 8 bytes rank through the p8 bound or the pair table: tiles with many such words."""
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, _long_word_texts(5, 1500), coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("name", VOCABS)
def test_corpora_and_worst_case(coracle, name):
    from splintr_amd import corpus
    texts = corpus.c1(100) + corpus.c2(300) + corpus.c3(60) + corpus.c4(2000) + corpus.c5(1, doc_bytes=1 << 19)
    assert_batch_equal(name, texts, coracle)
    assert_batch_equal(name, corpus.worst_case(20000), coracle)


def test_corpus_fixtures_sha256():
    from splintr_amd import corpus
    with open(os.path.join(ROOT, "tests", "golden", "corpus_fixtures.json")) as f:
        fx = json.load(f)
    for key, ent in fx.items():
        texts = getattr(corpus, ent["generator"])(ent["n"], **ent.get("kwargs", {}))
        ids, off = tok(ent["vocab"]).encode_batch_csr(texts)
        h = hashlib.sha256()
        for i in range(len(texts)):
            h.update(ids[int(off[i]):int(off[i + 1])].tobytes())
            h.update(b"|")
        assert h.hexdigest() == ent["sha256"], key


def _force_tiles(name, mode):
    """Development hook of the C ABI: 0 auto, 1 small tiles (tile-owned mode when the batch qualifies: 800+192
    up to 1.25 MB, 864+128 beyond), 4 queue mode, 5 tile-owned mode with 864+128 at any size.  (2 and 3 were the
    multi-pass pipeline, removed in round 4.)"""
    import ctypes
    from splintr_amd import _ffi
    st = (ctypes.c_uint64 * 16)()
    rc = _ffi.lib().spl_debug_phases(tok(name).handle, mode << 1, st)
    assert rc == 0


@pytest.mark.parametrize("geom", [1, 5, 4])
@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base"])
def test_tile_and_window_edges(coracle, name, geom):
    """Documents and runs placed around the tile edge and the right halo of EVERY tile geometry (tile-owned
    mode: 800 + 192 at this size, 864 + 128 forced by 5; queue mode: 768 + 224)."""
    rng = random.Random(5)
    texts = []
    for size in (1, 2, 31, 32, 33, 767, 768, 769, 799, 800, 801, 863, 864, 865, 991, 992, 993, 1023, 1024, 1025, 1535, 1536, 1537,
                 1599, 1600, 1601, 1727, 1728, 1729,
                 4095, 4096, 4097, 4575, 4576, 4577, 4607, 4608, 4609, 8191, 8192, 8193):
        texts.append(("ab cd, " * (size // 7 + 1))[:size])
    for lead in (700, 760, 768, 770, 795, 800, 805, 860, 864, 870, 900, 990, 4000, 4090, 4096, 4100, 4500, 4570):
        for run in (" " * 600, "a" * 700, "1" * 500, "=" * 490, "\n" * 481, "你" * 200, " \n" * 300, "x'" * 300,
                    "A" * 500 + "b", "é́" * 200):
            filler = ("lorem ipsum 12 " * 400)[:lead]
            texts.append(filler + run + " tail" + str(rng.randint(0, 9)))
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle)
        assert_batch_equal(name, ["".join(texts)], coracle)       # same content as ONE long document
        assert_batch_equal(name, fuzz_corpus(202, 4000, 60), coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("geom", [0, 4, 5])
@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base"])
def test_chunk_that_straddles_the_window_end(coracle, name, geom):
    """A chain that outgrows its window and ends in a multi-byte character on the window's edge: a run of numbers
    (three to a chunk) closed by U+2167, at every alignment.  The last chunk then reaches up to two bytes beyond
    the window -- its last token may START there --, or ends exactly on the edge, where the next chunk is a sync
    point of the neighbouring tile.  (Found by tools/gpu_stress.py, seed 22739: a token id read from behind
    the tile's id array, a chunk worked twice.)"""
    pad = lambda n: ("lorem ipsum " * 400)[:n]
    _force_tiles(name, geom)
    try:
        for nz, head in ((130, "ⅧⅧⅧⅧ"), (130, "Ⅷ"), (200, "ⅧⅧ"), (240, "ⅧⅧⅧⅧ"), (240, "Ⅷ"), (400, "ⅧⅧⅧⅧ")):
            run = head + "0" * nz + "ⅧⅧ"
            assert_batch_equal(name, [pad(k) + run + " and the end" for k in range(0, 1000)], coracle)
        assert_batch_equal(name, [pad(k) + "a" * 300 + "éé" + " x" for k in range(0, 900)], coracle)
        assert_batch_equal(name, [pad(k) + "-" * 300 + "——" + "\nx" for k in range(0, 900)], coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("geom", [0, 5])
@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base", "mistral_v3"])
def test_text_that_ends_just_behind_the_window_end(coracle, name, geom):
    """The chain's last chunk straddles the window's end and the TEXT ends with it, so that the position the chain
    is continued from is the next text's start: nothing of this chain is left there.  (Found by
    tools/dev/gpu_edge_sweep.py: the next text's first tokens came out twice.)"""
    pad = lambda n: ("lorem ipsum " * 400)[:n]
    _force_tiles(name, geom)
    try:
        for run in ("x'" * 100 + "'ſ'ſ", "'x" * 120 + "'’s'ſ", "0" * 130 + "ⅧⅧ", "a" * 300 + "éé", "-" * 260 + "——"):
            assert_batch_equal(name, [pad(k) + run for k in range(0, 1000)], coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base"])
def test_every_grid_size_of_the_xcd_tile_map(coracle, name):
    """Round 3: workgroup -> tile is a map that gives each of the eight XCDs a contiguous eighth of the tiles, in both
    kernels of the tile-owned mode (`xcd_tile()`); it must be a bijection for ANY tile count.  Batches of 1..41 tiles and
    around 8 x 156 tiles -- every residue of the count mod 8, tile counts below 8 (some XCDs get nothing) -- of the C3 mix
    (English, JSON, CJK: tiles that take the tail and tiles that do not), cut at arbitrary byte counts."""
    from splintr_amd import corpus
    blob = "".join(corpus.c3(420, seed=77, doc_bytes=2600))
    sizes = [800 * k - 137 for k in range(1, 42)] + [800 * k + 311 for k in range(1244, 1254)]
    rng = random.Random(3)
    for nb in sizes:
        at = rng.randrange(0, 4000)
        text = blob.encode("utf-8")[at:at + nb].decode("utf-8", "ignore")
        # a few documents per batch, so that document starts fall into first, middle and last tiles
        cuts = sorted(rng.sample(range(1, max(2, len(text))), min(3, max(0, len(text) - 1))))
        docs = [text[a:b] for a, b in zip([0] + cuts, cuts + [len(text)])]
        assert_batch_equal(name, docs, coracle)


@pytest.mark.parametrize("name", VOCABS)
def test_many_tiny_and_empty_documents(coracle, name):
    rng = random.Random(9)
    texts = []
    for _ in range(30000):
        r = rng.random()
        texts.append("" if r < 0.2 else rng.choice(["a", " ", "\n", "1", "é", "你", "'s", "ab", "a b", "12", "  "])
                     if r < 0.7 else "word " * rng.randint(1, 4))
    assert_batch_equal(name, texts, coracle)


def test_huge_single_chunk(coracle):
    for t in ("a" * 65536, " " * 65536 + "x", "ab" * 20000):
        assert_batch_equal("cl100k_base", [t, "after"], coracle)


@pytest.mark.parametrize("geom", [1, 4, 5])
@pytest.mark.parametrize("name", VOCABS)
def test_oversize_chunks_merge_by_rounds(coracle, name, geom):
    """Chunks beyond the LDS node lists are merged a whole rank at a time (bpe_block_rounds): low-entropy
    runs make long stretches of equal pair ranks (the parity selection inside runs) and merges whose
    new pairs rank below the round's rank (the round must stop there), at lengths either side of
    every capacity, in the tile kernel's tail (1) and in k_bpe_long (3)."""
    rng = random.Random(77)
    texts = []
    for alphabet in ("a", "ab", "abc", "aab", "etaoin", "ABab", "\u4f60\u597d", "\u4f60\u4f60\u597d\u5417", "\u00e9e", " ", "\n",
                     " \n", "=-", "-"):
        for size in (513, 1023, 1025, 2047, 2049, 2500, 5000):
            texts.append("start " + "".join(rng.choice(alphabet) for _ in range(size)) + " end")
    for alphabet, size in (("ab", 70000), ("a", 33333), (" ", 40000), ("abcdefgh", 30000), ("\u4f60\u597d\u5417", 20000)):
        texts.append("".join(rng.choice(alphabet) for _ in range(size)))
    texts.append(("ab" * 3000 + "a" * 3001 + "ba" * 2999) * 3)
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("name", VOCABS)
def test_special_tokens(coracle, name):
    with open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8") as f:
        lits = list(json.load(f)[name])
    rng = random.Random(3)
    base = fuzz_corpus(55, 3000, 20)
    texts = []
    for s in base:
        parts = [s]
        for _ in range(rng.randint(0, 3)):
            parts.insert(rng.randint(0, len(parts)), rng.choice(lits))
        texts.append("".join(parts))
    texts += ["".join(lits), lits[0] * 50, "<|", "<|endoftext", "x<" + lits[0][1:-1], lits[0][:-1] + " " + lits[0][-1]]
    assert_batch_equal(name, texts, coracle, special=True)
    assert_batch_equal(name, texts, coracle, special=False)    # plain encode_batch never scans
    t = tok(name)
    assert t.encode_with_special(texts[0]) == coracle(name).encode_with_special(texts[0])
    assert t.encode_batch_with_special(texts[:50]) == [coracle(name).encode_with_special(x) for x in texts[:50]]


def test_device_api_and_profile(coracle):
    import ctypes
    import torch
    from splintr_amd import _ffi, corpus
    from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
    t = tok("cl100k_base")
    texts = corpus.c2(200)
    batch = DeviceBatch(texts, torch.device("cuda", 0))
    reserve(t, batch.n_bytes, batch.n_docs)
    _ffi.lib().spl_profile_enable(t.handle, 1)
    _ffi.lib().spl_profile_reset(t.handle)
    for _ in range(3):
        encode_device(t, batch)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)()
    n = (ctypes.c_uint64 * 16)()
    _ffi.lib().spl_profile_read(t.handle, ms, n)
    _ffi.lib().spl_profile_enable(t.handle, 0)
    assert max(n) == 3 and sum(ms) > 0
    ids, off = result_csr(batch)
    o_ids, o_off = oracle_csr(coracle("cl100k_base"), texts)
    assert np.array_equal(ids, o_ids) and np.array_equal(off, o_off)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes, through size-independent properties
# ------------------------------------------------------------------------------------------------
def _check_replicated(name, base_texts, copies, coracle):
    """N copies of a base set in shuffled order: every copy of a document must get the ids the
    oracle gives the base document (positions, tile alignment and neighbours all differ)."""
    rng = random.Random(17)
    order = list(range(len(base_texts))) * copies
    rng.shuffle(order)
    texts = [base_texts[i] for i in order]
    ids, off = tok(name).encode_batch_csr(texts)
    o_ids, o_off = oracle_csr(coracle(name), base_texts)
    starts = off[:-1].astype(np.int64)
    lens = np.diff(off.astype(np.int64))
    o_lens = np.diff(o_off.astype(np.int64))
    assert np.array_equal(lens, o_lens[order])
    # checksum of checksums: per-document sums must match the oracle's, then spot-check ids exactly
    csum = np.concatenate([[0], np.cumsum(ids.astype(np.uint64))])
    o_csum = np.concatenate([[0], np.cumsum(o_ids.astype(np.uint64))])
    got = csum[off[1:].astype(np.int64)] - csum[starts]
    want = (o_csum[o_off[1:].astype(np.int64)] - o_csum[o_off[:-1].astype(np.int64)])[order]
    assert np.array_equal(got, want)
    for k in rng.sample(range(len(texts)), min(200, len(texts))):
        i = order[k]
        assert np.array_equal(ids[int(off[k]):int(off[k + 1])], o_ids[int(o_off[i]):int(o_off[i + 1])])
    return sum(len(t.encode("utf-8")) for t in texts)


@pytest.mark.parametrize("mode", [0, 4])          # the mode the size selects (tile-owned), and queue mode
def test_full_size_c3_o200k(coracle, mode):
    from splintr_amd import corpus
    _force_tiles("o200k_base", mode)
    try:
        n = _check_replicated("o200k_base", corpus.c3(500), 20, coracle)           # 10 000 x ~4 KB
        assert n > 35e6
    finally:
        _force_tiles("o200k_base", 0)


@pytest.mark.parametrize("mode", [0, 4])          # the mode the size selects (tile-owned), and queue mode
def test_full_size_c4_llama3(coracle, mode):
    from splintr_amd import corpus
    _force_tiles("llama3", mode)
    try:
        n = _check_replicated("llama3", corpus.c4(10000), 100, coracle)            # 1 000 000 prompts
        assert n > 150e6
    finally:
        _force_tiles("llama3", 0)


@pytest.mark.parametrize("mode", [0, 4])          # the mode the size selects (tile-owned), and queue mode
def test_full_size_c5_deepseek(coracle, mode):
    from splintr_amd import corpus
    _force_tiles("deepseek_v3", mode)
    try:
        n = _check_replicated("deepseek_v3", corpus.c5(2), 50, coracle)            # 100 x 2 MiB
        assert n >= 100 * (2 << 20) - 400
    finally:
        _force_tiles("deepseek_v3", 0)


@pytest.mark.parametrize("name,gen,ndocs,range_tiles,streams", [("cl100k_base", "c2", 12000, 2048, 2), ("cl100k_base", "c2", 12000, 2048, 1),
                                                                ("o200k_base", "c3", 10000, 32768, 2), ("deepseek_v3", "c5", 12, 8192, 2)])
def test_a_large_device_batch_as_ranges_of_its_tiles(coracle, name, gen, ndocs, range_tiles, streams):
    """launch_all sends a device-resident batch of many tiles out as ranges (k_pretok + k_tile_out per range, alternating between the caller's
    stream and a second one): the CSR must be the oracle's, and the one launch pair's, whatever the number of ranges."""
    import torch
    from splintr_amd import Tokenizer, corpus, _ffi
    from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
    L = _ffi.lib()
    texts = getattr(corpus, gen)(ndocs)
    dev = torch.device("cuda", 0)
    b = DeviceBatch(texts, dev)
    o_ids, o_off = oracle_csr(coracle(name), texts)
    got = []
    # (group_scan_min: the groups' prefix sums by k_group_scan -- 1: whatever the number of groups -- or added up by every tile of k_tile_out: 0)
    for rt, gsm in ((range_tiles, 1), (0, 1), (0, 0)):
        t = Tokenizer.from_pretrained(name)
        reserve(t, b.n_bytes + (1 << 20), b.n_docs + 16)
        assert L.spl_set_option(t.handle, b"range_tiles", rt) == 0 and L.spl_set_option(t.handle, b"range_streams", streams) == 0
        assert L.spl_set_option(t.handle, b"group_scan_min", gsm) == 0
        for _ in range(3):                                  # (cold, while the memo fills, warm; every call joins the second stream again)
            encode_device(t, b)
        torch.cuda.synchronize()
        ids, off = result_csr(b)
        assert np.array_equal(off, o_off) and np.array_equal(ids, o_ids), f"range_tiles={rt} group_scan_min={gsm}"
        got.append((ids.copy(), off.copy()))
    assert all(np.array_equal(got[0][0], g[0]) and np.array_equal(got[0][1], g[1]) for g in got[1:])


def test_gatherv_pack_unpack_two_simulated_ranks(coracle):
    """The slab pack / unpack kernels of the ragged all-gather, without a process group: two
    different shards are packed as if by two ranks, their slabs are laid side by side exactly as
    all_gather_into_tensor would, and the unpacked global CSR must equal the oracle's."""
    import torch
    from splintr_amd import _ffi, corpus
    from splintr_amd.device import DeviceBatch, encode_device, reserve
    t = tok("cl100k_base")
    dev = torch.device("cuda", 0)
    shards = [corpus.c2(70, seed=5) + ["", "x"], corpus.c4(300, seed=6)]
    L = _ffi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    max_docs = max(len(s) for s in shards)
    batches = [DeviceBatch(s, dev) for s in shards]
    for bt in batches:
        reserve(t, bt.n_bytes, bt.n_docs)
        encode_device(t, bt)
    torch.cuda.synchronize()
    max_tokens = max(int(bt.out_off[-1].item()) for bt in batches) + 7
    cap = max_tokens + max_docs + 4
    recv = torch.zeros(2 * cap, dtype=torch.int32, device=dev)
    for r, bt in enumerate(batches):
        assert L.spl_gatherv_pack(t.handle, bt.ids.data_ptr(), bt.out_off.data_ptr(), bt.n_docs,
                                  recv[r * cap:].data_ptr(), cap, max_docs, stream) == 0
    all_ids = torch.zeros(2 * max_tokens, dtype=torch.int32, device=dev)
    all_off = torch.zeros(2 * max_docs + 1, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    assert L.spl_gatherv_unpack(t.handle, recv.data_ptr(), 2, cap, max_docs, all_ids.data_ptr(), all_ids.numel(),
                                all_off.data_ptr(), status.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    texts = shards[0] + shards[1]
    o_ids, o_off = oracle_csr(coracle("cl100k_base"), texts)
    n = len(texts)
    assert np.array_equal(all_off[: n + 1].cpu().numpy().astype(np.uint64), o_off)
    assert np.array_equal(all_ids[: int(o_off[-1])].cpu().numpy().view(np.uint32), o_ids)
    # a slab that is too small is reported, not silently truncated
    small = max_docs + 4 + 10
    recv2 = torch.zeros(2 * small, dtype=torch.int32, device=dev)
    for r, bt in enumerate(batches):
        L.spl_gatherv_pack(t.handle, bt.ids.data_ptr(), bt.out_off.data_ptr(), bt.n_docs, recv2[r * small:].data_ptr(),
                           small, max_docs, stream)
    L.spl_gatherv_unpack(t.handle, recv2.data_ptr(), 2, small, max_docs, all_ids.data_ptr(), all_ids.numel(),
                         all_off.data_ptr(), status.data_ptr(), stream)
    torch.cuda.synchronize()
    assert int(status.item()) == 1


@pytest.mark.gpu
def test_gatherv_bucketed_unpack_two_simulated_ranks(coracle):
    """Bucketed form: two ranks x a bucket of depth 3 holding two batches each, laid out
    [rank][slot][slab] as one all_gather_into_tensor of whole buckets would; ONE unpack launch must
    rebuild both batches' global CSRs."""
    import torch
    from splintr_amd import _ffi, corpus
    from splintr_amd.device import DeviceBatch, encode_device, reserve
    t = tok("cl100k_base")
    dev = torch.device("cuda", 0)
    L = _ffi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    # shards[rank][batch]
    shards = [[corpus.c2(40, seed=11) + [""], corpus.c1(25, seed=12)],
              [corpus.c4(200, seed=13), ["solo document", "", "tail"]]]
    depth, world, nb = 3, 2, 2
    enc = [[DeviceBatch(s, dev) for s in row] for row in shards]
    for row in enc:
        for bt in row:
            reserve(t, bt.n_bytes, bt.n_docs)
            encode_device(t, bt)
    torch.cuda.synchronize()
    max_docs = max(bt.n_docs for row in enc for bt in row)
    max_tokens = max(int(bt.out_off[-1].item()) for row in enc for bt in row) + 5
    cap = max_tokens + max_docs + 4
    recv = torch.zeros(world * depth * cap, dtype=torch.int32, device=dev)
    for r in range(world):
        for j in range(nb):
            bt = enc[r][j]
            assert L.spl_gatherv_pack(t.handle, bt.ids.data_ptr(), bt.out_off.data_ptr(), bt.n_docs,
                                      recv[(r * depth + j) * cap:].data_ptr(), cap, max_docs, stream) == 0
    off_stride = world * max_docs + 1
    all_ids = torch.zeros(depth * world * max_tokens, dtype=torch.int32, device=dev)
    all_off = torch.zeros(depth * off_stride, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    assert L.spl_gatherv_unpack_group(t.handle, recv.data_ptr(), world, depth, nb, cap, max_docs, all_ids.data_ptr(),
                                      world * max_tokens, all_off.data_ptr(), off_stride, status.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    for j in range(nb):
        texts = shards[0][j] + shards[1][j]
        o_ids, o_off = oracle_csr(coracle("cl100k_base"), texts)
        g_off = all_off[j * off_stride:(j + 1) * off_stride]
        g_ids = all_ids[j * world * max_tokens:(j + 1) * world * max_tokens]
        assert np.array_equal(g_off[: len(texts) + 1].cpu().numpy().astype(np.uint64), o_off)
        assert np.array_equal(g_ids[: int(o_off[-1])].cpu().numpy().view(np.uint32), o_ids)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [0, 5, 4])
def test_encode_packed_slab_equals_pack_kernel(geom):
    """spl_encode_batch_device_packed must leave the same slab as encode + spl_gatherv_pack, in
    tile-owned mode (the last kernel writes it) and in queue mode (pack kernel queued)."""
    import torch
    from splintr_amd import _ffi, corpus
    from splintr_amd.device import DeviceBatch, encode_device, reserve
    t = tok("cl100k_base")
    dev = torch.device("cuda", 0)
    L = _ffi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    texts = corpus.c2(120, seed=21) + ["", "a" * 3000, " " * 1500 + "x", ""] + corpus.c4(200, seed=22)
    bt = DeviceBatch(texts, dev)
    reserve(t, bt.n_bytes, bt.n_docs)
    _force_tiles("cl100k_base", geom)
    try:
        encode_device(t, bt)
        torch.cuda.synchronize()
        T = int(bt.out_off[-1].item())
        max_docs, cap = bt.n_docs + 3, T + bt.n_docs + 3 + 4 + 11
        ref = torch.zeros(cap, dtype=torch.int32, device=dev)
        assert L.spl_gatherv_pack(t.handle, bt.ids.data_ptr(), bt.out_off.data_ptr(), bt.n_docs, ref.data_ptr(), cap,
                                  max_docs, stream) == 0
        got = torch.zeros(cap, dtype=torch.int32, device=dev)
        ids2 = torch.zeros_like(bt.ids)
        off2 = torch.zeros_like(bt.out_off)
        assert L.spl_encode_batch_device_packed(t.handle, bt.text.data_ptr(), bt.n_bytes, bt.doc_off.data_ptr(), bt.n_docs,
                                                0, ids2.data_ptr(), ids2.numel(), off2.data_ptr(), got.data_ptr(), cap,
                                                max_docs, stream) == 0
        torch.cuda.synchronize()
    finally:
        _force_tiles("cl100k_base", 0)
    assert torch.equal(off2, bt.out_off) and torch.equal(ids2[:T], bt.ids[:T])
    used = 3 + max_docs + T
    r, g = ref[:used].cpu().numpy(), got[:used].cpu().numpy()
    # offsets beyond n_docs + 1 are unspecified padding in both
    assert np.array_equal(r[: 2 + bt.n_docs + 1], g[: 2 + bt.n_docs + 1])
    assert np.array_equal(r[3 + max_docs:], g[3 + max_docs:])


@pytest.mark.gpu
def test_more_than_65536_documents_in_tile_owned_mode(coracle):
    """The tile kernel finds its first document by a 256-ary search of doc_off: more than 65 536
    documents need a third round; runs of empty documents share an offset."""
    rng = random.Random(17)
    words = ["alpha", " beta", "gamma,", " 12345", "δέλτα", " don't", "x", "", "", "\n\n", "  indent", "世界"]
    texts = [rng.choice(words) + (rng.choice(words) if rng.random() < 0.5 else "") for _ in range(140000)]
    assert_batch_equal("cl100k_base", texts, coracle)
    assert_batch_equal("o200k_base", texts[:70000], coracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["o200k_base", "deepseek_v3"])
def test_seven_megabytes_mixed_in_tile_owned_mode(coracle, name):
    """Just under the tile-owned limit (8 MB): several rounds of tiles, CJK-dense tiles with many long
    chunks next to prose and JSON."""
    from splintr_amd import corpus
    texts = corpus.c3(1800, seed=77)
    assert sum(len(t.encode()) for t in texts) < (8 << 20)
    assert_batch_equal(name, texts, coracle)


@pytest.mark.gpu
def test_document_search_with_skewed_document_sizes(coracle):
    """The tile kernel's first search round interpolates on document index; batches whose document
    sizes are far from uniform must fall through to the plain search."""
    big = [("lorem ipsum dolor 42, " * 120)[:2048 + 7 * i % 300] for i in range(320)]
    tiny = [("ab" if i % 3 else "") + str(i % 10) for i in range(4000)]
    for texts in (big + tiny, tiny + big, tiny[:500] + big + tiny[500:] + big[:40]):
        assert_batch_equal("cl100k_base", texts, coracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base"])
def test_queue_mode_with_window_edges_and_long_runs(coracle, name):
    """Queue mode (tile-owned tiles, global queues for long chunks and outgrown chains, CSR assembled
    per tile range; forced here, it is what batches beyond 256 MB run in) on a batch that contains the window-edge documents, runs of every
    class far longer than a window, and many tiny / empty documents."""
    from splintr_amd import corpus
    rng = random.Random(23)
    texts = list(corpus.c2(8600, seed=31))
    for lead in (700, 760, 768, 770, 900, 990, 1000, 1023, 1024, 1500):
        for run in (" " * 600, "a" * 3000, "1" * 500, "=" * 4900, "\n" * 481, "你" * 1200, " \n" * 300, "x'" * 300,
                    "A" * 500 + "b", "é́" * 200):
            filler = ("lorem ipsum 12 " * 400)[:lead]
            texts.insert(rng.randrange(len(texts)), filler + run + " tail" + str(rng.randint(0, 9)))
    for _ in range(3000):
        texts.insert(rng.randrange(len(texts)), rng.choice(["", "a", " ", "\n", "é", "你好", "12", "  "]))
    texts += list(corpus.worst_case(20000))
    assert sum(len(t.encode("utf-8")) for t in texts) > (8 << 20)
    _force_tiles(name, 4)
    try:
        assert_batch_equal(name, texts, coracle)
    finally:
        _force_tiles(name, 0)
    assert_batch_equal(name, texts, coracle)                      # and in the mode the size selects (tile-owned)


def _multibyte_texts(seed, n_docs, doc_bytes):
    """CJK / kana / hangul / emoji text (splintr_amd.corpus.cjk) with everything the segment merge has
    to get right mixed in: runs without punctuation far beyond a table pass (256 rows) and beyond a
    chain window (2048 bytes), Latin words glued to CJK characters (one \\p{L}+ chunk), multi-character
    tokens, repeated characters, lone continuation bytes' neighbours (4-byte emoji), combining marks."""
    from splintr_amd import corpus
    rng = random.Random(seed)
    han = "的一是不了人我在有他这为之大来以个中上们到说国和地也子时道出而要于就下得可你年生自会那后能对着事其里所去行过家十用发天如然作方成者多日都三小军二无同么经法当起与好看学进种将还分此心前面又定见只主没公从"
    kana = "あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんアイウエオカキクケコ"
    hangul = "가나다라마바사아자차카타파하각간갈감강개거건걸검것게결경계고공과관교구국군그극근금기길김나남내너녀년노"
    out = []
    for _ in range(n_docs):
        parts = []
        size = 0
        while size < doc_bytes:
            r = rng.random()
            if r < 0.55:
                s = corpus.cjk(rng, rng.randint(30, 400))
            elif r < 0.65:
                s = "".join(rng.choice(han) for _ in range(rng.choice((40, 90, 200, 700, 1500))))       # no punctuation at all
            elif r < 0.72:
                s = "".join(rng.choice(kana) for _ in range(rng.choice((30, 60, 120, 300))))
            elif r < 0.78:
                s = "".join(rng.choice(hangul) for _ in range(rng.choice((25, 50, 100))))
            elif r < 0.86:
                s = "".join(rng.choice(["data", "Token", "x", "über", "naïve", "Ελλάδα", "мир"]) + rng.choice(han) * rng.randint(1, 3)
                            for _ in range(rng.randint(3, 30)))
            elif r < 0.92:
                s = rng.choice(han) * rng.choice((3, 17, 64, 65, 129, 300)) + rng.choice(["。", "", " "])
            elif r < 0.96:
                s = "".join(rng.choice(["👨‍👩‍👧", "🙂", "🇯🇵", "é", "की", "ไทย"]) for _ in range(rng.randint(1, 40)))
            else:
                s = rng.choice([" ", "\n", "，", "、", "「", "」 ", "123", " the "]) * rng.randint(1, 5)
            parts.append(s)
            size += len(s.encode("utf-8"))
        out.append("".join(parts))
    return out


@pytest.mark.parametrize("geom", [0, 1, 4])
@pytest.mark.parametrize("name", VOCABS)
def test_multibyte_text_merges_by_segments(coracle, name, geom):
    """Multi-byte text goes through the segment merge (bpe_wave64_tab's groups, bpe_tail_segments in the
    tile tail, k_bpe_segments over the global queue in queue mode): ~1.5 MB of it, every
    vocabulary, documents of 200 B to 20 KB so that tile and window edges fall everywhere."""
    texts = _multibyte_texts(41, 60, 20000) + _multibyte_texts(42, 400, 600) + _multibyte_texts(43, 300, 200)
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle)
        assert_batch_equal(name, ["".join(texts[:40])], coracle)       # one document of ~0.8 MB
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("name", ["o200k_base", "deepseek_v3"])
def test_multibyte_text_in_queue_mode(coracle, name):
    """The same through queue mode (forced; it is what batches beyond 256 MB run in): long chunks travel
    the global queue to k_bpe_segments and k_bpe_long."""
    texts = _multibyte_texts(51, 450, 20000)
    assert sum(len(t.encode("utf-8")) for t in texts) > (8 << 20)
    _force_tiles(name, 4)
    try:
        assert_batch_equal(name, texts, coracle)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("geom", [0, 4])
@pytest.mark.parametrize("name", ["cl100k_base", "deepseek_v3"])
def test_multibyte_text_with_special_tokens(coracle, name, geom):
    """Special-token literals inside multi-byte text: their spans cut the chunks that the segment merge
    then works on (tile-owned mode with the skip bitmaps; queue mode has no special-token form and falls back to it)."""
    with open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8") as f:
        lits = list(json.load(f)[name])
    rng = random.Random(61)
    texts = []
    for s in _multibyte_texts(62, 300, 1500):
        cut = sorted(rng.randrange(len(s) + 1) for _ in range(rng.randint(0, 4)))
        parts, prev = [], 0
        for c in cut:
            parts += [s[prev:c], rng.choice(lits)]
            prev = c
        texts.append("".join(parts) + s[prev:])
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle, special=True)
    finally:
        _force_tiles(name, 0)


@pytest.mark.parametrize("geom", [0, 4])
@pytest.mark.parametrize("name", VOCABS)
def test_segment_pass_corner_cases(coracle, name, geom):
    """Crafted inputs for the corners of the segment pass: more multi-byte medium chunks in a tile than
    its list takes (16), a chunk cut at the 256-row limit with 0 / 1 / 2 bytes left over, a 65..128-byte
    segment (two nodes per lane), a segment beyond 128 bytes next to a full list (no room to set it
    aside), chunks that end exactly at row 256, and single bytes between hard boundaries."""
    han = "的一是不了人我在有他这为之大来以个中上们到说国和地也子时道出而要于就下得可你年生自会那后能对着事其里所去行过家十用发天如然作方成者多日都三小军二无同么经法当起与好看学进种将还分此心前面又定见只主没公从"
    rng = random.Random(97)
    texts = ["你好世界你好，" * 3000,                                   # ~38 medium chunks per tile
             "".join(rng.choice(han) for _ in range(6)).join(["，"] * 4000)]
    for tail in ("", "q", "qx", "zq", "ab", "the", "é", "éa", "́"):  # a cut chunk and what is left of it
        for k in (84, 85, 86, 170, 171):
            texts.append("x " + "".join(rng.choice(han) for _ in range(k)) + tail + " y")
    for w in (60, 64, 65, 100, 128, 129, 200, 300):                        # one long soft segment inside CJK text
        word = "".join(rng.choice("etaoinshrdlu") for _ in range(w))
        texts.append("你好" * 10 + word + "世界" * 10 + "，" + "你好世界，" * 40 + word)
        texts.append(("你好世界，" * 30 + "a" * w + "。") * 4)
    for pad in range(0, 40, 3):                                            # chunks ending at row 256 and around it
        texts.append("，".join("".join(rng.choice(han) for _ in range(n)) for n in (20 + pad, 30, 35, 25, 40, 33)))
    texts.append("".join(rng.choice(han) + rng.choice("aeiou") for _ in range(2000)))   # 1-byte segments between characters
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle)
        assert_batch_equal(name, ["".join(texts)], coracle)
    finally:
        _force_tiles(name, 0)


def test_special_tokens_in_a_large_tile_owned_batch(coracle):
    """SPL_WITH_SPECIAL beyond 8 MB runs tile-owned too (up to 256 MB): ~9 MB of English/code and
    multi-byte text with literals inserted."""
    from splintr_amd import corpus
    name = "cl100k_base"
    with open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8") as f:
        lits = list(json.load(f)[name])
    rng = random.Random(71)
    texts = list(corpus.c2(6000, seed=72)) + _multibyte_texts(73, 300, 10000)
    for i in range(0, len(texts), 3):
        t = texts[i]
        c = rng.randrange(len(t) + 1)
        texts[i] = t[:c] + rng.choice(lits) + t[c:]
    assert sum(len(t.encode("utf-8")) for t in texts) > (8 << 20)
    assert_batch_equal(name, texts, coracle, special=True)


@pytest.mark.parametrize("geom", [0, 4])
@pytest.mark.parametrize("name", ["cl100k_base", "o200k_base"])
def test_chunks_longer_than_a_chain_window(coracle, name, geom):
    """A chunk that no staged window holds (2 KiB) is a run of one repeated character: tile-owned mode finds
    its end in parallel and lets the scanner see it with the middle cut out.  Runs of 1- to 4-byte
    characters, what follows them (the rules at a run's end differ by class), a run that ends with the
    document or the text, two documents of the same character back to back, runs that are NOT periodic
    (walked from HBM), a periodic run right after a non-periodic one in the same chunk."""
    rng = random.Random(113)
    texts = []
    for ch in ("a", " ", "\n", "=", "\u00e9", "\u4f60", "\ud55c", "\U0001F642", "\u093c"):
        for n, afters in ((2100, ("", " x", "x", "\n", "1", "'s", "\u3002", ch + "b", " " + ch)), (4500, (" x", "x"))):
            for after in afters:
                texts.append("lead " + ch * n + after)
    texts += ["a" * 3000, "a" * 3000, " " * 4000, "", " " * 4000 + "z", "\u4f60" * 1500, "\u4f60" * 1500]
    texts.append("".join(rng.choice("abcdefgh") for _ in range(3000)) + " tail")           # not periodic
    texts.append("".join(rng.choice("abcdefgh") for _ in range(1200)) + "q" * 3000 + "rst uvw")
    texts.append("ab" * 1500 + " " + "xyz" * 1000 + " " + "\u4f60\u597d" * 700)              # period of two / three characters
    texts.append("x" * 2040 + " " + "y" * 2050 + "\n" + "z" * 2300)
    _force_tiles(name, geom)
    try:
        assert_batch_equal(name, texts, coracle)
        assert_batch_equal(name, ["".join(texts)], coracle)
    finally:
        _force_tiles(name, 0)
