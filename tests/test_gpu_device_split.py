"""SURVEY 8f rank 2, "arbitrary pattern", on the GPU: the device splitter (splintr_amd/csrc/spl_rx_split.h -- the host splitter's
program run at every text position, then the walk from the start of the corpus by pointer doubling) leaves bit for bit the two
bitmaps the host splitter leaves (which tests/test_host_regex.py pins to PCRE2), or says that it gave up.
Reference: Tokenizer::new compiles any pattern and encode() walks find_iter's matches (src/core/tokenizer.rs:410-456, 729-808)."""
import os

import numpy as np
import pytest

from fuzzgen import cased_corpus, fuzz_corpus, latin_corpus
from test_host_regex import (DEEPSEEK_LIKE, GPT2_PATTERN, MIXED, QWEN2, SCRIPTS, SCRIPTS_NEG, SPARSE, TIKTOKEN_CL100K, TIKTOKEN_O200K, VARIANT_A,
                             VARIANT_B, WORDS_DIGITS)

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "..", "splintr_amd", "data")
PATTERNS = {"gpt2": GPT2_PATTERN, "variant_a": VARIANT_A, "variant_b": VARIANT_B, "sparse": SPARSE, "mixed": MIXED,
            "tiktoken_cl100k": TIKTOKEN_CL100K, "tiktoken_o200k": TIKTOKEN_O200K, "qwen2": QWEN2, "deepseek_like": DEEPSEEK_LIKE,
            "words_digits": WORDS_DIGITS, "scripts": SCRIPTS, "scripts_neg": SCRIPTS_NEG}


def _blob(name):
    with open(os.path.join(DATA, name + ".splv"), "rb") as f:
        return f.read()


def _both(t, texts):
    """(host starts, host gaps, device starts, device gaps, status) for one packed batch"""
    import torch
    from splintr_amd import _ffi
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    parts = [x.encode("utf-8") if isinstance(x, str) else x for x in texts]
    blob = b"".join(parts)
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
    d_text = torch.from_numpy(np.frombuffer(blob + b"\0" * ((-len(blob)) % 16 + 16), dtype=np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    words = len(blob) // 32 + 2
    st, gp = np.zeros(words, dtype=np.uint32), np.zeros(words, dtype=np.uint32)
    assert L.spl_split_host(t.handle, blob, off.ctypes.data, len(parts), st.ctypes.data, gp.ctypes.data) == 0, _ffi.last_error()
    d_st = torch.full((words + 2,), -1, dtype=torch.int32, device=dev)
    d_gp = torch.full((words + 2,), -1, dtype=torch.int32, device=dev)
    d_status = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = L.spl_split_device(t.handle, d_text.data_ptr(), len(blob), d_off.data_ptr(), len(parts), d_st.data_ptr(), d_gp.data_ptr(),
                            d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _ffi.last_error()
    torch.cuda.synchronize()
    return (st, gp, d_st[:words].cpu().numpy().view(np.uint32), d_gp[:words].cpu().numpy().view(np.uint32), int(d_status[0].item()))


def _both_device_only(t, texts):
    """the device half of _both (a text whose host split does not finish)"""
    import torch
    from splintr_amd import _ffi
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    parts = [x.encode("utf-8") for x in texts]
    blob = b"".join(parts)
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
    d_text = torch.from_numpy(np.frombuffer(blob + b"\0" * ((-len(blob)) % 16 + 16), dtype=np.uint8).copy()).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    words = len(blob) // 32 + 4
    d_st, d_gp = torch.zeros(words, dtype=torch.int32, device=dev), torch.zeros(words, dtype=torch.int32, device=dev)
    d_status = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = L.spl_split_device(t.handle, d_text.data_ptr(), len(blob), d_off.data_ptr(), len(parts), d_st.data_ptr(), d_gp.data_ptr(),
                            d_status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _ffi.last_error()
    torch.cuda.synchronize()
    return None, None, None, None, int(d_status[0].item())


def _first_diff(a, b):
    w = int(np.nonzero(a != b)[0][0])
    return w * 32 + int(np.log2((int(a[w]) ^ int(b[w])) & -(int(a[w]) ^ int(b[w]))))


@pytest.mark.parametrize("key", sorted(PATTERNS))
def test_device_bitmaps_equal_the_host_splitter_s(key):
    from splintr_amd import Tokenizer
    t = Tokenizer.from_bytes(_blob("cl100k_base"), PATTERNS[key])
    texts = fuzz_corpus(515, 3000, 40) + latin_corpus(3, 600, 90) + cased_corpus(5, 600, 70)
    texts += ["", " ", "\n", "a", "'", "x's'S'ſ'K'K", "http://a.b/c?d=e www.x.y z", "你好你好 你 好", "a\nb\r\nc", "12345678901" * 9, "a  ", "  \n",
              "it'S 'LL 'ſ 'Ve", "$12 €3 £ -- — ―", "foo_bar1 baz", "ひらがな カタカナ 漢字 한글" * 12, "end  \n", "", "", "x"]
    for batch in (texts, texts[::-1], ["".join(texts[:400])], [x for x in texts if len(x) < 4]):
        st, gp, dst, dgp, status = _both(t, batch)
        assert status == 0, f"the device matcher gave up (status {status})"
        assert np.array_equal(st, dst), f"start bits differ first at byte {_first_diff(st, dst)}"
        assert np.array_equal(gp, dgp), f"gap bits differ first at byte {_first_diff(gp, dgp)}"


@pytest.mark.parametrize("key", sorted(PATTERNS))
def test_device_bitmaps_equal_pcre2_find_iter(key):
    """spl_split_device against the ORACLE directly -- match offsets from PCRE2's find_iter (oracle/pyoracle.py, UTF | UCP) turned into
    the two bitmaps by the rule of include/splintr_hip.h (a start bit where a match or a stretch of uncovered bytes begins, a gap bit
    on every uncovered byte) -- with no splitter of the product in between."""
    from oracle import pyoracle as O
    from splintr_amd import Tokenizer
    if not O.pcre2_available():
        pytest.skip("libpcre2-8 not present")
    t = Tokenizer.from_bytes(_blob("cl100k_base"), PATTERNS[key])
    rx = O.Pcre2Pattern(PATTERNS[key])
    texts = fuzz_corpus(811, 1500, 40) + latin_corpus(13, 300, 90) + cased_corpus(15, 300, 70)
    texts += ["", " ", "\n", "a", "'", "x's'S'\u017f'K'\u212a", "http://a.b/c?d=e www.x.y z", "\u4f60\u597d\u4f60\u597d \u4f60 \u597d", "a\nb\r\nc",
              "12345678901" * 9, "a  ", "  \n", "it'S 'LL '\u017f 'Ve", "$12 \u20ac3 \u00a3 -- \u2014 \u2015", "foo_bar1 baz", "end  \n", "", "x"]
    for batch in (texts, texts[::-1], ["".join(texts[:300])]):
        parts = [x.encode("utf-8") for x in batch]
        n = sum(len(x) for x in parts)
        words = n // 32 + 2
        st, gp = np.zeros(words, dtype=np.uint32), np.zeros(words, dtype=np.uint32)

        def setbit(bm, q):
            bm[q >> 5] |= np.uint32(1 << (q & 31))
        base = 0
        for doc in parts:
            at = 0
            for s, e in rx.find_iter(doc):
                if s > at:                                   # uncovered bytes: ONE start bit, a gap bit each
                    setbit(st, base + at)
                    for q in range(at, s):
                        setbit(gp, base + q)
                setbit(st, base + s)
                at = e
            if at < len(doc):
                setbit(st, base + at)
                for q in range(at, len(doc)):
                    setbit(gp, base + q)
            base += len(doc)
        _, _, dst, dgp, status = _both(t, batch)
        assert status == 0, f"the device matcher gave up (status {status})"
        assert np.array_equal(st, dst), f"start bits differ from PCRE2's first at byte {_first_diff(st, dst)}"
        assert np.array_equal(gp, dgp), f"gap bits differ from PCRE2's first at byte {_first_diff(gp, dgp)}"


def test_invalid_utf8_and_document_boundaries_inside_characters():
    """raw bytes: stray continuation bytes, truncated sequences at document ends, documents of one byte"""
    from splintr_amd import Tokenizer
    rng = np.random.default_rng(99)
    t = Tokenizer.from_bytes(_blob("cl100k_base"), TIKTOKEN_CL100K)
    docs = []
    for i in range(1500):
        n = int(rng.integers(0, 60))
        kind = i % 3
        if kind == 0:
            docs.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))
        elif kind == 1:
            docs.append(("héllo wörld 你好 " * 3).encode()[:n])                  # cut inside characters
        else:
            docs.append(bytes(rng.choice(np.frombuffer(b" a1\n\xe4\xbd\xa0\x80\xf0\x9f'sT", dtype=np.uint8), n)))
    st, gp, dst, dgp, status = _both(t, docs)
    assert status == 0
    assert np.array_equal(st, dst), f"start bits differ first at byte {_first_diff(st, dst)}"
    assert np.array_equal(gp, dgp), f"gap bits differ first at byte {_first_diff(gp, dgp)}"


def test_long_matches_skip_blocks_and_longer_ones_are_reported():
    """hops of 300..1000 bytes jump over whole 256-position blocks; beyond the matcher's reach the status word says so"""
    from splintr_amd import Tokenizer
    t = Tokenizer.from_bytes(_blob("cl100k_base"), GPT2_PATTERN)
    rng = np.random.default_rng(5)
    parts = []
    for i in range(300):
        n = int(rng.integers(200, 1000))
        parts.append((" " * n, "a" * n, "7" * n, "-" * n, "漢" * (n // 3))[i % 5])
        parts.append(" ab 12 ")
    for batch in (["".join(parts)], parts, ["".join(parts[k:k + 7]) for k in range(0, len(parts), 7)]):
        st, gp, dst, dgp, status = _both(t, batch)
        assert status == 0
        assert np.array_equal(st, dst), f"start bits differ first at byte {_first_diff(st, dst)}"
        assert np.array_equal(gp, dgp)
    # digits in pairs never fall into step: every block of a long digit run is open, the walk is carried through them
    t3 = Tokenizer.from_bytes(_blob("cl100k_base"), VARIANT_A)
    st, gp, dst, dgp, status = _both(t3, ["1" * 5000 + " x " + "23" * 3001, "9" * 1025])
    assert status == 0 and np.array_equal(st, dst) and np.array_equal(gp, dgp)
    _, _, _, _, status = _both(t, ["x" * 5000 + " tail"])
    assert status != 0
    # ... and so is a stretch of more than a thousand blocks that never close (every block would walk back through all of it)
    _, _, _, _, status = _both(t3, ["7" * 400000 + " x"])
    assert status != 0
    ids, off = t3.encode_batch_csr(["7" * 400000 + " x", "ab 12"])
    assert int(off[-1]) == len(ids) and len(ids) > 100000


def test_encode_batch_uses_the_device_splitter_and_falls_back_by_itself():
    from splintr_amd import Tokenizer, _ffi
    L = _ffi.lib()
    t = Tokenizer.from_bytes(_blob("cl100k_base"), GPT2_PATTERN)
    h = Tokenizer.from_bytes(_blob("cl100k_base"), GPT2_PATTERN)
    assert L.spl_set_option(h.handle, b"device_split", 0) == 0
    texts = fuzz_corpus(8, 2000, 60) + latin_corpus(2, 500, 200)
    a_ids, a_off = t.encode_batch_csr(texts)
    b_ids, b_off = h.encode_batch_csr(texts)
    assert np.array_equal(a_off, b_off) and np.array_equal(a_ids, b_ids)
    assert L.spl_device_split_fallbacks(t.handle) == 0
    long_texts = texts + ["y" * 70000 + " z"]
    a_ids, a_off = t.encode_batch_csr(long_texts)
    b_ids, b_off = h.encode_batch_csr(long_texts)
    assert np.array_equal(a_off, b_off) and np.array_equal(a_ids, b_ids)
    assert L.spl_device_split_fallbacks(t.handle) == 1
    # several pipeline chunks
    assert L.spl_set_option(t.handle, b"chunk_bytes", 1 << 16) == 0 and L.spl_set_option(t.handle, b"single_chunk_max_bytes", 1 << 16) == 0
    big = texts * 6
    a_ids, a_off = t.encode_batch_csr(big)
    b_ids, b_off = h.encode_batch_csr(big)
    assert np.array_equal(a_off, b_off) and np.array_equal(a_ids, b_ids)
    assert L.spl_device_split_fallbacks(t.handle) == 1


def test_one_long_match_costs_one_document_not_the_batch():
    """VERDICT r04 #3a: a C3-sized batch (40 MB, five pipeline chunks) in which ONE document holds a 64 KB run of '=' and another an 8 KB
    base64 line.  The device splitter gives up at the positions of the run (a match longer than its reach); only the documents those
    positions lie in are split on the host cores, their bits patched, every other document keeps the device split: the fallback counter
    counts DOCUMENTS (<= 2), the ids equal the host splitter's, and the call takes about what the batch without the two takes.  Also
    through the one-chunk path (optimistic run + one more tile pass) and the device-text entry point."""
    import base64
    import time
    import torch
    from splintr_amd import Tokenizer, corpus, _ffi
    from splintr_amd.device import DeviceBatch
    L = _ffi.lib()
    t = Tokenizer.from_bytes(_blob("o200k_base"), GPT2_PATTERN)
    h = Tokenizer.from_bytes(_blob("o200k_base"), GPT2_PATTERN)
    assert L.spl_set_option(h.handle, b"device_split", 0) == 0
    docs = corpus.c3(10000)
    b64 = base64.b64encode(bytes(range(256)) * 24).decode()
    assert len(b64) == 8192
    bad = list(docs)
    bad[1234] = bad[1234][:700] + "\n" + "=" * 65536 + "\n" + bad[1234][700:]
    bad[7777] = bad[7777][:1500] + " " + b64 + " " + bad[7777][1500:]
    want = h.encode_batch_csr(bad)

    def timed(tok, texts, n=5):
        tok.encode_batch_csr(texts)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = tok.encode_batch_csr(texts)
            ts.append(time.perf_counter() - t0)
        return r, sorted(ts)[n // 2]
    _, t_clean = timed(t, docs)
    assert L.spl_device_split_fallbacks(t.handle) == 0
    before = L.spl_device_split_fallbacks(t.handle)
    got, t_bad = timed(t, bad)
    per_call = (L.spl_device_split_fallbacks(t.handle) - before) / 6
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    assert 1 <= per_call <= 2, per_call                      # documents, not batches; the base64 line is no long MATCH under this pattern
    print(f"40 MB batch: {t_clean * 1e3:.2f} ms without, {t_bad * 1e3:.2f} ms with the two documents ({t_bad / t_clean:.3f} x)")
    assert t_bad <= 1.25 * t_clean, (t_clean, t_bad)
    # (how much of that is the fallback, how much the 64 KB chunk itself -- a single-class run is merged a rank per round, milliseconds
    #  whoever split it: the same two batches through the built-in pattern's scanner, which has no fallback)
    tb = Tokenizer.from_pretrained("o200k_base")
    _, tb_clean = timed(tb, docs)
    _, tb_bad = timed(tb, bad)
    print(f"built-in o200k_base pattern, same batches: {tb_clean * 1e3:.2f} ms without, {tb_bad * 1e3:.2f} ms with ({tb_bad / tb_clean:.3f} x)")
    # one chunk (the optimistic path): a 1 MB batch with the run in it
    small = docs[:250]
    small[100] = small[100][:300] + "=" * 3000 + small[100][300:]
    before = L.spl_device_split_fallbacks(t.handle)
    a, b = t.encode_batch_csr(small), h.encode_batch_csr(small)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and L.spl_device_split_fallbacks(t.handle) - before == 1
    # the device-text entry point: the documents' text comes to the host, nothing else
    dev = torch.device("cuda", 0)
    db = DeviceBatch(small, dev)
    db.ids.fill_(-1)
    before = L.spl_device_split_fallbacks(t.handle)
    rc = L.spl_encode_batch_device(t.handle, db.text.data_ptr(), db.n_bytes, db.doc_off.data_ptr(), db.n_docs, 0, db.ids.data_ptr(), db.ids.numel(),
                                   db.out_off.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _ffi.last_error()
    torch.cuda.synchronize()
    off = db.out_off.cpu().numpy().astype(np.uint64)
    assert np.array_equal(off, b[1]) and np.array_equal(db.ids[:int(off[-1])].cpu().numpy().view(np.uint32), b[0])
    assert L.spl_device_split_fallbacks(t.handle) - before == 1
    # ... and with special tokens in the documents around (the literals come from the GPU's scan, the host splits around them)
    sp = {"<|endoftext|>": 200100}
    ts_, hs_ = Tokenizer.from_bytes(_blob("o200k_base"), GPT2_PATTERN, sp), Tokenizer.from_bytes(_blob("o200k_base"), GPT2_PATTERN, sp)
    assert L.spl_set_option(hs_.handle, b"device_split", 0) == 0
    mixed = [x + "<|endoftext|>" for x in small]
    mixed[100] = mixed[100][:200] + "<|endoftext|>" + mixed[100][200:]
    a, b = ts_.encode_batch_csr(mixed, with_special=True), hs_.encode_batch_csr(mixed, with_special=True)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and L.spl_device_split_fallbacks(ts_.handle) == 1


def test_a_pattern_that_backtracks_without_end_is_given_up_quickly_and_reported_by_the_host():
    """(?:a|aa)+b over a long run of a's: every attempt runs into the device matcher's step limit; the first one to do so stops the rest, and
    the host splitter (its own budget: 64 n + 10^6 steps per attempt) raises the error the reference's engines would turn into a timeout"""
    import time
    from splintr_amd import Tokenizer
    t = Tokenizer.from_bytes(_blob("cl100k_base"), r"(?:a|aa)+b|[^a]")
    texts = ["a" * 60 + " x"] * 4000
    t0 = time.time()
    _, _, _, _, status = _both_device_only(t, texts)
    assert status != 0 and time.time() - t0 < 20.0
    with pytest.raises(Exception, match="matching budget|backtracking"):
        t.encode_batch(texts[:50])


def test_random_patterns_device_bitmaps_equal_the_host_splitter_s():
    """Random patterns (tests/patgen.py: the CPU suite pins the host splitter to PCRE2 on the same generator) -- what regex_device_image calls a
    SIMPLE alternative, which tail it gives it and what it leaves to the matcher program is decided per pattern: here every accepted pattern's
    device bitmaps are compared with the host's on texts that make overlapping class sets backtrack."""
    import random
    from splintr_amd import Tokenizer
    from patgen import random_pattern, random_texts
    rng = random.Random(424242)
    blob = _blob("cl100k_base")
    done = gave_up = 0
    for _ in range(260):
        pat = random_pattern(rng)
        try:
            t = Tokenizer.from_bytes(blob, pat)
        except ValueError:
            continue
        if not t.has_custom_pattern:
            continue
        texts = random_texts(rng, 300, 30)
        for batch in (texts, ["".join(texts)]):
            try:
                st, gp, dst, dgp, status = _both(t, batch)
            except AssertionError as e:
                assert "budget" in str(e), (pat, str(e)[:300])      # (a pattern that backtracks without end on this text: the host says so)
                gave_up += 1
                continue
            if status != 0:
                gave_up += 1                 # (a deep stack: reported, the host splitter's turn)
                continue
            assert np.array_equal(st, dst), f"{pat!r}: start bits differ first at byte {_first_diff(st, dst)}"
            assert np.array_equal(gp, dgp), f"{pat!r}: gap bits differ first at byte {_first_diff(gp, dgp)}"
        done += 1
    assert done >= 100 and gave_up <= done // 4, (done, gave_up)


def test_special_tokens_are_found_by_the_gpu_scan_and_the_device_splitter_works_between_them():
    """encode_batch_with_special on a custom-pattern handle: the literals come from the GPU's own scan (k_special_scan / the general matcher),
    the device splitter takes each literal as a stretch of dropped bytes with the text ending in front of it and beginning anew behind it
    (tokenizer.rs:842-874) -- same ids as with the split kept on the host, and nothing fell back"""
    import random
    from splintr_amd import Tokenizer, _ffi
    L = _ffi.lib()
    for sp in ({"<|endoftext|>": 100257, "<|fim|>": 100258}, {"<|a|>": 100300, "<|a|>x": 100301, "<|endoftext|>": 100257, "\n\n": 100302}):
        t = Tokenizer.from_bytes(_blob("cl100k_base"), TIKTOKEN_CL100K, sp)
        h = Tokenizer.from_bytes(_blob("cl100k_base"), TIKTOKEN_CL100K, sp)
        assert L.spl_set_option(h.handle, b"device_split", 0) == 0
        rng = random.Random(11)
        lits = list(sp)
        texts = []
        for x in fuzz_corpus(31, 1500, 30) + latin_corpus(5, 300, 60):
            for _ in range(rng.choice([0, 1, 1, 2])):
                c = rng.randrange(len(x) + 1)
                x = x[:c] + rng.choice(lits) + x[c:]
            texts.append(x)
        texts += ["<|endoftext|>", "<|a|>x<|a|>", "a<|fim|>", "<|fim|>a", "<|fi", "<|a|><|a|>x<|endoftext|>tail", "", "<|endoftext|>" * 40, " <|fim|> "]
        for batch in (texts, ["".join(texts[:300])]):
            a = t.encode_batch_with_special(batch)
            b = h.encode_batch_with_special(batch)
            assert a == b
            assert t.encode_batch(batch) == h.encode_batch(batch)
        assert L.spl_device_split_fallbacks(t.handle) == 0


def test_two_pipelines_on_one_gpu_each_with_its_own_splitter_state():
    """spl_set_devices with the ordinal listed twice: two lanes, each context with its own image, workspace, status words and generation"""
    from splintr_amd import Tokenizer, _ffi
    L = _ffi.lib()
    sp = {"<|endoftext|>": 100257}
    t = Tokenizer.from_bytes(_blob("cl100k_base"), QWEN2, sp).set_devices([0, 0])
    h = Tokenizer.from_bytes(_blob("cl100k_base"), QWEN2, sp)
    assert L.spl_set_option(h.handle, b"device_split", 0) == 0
    texts = (fuzz_corpus(77, 3000, 60) + latin_corpus(9, 2000, 300)) * 4
    texts = [x + ("<|endoftext|>" if i % 7 == 0 else "") for i, x in enumerate(texts)]
    assert sum(len(x.encode()) for x in texts) > (3 << 20)
    for special in (False, True):
        for _ in range(3):                                      # (the status words rotate, the generations advance)
            a = t.encode_batch_csr(texts, with_special=special)
            b = h.encode_batch_csr(texts, with_special=special)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert L.spl_device_split_fallbacks(t.handle) == 0
    a = t.encode_batch_csr(texts + ["k" * 3000 + " end"])
    b = h.encode_batch_csr(texts + ["k" * 3000 + " end"])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    assert L.spl_device_split_fallbacks(t.handle) == 1
    a = t.encode_batch_csr(texts)
    assert np.array_equal(a[0], h.encode_batch_csr(texts)[0]) and L.spl_device_split_fallbacks(t.handle) == 1
