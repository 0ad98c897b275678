"""the reference's on-disk vocabulary format (tiktoken text, `base64 rank` lines,
src/core/vocab.rs:57-89) through spl_create, and the reference's constructors on it
(src/python/bindings.rs:70-83, 174-187).  The parser runs on the host before anything touches a
GPU, so its error behaviour is checked without one; the -m gpu part builds tokenizers from
tiktoken text regenerated from the packed containers and compares ids with from_pretrained."""
import base64
import ctypes
import os

import pytest

from conftest import ROOT

DATA = os.path.join(ROOT, "splintr_amd", "data")


def tiktoken_text(name: str) -> bytes:
    from oracle import pyoracle as O
    enc, _ = O.load_splv(os.path.join(DATA, name + ".splv"))
    return b"".join(base64.b64encode(k) + b" " + str(v).encode() + b"\n" for k, v in enc.items())


def _create(blob: bytes, pattern: int = 0, flags: int = 0):
    from splintr_amd import _ffi
    L = _ffi.lib()
    with open(os.path.join(DATA, "unicode_classes.bin"), "rb") as f:
        ucls = f.read()
    opts = _ffi.SplOpts(pattern, 0, flags)
    h = L.spl_create(blob, len(blob), ucls, len(ucls), ctypes.byref(opts))
    err = _ffi.last_error()
    if h:
        L.spl_destroy(h)
    return bool(h), err


def _parsed(ok, err):
    """The file was taken: the handle exists (a GPU box), or the one thing that failed is the device this box lacks."""
    return ok or "hipSetDevice" in err or "ROCm-capable" in err


def test_parser_errors_are_the_reference_s():
    ok, err = _create(b"SGVsbG8= 0\nnospace\n")
    assert not ok and "Missing space separator" in err                 # vocab.rs:69-72
    ok, err = _create(b"SGVsbG8 0\n")
    assert not ok and "base64" in err                                  # not canonical base64 (STANDARD engine)
    ok, err = _create(b"SGVs*G8= 0\n")
    assert not ok and "base64" in err
    ok, err = _create(b"SGVsbG8= twelve\n")
    assert not ok and "Invalid rank" in err                            # vocab.rs:80-83
    ok, err = _create(b"SGVsbG8= 12\xc2\xa0\n")                        # str::trim takes Unicode White_Space (NBSP) off the rank:
    assert _parsed(ok, err)                                            # the line parses
    ok, err = _create("SGVsbG8= \u3000\u200312\u2028\n".encode())
    assert _parsed(ok, err)
    ok, err = _create(b"SGVsbG8= 1\xc2\xa02\n")                        # ... but not out of its middle
    assert not ok and "Invalid rank" in err
    ok, err = _create(b"SGVsbG8= 99999999999\n")
    assert not ok and "Invalid rank" in err                            # does not fit u32
    ok, err = _create(b"")
    assert not ok and "empty vocabulary" in err
    ok, err = _create(b"SGVsbG8= 0\n")                                 # one key and no single byte: a vocabulary (bpe.rs:203-215 has three bytes)
    assert _parsed(ok, err)
    ok, err = _create(b"YQ== 5\nYg== 5\n")                              # two keys, one id: refused (see the header)
    assert not ok and "share the id 5" in err
    ok, err = _create(b"YQ== 3000000\n")
    assert not ok and "21 bits" in err
    ok, err = _create(b"SPLVjunkjunkjunkjunkjunk")
    assert not ok and "container" in err


def test_well_formed_text_passes_the_parser():
    blob = tiktoken_text("cl100k_base")
    assert blob.count(b"\n") == 100256
    # blank lines and CRLF line ends are accepted (lines are split at \n, the rank is trimmed: vocab.rs:61-83)
    lines = blob.split(b"\n")
    text = b"\r\n".join(lines[:3]) + b"\r\n\n\n" + b"\n".join(lines[3:])
    ok, err = _create(text)
    # without a GPU the call fails LATER, at the device; with one it succeeds
    assert ok or "hip" in err.lower(), err
    # ... but a blank between the rank and the line end makes the LAST space the separator, as in the reference
    ok, err = _create(lines[0] + b" \n" + b"\n".join(lines[1:]))
    assert not ok and "base64" in err


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cl100k_base", "deepseek_v3"])
def test_constructors_on_tiktoken_text(tmp_path, name):
    import json
    from splintr_amd import Tokenizer, CL100K_BASE_PATTERN, LLAMA3_PATTERN
    from fuzzgen import fuzz_corpus
    pattern = CL100K_BASE_PATTERN if name == "cl100k_base" else LLAMA3_PATTERN
    byte_level = name == "deepseek_v3"
    with open(os.path.join(DATA, "special_tokens.json"), encoding="utf-8") as f:
        special = json.load(f)[name]
    blob = tiktoken_text(name)
    path = tmp_path / (name + ".tiktoken")
    path.write_bytes(blob)
    ref = Tokenizer.from_pretrained(name)
    texts = fuzz_corpus(11, 1500) + ["Hello, world!", "你好世界", "<|endoftext|> x <|im_start|>"]
    want, want_s = ref.encode_batch(texts), ref.encode_batch_with_special(texts)
    made = [Tokenizer(str(path), pattern, special, byte_level=byte_level),                      # bindings.rs:70-83
            (Tokenizer.from_bytes_byte_level if byte_level else Tokenizer.from_bytes)(blob, pattern, special)]   # :174-187
    for t in made:
        assert t.vocab_size == ref.vocab_size
        assert t.encode_batch(texts) == want and t.encode_batch_with_special(texts) == want_s
        assert t.decode_batch(want[:50]) == texts[:50]
    # a later duplicate key replaces the earlier one (vocab.rs:86: encoder.insert)
    # (a FRESH id: two different keys sharing one id is what spl_create refuses, see the header)
    dup = blob + base64.b64encode(b"Hello") + b" 250000\n"
    t = (Tokenizer.from_bytes_byte_level if byte_level else Tokenizer.from_bytes)(dup, pattern)
    assert t.encode("Hello") == [250000] and t.vocab_size == 250001
    with pytest.raises(ValueError):
        Tokenizer.from_bytes(blob + base64.b64encode(b"Hello") + b" 77\n", pattern)   # id 77 already names another key
    # errors: the constructor raises IOError for everything (bindings.rs:79-80), from_bytes ValueError (:183-184)
    with pytest.raises(IOError):
        Tokenizer(str(tmp_path / "missing.tiktoken"), pattern)
    bad = tmp_path / "bad.tiktoken"
    bad.write_bytes(b"not base64!! 0\n")
    with pytest.raises(IOError):
        Tokenizer(str(bad), pattern)
    with pytest.raises(ValueError):
        Tokenizer.from_bytes(b"not base64!! 0\n", pattern)
    with pytest.raises(ValueError):
        Tokenizer.from_bytes(blob, r"\p{Alphabetic}+|\s+")              # a construct the matcher refuses (binary properties; scripts are accepted since round 5)
    th = (Tokenizer.from_bytes_byte_level if byte_level else Tokenizer.from_bytes)(blob, r"\p{Han}+|\p{Latin}+|\s+|.")
    assert th.has_custom_pattern and th.decode(th.encode("漢字 and Latin")) == "漢字 and Latin"
    with pytest.raises(IOError):
        Tokenizer(str(path), r"(?<=a)b|\s+")
    t = (Tokenizer.from_bytes_byte_level if byte_level else Tokenizer.from_bytes)(blob, r"\w+|\s+|[^\w\s]+")   # (round 4: \w is accepted; split on the host cores)
    assert t.has_custom_pattern and t.decode(t.encode("snake_case x1, y")) == "snake_case x1, y"
