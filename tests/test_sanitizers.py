"""SURVEY 5 / VERDICT r02: the threaded host code under sanitizers, on CPU.
  * the C oracle's batch path (pthread pool pulling documents off a shared counter, mutex-guarded memo):
    make -C oracle asan tsan  -> AddressSanitizer + UBSan and ThreadSanitizer builds of oracle.c + san_driver.c;
  * the product's host-only code -- table builder, tiktoken parser, the host splitter writing shared bitmaps from
    several threads -- as tests/san/host_san.cpp, built with g++ -fsanitize=address,undefined and -fsanitize=thread.
(The host pipeline of spl_encode_batch needs a GPU: tests/test_gpu_sanitizers.py runs it under ThreadSanitizer.)"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

DATA = os.path.join(ROOT, "splintr_amd", "data")
SAN = os.path.join(ROOT, "tests", "san")


@pytest.fixture(scope="module")
def corpus_file(tmp_path_factory):
    from splintr_amd import corpus
    from fuzzgen import fuzz_corpus
    texts = corpus.c2(150) + corpus.c3(10) + fuzz_corpus(11, 400, 30) + ["", "a" * 3000, " " * 2000]
    bs = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    p = tmp_path_factory.mktemp("san") / "corpus.bin"
    with open(p, "wb") as f:
        f.write(np.uint64(len(bs)).tobytes() + off.tobytes() + b"".join(bs))
    return str(p)


def _run(exe, corpus_file, env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([exe, os.path.join(DATA, "cl100k_base.splv"), os.path.join(DATA, "unicode_classes.bin"), corpus_file],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    for word in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:", "LeakSanitizer"):
        assert word not in p.stderr, p.stderr[-3000:]
    return p.stdout


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_oracle_batch_path_under_sanitizers(corpus_file, kind):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), kind])
    out = _run(os.path.join(ROOT, "oracle", "_build", "oracle_" + kind), corpus_file,
               {"ASAN_OPTIONS": "detect_leaks=1", "TSAN_OPTIONS": "halt_on_error=0"})
    assert out.count("checksum") == 4


@pytest.mark.parametrize("kind, flags", [("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                                         ("tsan", ["-fsanitize=thread"])])
def test_product_host_code_under_sanitizers(corpus_file, kind, flags):
    os.makedirs(os.path.join(SAN, "_build"), exist_ok=True)
    exe = os.path.join(SAN, "_build", "host_san_" + kind)
    csrc = os.path.join(ROOT, "splintr_amd", "csrc")
    srcs = [os.path.join(SAN, "host_san.cpp"), os.path.join(csrc, "spl_tables.cpp"), os.path.join(csrc, "spl_regex.cpp")]
    deps = srcs + [os.path.join(csrc, h) for h in ("spl_regex.h", "spl_tables.h", "spl_common.h")]
    # staleness by content, not by modification time (arbitrary after a checkout or a snapshot copy)
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    stamp, want = exe + ".srchash", h.hexdigest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not os.path.exists(exe) or have != want:
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fno-omit-frame-pointer"] + flags + ["-o", exe] + srcs)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    out = _run(exe, corpus_file, {"ASAN_OPTIONS": "detect_leaks=1"})
    assert out.count("pattern ok") == 3 and "unsalted groups 0" in out
