import os, sys, base64
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
import test_gpu_device_split as D
from test_host_regex import GPT2_PATTERN
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
t = Tokenizer.from_bytes(D._blob("o200k_base"), GPT2_PATTERN)
h = Tokenizer.from_bytes(D._blob("o200k_base"), GPT2_PATTERN)
L.spl_set_option(h.handle, b"device_split", 0)
docs = corpus.c3(10000)
b64 = base64.b64encode(bytes(range(256)) * 24).decode()
for variant in ("run", "b64", "both"):
    bad = list(docs)
    if variant in ("run", "both"): bad[1234] = bad[1234][:700] + "\n" + "=" * 65536 + "\n" + bad[1234][700:]
    if variant in ("b64", "both"): bad[7777] = bad[7777][:1500] + " " + b64 + " " + bad[7777][1500:]
    want = h.encode_batch_csr(bad)
    b0 = L.spl_device_split_fallbacks(t.handle)
    got = t.encode_batch_csr(bad)
    fb = L.spl_device_split_fallbacks(t.handle) - b0
    dg, dw = np.diff(got[1].astype(np.int64)), np.diff(want[1].astype(np.int64))
    diff = np.nonzero(dg != dw)[0]
    print(variant, "fallback docs", fb, "docs with other counts:", diff[:10], dg[diff[:5]], dw[diff[:5]], "ids equal:", np.array_equal(got[0], want[0]))
for n_docs, run in ((250, 3000), (250, 65536), (2500, 3000), (2500, 65536)):
    small = list(docs[:n_docs])
    small[100] = small[100][:300] + "=" * run + small[100][300:]
    want = h.encode_batch_csr(small)
    b0 = L.spl_device_split_fallbacks(t.handle)
    got = t.encode_batch_csr(small)
    fb = L.spl_device_split_fallbacks(t.handle) - b0
    dg, dw = np.diff(got[1].astype(np.int64)), np.diff(want[1].astype(np.int64))
    diff = np.nonzero(dg != dw)[0]
    print(n_docs, run, "fallback docs", fb, "docs with other counts:", diff[:10], dg[diff[:5]], dw[diff[:5]], "ids equal:", np.array_equal(got[0], want[0]))
