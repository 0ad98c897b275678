#!/usr/bin/env python3
"""Generates tests/golden/corpus_fixtures.json: SHA-256 of the token-id streams of the synthetic
corpora, computed with the PCRE2-backed PYTHON oracle (oracle/pyoracle.py) -- the layer closest to
the reference's own backends.  Run in the build container; the fixture file is committed."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from splintr_amd import corpus  # noqa: E402

SPECS = {
    "c1_cl100k": dict(vocab="cl100k_base", generator="c1", n=200),
    "c2_cl100k": dict(vocab="cl100k_base", generator="c2", n=1000),
    "c2_wide_cl100k": dict(vocab="cl100k_base", generator="c2_wide", n=1000),
    "c3_o200k": dict(vocab="o200k_base", generator="c3", n=150),
    "c4_llama3": dict(vocab="llama3", generator="c4", n=3000),
    "c5_deepseek": dict(vocab="deepseek_v3", generator="c5", n=2, kwargs={"doc_bytes": 1 << 18}),
}

out = {}
for key, sp in SPECS.items():
    t = O.Oracle.from_pretrained(sp["vocab"], engine="pcre2")
    texts = getattr(corpus, sp["generator"])(sp["n"], **sp.get("kwargs", {}))
    h = hashlib.sha256()
    nt = 0
    for text in texts:
        ids = t.encode(text)
        nt += len(ids)
        h.update(np.asarray(ids, dtype=np.uint32).tobytes())
        h.update(b"|")
    out[key] = dict(sp, sha256=h.hexdigest(), n_tokens=nt, n_bytes=sum(len(x.encode()) for x in texts))
    print(key, out[key])
with open(os.path.join(ROOT, "tests", "golden", "corpus_fixtures.json"), "w") as f:
    json.dump(out, f, indent=1)
