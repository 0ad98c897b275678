#!/usr/bin/env python3
"""Supplementary measurements (not the bench.py line): HBM-resident encode_batch throughput of the
BASELINE.json configs at (scaled) size on ONE GPU, each checked bit-exact against the oracle first.
usage: python tools/bench_configs.py [scale]    scale 1.0 = full C3/C4/C5 sizes (slow to generate)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.coracle import COracle  # noqa: E402
from splintr_amd import Tokenizer, corpus  # noqa: E402
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25


def _cjk_docs(n):
    import random
    rng = random.Random(7)
    return [corpus.cjk(rng, 4200)[:4096] for _ in range(n)]

dev = torch.device("cuda", 0)
CFG = [
    ("C1 cl100k 1000 x ~1 KB English", "cl100k_base", lambda: corpus.c1(1000)),
    ("C2 cl100k 1000 x ~1 KB English/code (bench)", "cl100k_base", lambda: corpus.c2(1000)),
    ("C3 o200k %d x 4 KB prose+JSON+CJK" % int(10000 * scale), "o200k_base", lambda: corpus.c3(int(10000 * scale))),
    ("C4 llama3 %d short prompts" % int(1000000 * scale * 0.25), "llama3", lambda: corpus.c4(int(1000000 * scale * 0.25))),
    ("C5 deepseek_v3 %d x 2 MiB" % max(1, int(100 * scale * 0.25)), "deepseek_v3", lambda: corpus.c5(max(1, int(100 * scale * 0.25)))),
    # not a BASELINE config: the worst case of the segment merge's target, text that is multi-byte throughout
    ("X1 cl100k 250 x 4 KB CJK / kana / hangul only", "cl100k_base", lambda: _cjk_docs(250)),
    ("X2 o200k 2500 x 4 KB CJK / kana / hangul only", "o200k_base", lambda: _cjk_docs(2500)),
]
print(f"{'config':52s} {'MB':>8s} {'tokens':>10s} {'us/step':>10s} {'GB/s':>8s}  parity")
for label, vocab, gen in CFG:
    texts = gen()
    tok = Tokenizer.from_pretrained(vocab)
    batch = DeviceBatch(texts, dev)
    reserve(tok, batch.n_bytes, batch.n_docs)
    encode_device(tok, batch)
    torch.cuda.synchronize()
    ids, off = result_csr(batch)
    o_ids, o_off = COracle(vocab).encode_packed(np.frombuffer(b"".join(t.encode() for t in texts), dtype=np.uint8),
                                                batch.host_offsets, threads=os.cpu_count() or 8)
    ok = np.array_equal(ids, o_ids) and np.array_equal(off, o_off)
    for _ in range(5):
        encode_device(tok, batch)
    torch.cuda.synchronize()
    reps = 50 if batch.n_bytes < 4e6 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        encode_device(tok, batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:52s} {batch.n_bytes / 1e6:8.2f} {int(off[-1]):10d} {dt * 1e6:10.1f} {batch.n_bytes / dt / 1e9:8.2f}  "
          f"{'bit-exact' if ok else 'MISMATCH'}")
    del tok, batch
