"""Dev aid: kernel-only time of the multi-byte configs (X1, X2, C3) for the library in place."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import random
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
dev = torch.device("cuda", 0)
rng = random.Random(7)
cjk = [corpus.cjk(rng, 4200)[:4096] for _ in range(2500)]
for label, vocab, texts in (("X1", "cl100k_base", cjk[:250]), ("X2", "o200k_base", cjk), ("C3", "o200k_base", corpus.c3(2500))):
    tok = Tokenizer.from_pretrained(vocab)
    b = DeviceBatch(texts, dev)
    reserve(tok, b.n_bytes, b.n_docs)
    for _ in range(5):
        encode_device(tok, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        encode_device(tok, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} {label} {dt * 1e6:9.1f} us {b.n_bytes / dt / 1e9:6.2f} GB/s")
