"""Dev aid: the bench rotation (1000 x ~1 KB, one batch at a time) launched on torch's null stream against a stream of its own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
dev = torch.device("cuda", 0)
tok = Tokenizer.from_pretrained("cl100k_base")
batches = [DeviceBatch(corpus.c2(1000, seed=1002 + k), dev) for k in range(8)]
reserve(tok, max(b.n_bytes for b in batches), 1000)
nb = sum(b.n_bytes for b in batches) / 8
side = torch.cuda.Stream(dev)
def region(n=20):
    for i in range(n): encode_device(tok, batches[i % 8])
def timed(ctx):
    rs = []
    with ctx:
        for _ in range(10): region()
        for _ in range(25):
            torch.cuda.synchronize(); t0 = time.perf_counter(); region(); torch.cuda.synchronize(); rs.append((time.perf_counter() - t0) / 20)
    rs.sort(); return rs[len(rs) // 2]
import contextlib
for rep in range(3):
    a = timed(contextlib.nullcontext()); b = timed(torch.cuda.stream(side))
    print(f"null stream {a*1e6:.2f} us/step = {nb/a/1e9:.2f} GB/s | side stream {b*1e6:.2f} us/step = {nb/b/1e9:.2f} GB/s")
