"""Dev aid: kernel-only throughput (HBM-resident) of the o200k-family configurations, for A/B of the library in place."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
for name, vocab, texts in (("c3", "o200k_base", corpus.c3(2500)), ("c4", "llama3", corpus.c4(100000)), ("c5", "deepseek_v3", corpus.c5(8)),
                           ("m2", "mistral_v3", corpus.c2(4000)), ("c2x8", "cl100k_base", corpus.c2(8000))):
    tok = Tokenizer.from_pretrained(vocab)
    b = DeviceBatch(texts, torch.device("cuda", 0))
    reserve(tok, b.n_bytes, b.n_docs)
    for _ in range(3): encode_device(tok, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): encode_device(tok, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(sys.argv[1] if len(sys.argv) > 1 else "", name, vocab, b.n_bytes, f"{dt * 1e6:.0f} us  {b.n_bytes / dt / 1e9:.2f} GB/s")
