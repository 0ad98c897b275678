"""Dev aid: step time of encode_batch vs encode_batch_with_special on the bench batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(corpus.c2(1000), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
for sp in (False, True, False, True):
    for _ in range(30): encode_device(tok, batch, with_special=sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): encode_device(tok, batch, with_special=sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"with_special={sp}: {dt * 1e6:.1f} us/step  {batch.n_bytes / dt / 1e6:.0f} MB/s")
