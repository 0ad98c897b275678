"""Dev aid: bench batch, steps alternating over TWO handles on two streams (consecutive steps overlap)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
dev = torch.device("cuda", 0)
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 2
texts = corpus.c2(1000)
toks = [Tokenizer.from_pretrained("cl100k_base") for _ in range(nh)]
batches = [DeviceBatch(texts, dev) for _ in range(nh)]
streams = [torch.cuda.Stream(dev) for _ in range(nh)]
for t, b in zip(toks, batches): reserve(t, b.n_bytes, b.n_docs)
def run(k):
    for i in range(k):
        j = i % nh
        with torch.cuda.stream(streams[j]):
            encode_device(toks[j], batches[j])
run(40); torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(400)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 400)
ids0, off0 = result_csr(batches[0]); ids1, off1 = result_csr(batches[-1])
import numpy as np
print(f"{nh} handles/streams: {best * 1e6:.1f} us/step -> {batches[0].n_bytes / best / 1e6:.0f} MB/s; results equal: {np.array_equal(ids0, ids1) and np.array_equal(off0, off1)}")
