"""Dev aid: what makes BASELINE config 4 (1 M chat prompts of ~215 B) slower per byte than long documents -- kernel-only rates of variants of
the same text: as it is; the same prompts joined ten to a document; ASCII-only prompts; both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
dev = torch.device("cuda", 0)
tok = Tokenizer.from_pretrained(sys.argv[1] if len(sys.argv) > 1 else "llama3")
docs = corpus.c4(100000)
asc = [d for d in docs if d.isascii()]
def join(ds, k): return ["\n".join(ds[i:i + k]) for i in range(0, len(ds), k)]
for name, ds in (("as it is", docs), ("ten prompts per document", join(docs, 10)), ("ASCII prompts only", asc), ("ASCII, ten per document", join(asc, 10)),
                 ("non-ASCII prompts only", [d for d in docs if not d.isascii()])):
    b = DeviceBatch(ds, dev)
    reserve(tok, b.n_bytes, b.n_docs)
    for _ in range(3): encode_device(tok, b)
    torch.cuda.synchronize(); ts = []
    for _ in range(15):
        t0 = time.perf_counter(); encode_device(tok, b); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name:28s} {b.n_docs:7d} docs {b.n_bytes:9d} B  {b.n_bytes/ts[7]/1e9:6.2f} GB/s")
