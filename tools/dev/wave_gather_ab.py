"""Dev aid: WaveGather at the wave sizes of an 8-GPU run (8 waves of ~3.4 MB per rank), one-rank communicator: one encode stream against two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
import splintr_amd.device as D
from splintr_amd.device import DeviceBatch, encode_device, reserve, Comm, WaveGather
dev = torch.device("cuda", 0)
vocab = "llama3"; parts = [corpus.c4(15600, seed=1004 + k) for k in range(8)]
tok, tok2 = Tokenizer.from_pretrained(vocab), Tokenizer.from_pretrained(vocab)
subs = [DeviceBatch(p, dev) for p in parts]
for t in (tok, tok2): reserve(t, max(b.n_bytes for b in subs), max(b.n_docs for b in subs))
nb = sum(b.n_bytes for b in subs)
for b in subs: encode_device(tok, b)
torch.cuda.synchronize()
ntok = [int(b.out_off[-1].item()) for b in subs]
comm = Comm(Comm.unique_id(), 0, 1, 0)
B2B = int(os.environ.get("B2B", "1"))           # steps back to back between synchronisations
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(B2B): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / B2B)
    ts.sort(); return ts[len(ts) // 2] * 1e3
def mk(t2):
    return WaveGather(tok, dev, comm, 8, max_docs=max(b.n_docs for b in subs), max_tokens=int(max(ntok) * 1.02) + 64,
                      total_tokens_cap=sum(ntok) + 64, total_docs_cap=sum(b.n_docs for b in subs), tok2=t2)
def step(g):
    g.begin()
    for b in subs: g.encode_and_submit(b)
    g.finish()
which = sys.argv[1] if len(sys.argv) > 1 else "pool"
if which == "fresh":           # encode streams created here and now, whatever the pool hands out
    D._ENC_STREAMS.clear()
elif which == "prio":          # encode streams at low priority... (torch: 0 is the lowest), exchange at high: as shipped
    pass
g1, g2 = mk(None), mk(tok2)
side = torch.cuda.Stream(dev)
def on_side(g):
    with torch.cuda.stream(side): step(g)
print(f"{nb} B in 8 waves: one encode stream (the null stream) {timed(lambda: step(g1)):.3f} ms | one (a side stream) {timed(lambda: on_side(g1)):.3f} ms | two {timed(lambda: step(g2)):.3f} ms | two, called from a side stream {timed(lambda: on_side(g2)):.3f} ms", end="")
# the encodes alone, same two forms
es = D.encode_streams(dev)
def enc1():
    for b in subs: encode_device(tok, b)
def enc2():
    for k, b in enumerate(subs):
        with torch.cuda.stream(es[k & 1]): encode_device((tok, tok2)[k & 1], b)
print(f" | encodes alone: {timed(enc1):.3f} / {timed(enc2):.3f} ms")
