"""Dev aid: what a rank of an 8-GPU strong-scaling run does per step -- eight wave slices of ~3.4 MB encoded one after the other -- on one handle
and stream against two handles on two streams in alternation (kernel-only, HBM-resident).  usage: wave_overlap.py [c4|c5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda", 0)
if cfg == "c4":
    vocab = "llama3"; parts = [corpus.c4(15600, seed=1004 + k) for k in range(8)]          # 1 M prompts / 8 ranks / 8 waves
else:
    vocab = "deepseek_v3"
    doc = corpus.c5(2, seed=1005, doc_bytes=2 << 20)
    parts = [[d[: len(d) * 13 // 16] for d in doc] for _ in range(8)]                        # ~3.3 MB per wave
toks = [Tokenizer.from_pretrained(vocab) for _ in range(3)]
subs = [DeviceBatch(p, dev) for p in parts]
nb = sum(b.n_bytes for b in subs)
for t in toks: reserve(t, max(b.n_bytes for b in subs), max(b.n_docs for b in subs))
strs = [torch.cuda.Stream(dev) for _ in range(3)]
def run(nh):
    for k, b in enumerate(subs):
        if nh == 1: encode_device(toks[0], b)
        else:
            with torch.cuda.stream(strs[k % nh]): encode_device(toks[k % nh], b)
def timed(nh, reps=30):
    for _ in range(3): run(nh)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(nh); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2]
one = DeviceBatch([t for p in parts for t in p], dev)
tb = Tokenizer.from_pretrained(vocab); reserve(tb, one.n_bytes, one.n_docs)
def whole():
    encode_device(tb, one)
for _ in range(3): whole()
torch.cuda.synchronize(); ts = []
for _ in range(30):
    t0 = time.perf_counter(); whole(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"{cfg}: 8 waves, {nb} B ({nb // 8} per wave): ONE launch {ts[15]*1e3:.3f} ms = {nb/ts[15]/1e9:.1f} GB/s", end="")
for nh in (1, 2, 3):
    t = timed(nh); print(f" | {nh} handle(s) {t*1e3:.3f} ms = {nb/t/1e9:.1f} GB/s", end="")
print()
