"""Dev aid (round 6): what a rank of an 8-GPU strong-scaling run does per step -- its slices of the batch's waves encoded one after the other
(one handle / stream, or two in alternation) -- for 8 EQUAL waves against TAPERED plans (splintr_amd.distributed.wave_fractions),
kernel-only, HBM-resident, memo warm and off.  usage: wave_taper.py [c4|c5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
from splintr_amd.distributed import wave_fractions
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda", 0)
L = _ffi.lib()
if cfg == "c4":
    vocab = "llama3"; pool = corpus.c4(125000, seed=1004)                 # a rank's eighth of the million prompts (26.9 MB)
else:
    vocab = "deepseek_v3"
    docs = corpus.c5(13, seed=1005, doc_bytes=2 << 20)                    # ~26 MB: a rank's eighth of 100 x 2 MiB, as pieces of 128 KiB
    pool = []
    for d in docs:
        raw = d.encode(); n = len(raw); cuts = [0]
        for j in range(1, 16):
            i = raw.find(b"\n", n * j // 16)
            while i >= 0 and i + 1 < n and not (raw[i + 1:i + 2].isalnum() and raw[i + 1] < 128): i = raw.find(b"\n", i + 1)
            cuts.append(i + 1 if i >= 0 else n)
        cuts = sorted(set(cuts + [n]))
        pool += [raw[a:b].decode() for a, b in zip(cuts, cuts[1:]) if b > a]
nb_all = sum(len(t.encode()) for t in pool)
toks = [Tokenizer.from_pretrained(vocab) for _ in range(2)]
strs = [torch.cuda.Stream(dev) for _ in range(2)]
def plan(fr):
    b, acc = [0], 0.0
    for f in fr[:-1]:
        acc += f; b.append(int(round(len(pool) * acc)))
    b.append(len(pool))
    return [DeviceBatch(pool[b[k]:b[k + 1]], dev) for k in range(len(fr))]
def run(subs, nh):
    for k, b in enumerate(subs):
        if nh == 1: encode_device(toks[0], b)
        else:
            with torch.cuda.stream(strs[k % nh]): encode_device(toks[k % nh], b)
def timed(subs, nh, reps=20):
    for _ in range(3): run(subs, nh)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(subs, nh); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2]
for t in toks: reserve(t, nb_all + (1 << 20), len(pool) + 16)
print(f"{cfg}: a rank's share at N = 8: {len(pool)} items, {nb_all} B", flush=True)
for memo in (1, 0):
    for t in toks: L.spl_set_option(t.handle, b"memo", memo)
    whole = plan([1.0])
    t1 = timed(whole, 1)
    print(f"  memo={memo} ONE launch: {t1*1e3:.3f} ms {nb_all/t1/1e9:.1f} GB/s", flush=True)
    del whole
    for name, fr in (("8 equal", wave_fractions(8, 1.0)), ("7 x 0.7", wave_fractions(7, 0.7)), ("6 x 0.65", wave_fractions(6, 0.65)), ("5 x 0.6", wave_fractions(5, 0.6)), ("4 x 0.5", wave_fractions(4, 0.5))):
        subs = plan(fr)
        a, b2 = timed(subs, 1), timed(subs, 2)
        last = subs[-1].n_bytes
        print(f"  memo={memo} {name:9s}: one stream {a*1e3:.3f} ms ({nb_all/a/1e9:.1f} GB/s), two in alternation {b2*1e3:.3f} ms ({nb_all/b2/1e9:.1f} GB/s); slices MB " + " ".join(f"{s.n_bytes/1e6:.1f}" for s in subs) + f"; last wave {100*last/nb_all:.1f} % of the share", flush=True)
        del subs
