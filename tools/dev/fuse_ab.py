"""Dev aid (round 6): the fused mode (ONE launch, spl_k_fuse.h) against k_pretok + k_tile_out in ONE process, the option
toggled between blocks: parity vs the oracle, the bench rotation (us per step), a size sweep, Tokenizer.encode latency.
   python tools/dev/fuse_ab.py [label]       (SPL_LIB_PATH selects an A/B build)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
from oracle.coracle import COracle
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("SPL_LIB_PATH", "default"))
L = _ffi.lib(); dev = torch.device("cuda", 0)
def opt(tok, k, v):
    if L.spl_set_option(tok.handle, k.encode(), int(v)) != 0: raise RuntimeError(_ffi.last_error())
def packed(texts):
    bs = [t.encode() for t in texts]; off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs), dtype=np.uint8), off
def check(tok, orc, b, t):
    encode_device(tok, b); torch.cuda.synchronize()
    ids, off = result_csr(b); tn, _ = packed(t); o_ids, o_off = orc.encode_packed(tn, b.host_offsets, threads=32)
    return np.array_equal(ids, o_ids) and np.array_equal(off, o_off)
def rotation(tok, batches, n=400, reps=5):
    ts = []
    for rep in range(reps):
        for i in range(40): encode_device(tok, batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n): encode_device(tok, batches[i % len(batches)])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    quick = os.environ.get("FUSE_AB_QUICK") == "1"
    for vocab, gen, seed in (("cl100k_base", "c2", 1002), ("cl100k_base", "c2_wide", 2002), ("o200k_base", "c3", 3003))[:2 if quick else 3]:
        orc = COracle(vocab)
        if gen == "c3": sets = [corpus.c3(250, seed=seed + k) for k in range(4)]
        else: sets = [getattr(corpus, gen)(1000, seed=seed + k) for k in range(8)]
        batches = [DeviceBatch(t, dev) for t in sets]
        tok = Tokenizer.from_pretrained(vocab)
        reserve(tok, 9 << 20, 80000)
        for fuse in (1, 0, 1, 0):
            opt(tok, "fuse", fuse)
            ok = all(check(tok, orc, b, t) for b, t in zip(batches[:3], sets[:3]))
            med, best = rotation(tok, batches)
            nb = sum(b.n_bytes for b in batches) / len(batches)
            print(f"[{label}] {vocab} {gen} fuse={fuse}: {med*1e6:6.2f} us/step (best {best*1e6:6.2f})  {nb/med/1e9:6.2f} GB/s  {'ok' if ok else 'MISMATCH'}", flush=True)
    if quick: sys.exit(0)
    # size sweep: where the fused form stops paying (fuse_max_tiles lifted), English / code
    tok = Tokenizer.from_pretrained("cl100k_base"); orc = COracle("cl100k_base")
    reserve(tok, 40 << 20, 80000)
    for ndocs in (1, 2, 4, 16, 64, 250, 500, 1000, 1200):
        texts = corpus.c2(ndocs, seed=77)
        b = DeviceBatch(texts, dev)
        row = []
        for fuse in (1, 0):
            opt(tok, "fuse", fuse)
            ok = check(tok, orc, b, texts)
            med, best = rotation(tok, [b], n=200 if ndocs <= 2000 else 40, reps=3)
            row.append(f"fuse={fuse} {med*1e6:8.2f} us {b.n_bytes/med/1e9:6.2f} GB/s {'ok' if ok else 'MISMATCH'}")
        print(f"[{label}] sweep {ndocs:6d} docs {b.n_bytes:9d} B: " + " | ".join(row), flush=True)
# Tokenizer.encode latency (host path: encode_small; through the shim, which binds the in-tree library)
if os.environ.get("SPL_LIB_PATH"): sys.exit(0)
tok = Tokenizer.from_pretrained("cl100k_base")
docs = corpus.c2(64)
for fuse in (1, 0, 1, 0):
    opt(tok, "fuse", fuse)
    for text in ("Hello, world!", docs[0][:500], docs[0], (docs[0] + docs[1] + docs[2] + docs[3] + docs[4])[:4000]):
        for _ in range(300): tok.encode(text)
        ts = []
        for _ in range(2000):
            t0 = time.perf_counter(); ids = tok.encode(text); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"[{label}] encode fuse={fuse} {len(text.encode()):5d} B {len(ids):4d} tok  p10 {ts[200]*1e6:6.1f} p50 {ts[1000]*1e6:6.1f} p90 {ts[1800]*1e6:6.1f} us", flush=True)
