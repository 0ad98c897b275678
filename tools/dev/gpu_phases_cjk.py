"""Dev aid: phase stamps of k_pretok (debug build) on a pure-CJK / pure-JSON batch."""
import ctypes, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "cjk"
rng = random.Random(7)
texts = [getattr(corpus, kind)(rng, 4200)[:4096] for _ in range(250)]
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
for _ in range(5): encode_device(tok, batch)
torch.cuda.synchronize()
L.spl_debug_phases(tok.handle, 0, st)
names = ["stage text", "barrier", "classify", "sync flags", "chains", "enumerate", "probe", "merge", "flush"]
for i in range(8): print(f"  {names[i + 1]:12s} {st[i + 1] - st[i]:8d}")
print("  total        %8d" % (st[8] - st[0]), " medium(w0) %d short(w0) %d wait %d" % (st[9] - st[6], st[10] - st[9], st[7] - st[10]))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): encode_device(tok, batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"{kind}: {batch.n_bytes} B step {dt * 1e6:.1f} us -> {batch.n_bytes / dt / 1e6:.0f} MB/s")
import numpy as np
nb_ = 4096
rec = (ctypes.c_uint64 * (4 * nb_))()
L.spl_debug_blocks(tok.handle, rec, nb_)
R = np.ctypeslib.as_array(rec).reshape(nb_, 4).astype(np.int64)
nblk = (batch.n_bytes + 767) // 768
R = R[:min(nblk, nb_)]
rel = R - R[:, 0].min()
for col, nm in ((1, "merge done"), (2, "counts done"), (3, "end")):
    print("  %-12s p50 %6d p90 %6d p99 %6d max %6d (10 ns ticks)" % ((nm,) + tuple(np.percentile(rel[:, col], [50, 90, 99, 100]))))
dur_tail = rel[:, 2] - rel[:, 1]
worst = np.argsort(-dur_tail)[:5]
blob = "".join(texts).encode()
for w in worst:
    seg = blob[w * 768:(w + 1) * 768 + 224].decode("utf-8", "ignore")
    import regex
    runs = sorted((len(m.group().encode()) for m in regex.finditer(r"\p{L}+", seg)), reverse=True)[:4]
    print("  wg %d: merge done %d, tail %d ticks; longest letter runs in its window (bytes): %s" % (w, rel[w, 1], dur_tail[w], runs))
