"""Dev aid: spl_decode_batch host -> host on the C3 batch's ids, pipeline against one piece."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
vocab, gen, n = {"c3": ("o200k_base", "c3", 10000), "c4": ("llama3", "c4", 250000), "c2x8": ("cl100k_base", "c2", 8000)}[cfg]
texts = getattr(corpus, gen)(n)
t = Tokenizer.from_pretrained(vocab)
ids, off = t.encode_batch_csr(texts)
ids = np.ascontiguousarray(ids, dtype=np.uint32); off = np.ascontiguousarray(off, dtype=np.uint64)
nbytes = sum(len(x.encode()) for x in texts)
p = L.spl_host_alloc(ids.nbytes + 64); ctypes.memmove(p, ids.ctypes.data, ids.nbytes)
sets = [("pipeline", {}), ("one piece", {"decode_chunk_ids": 1 << 40})] + [(f"chunk {c >> 10}k", {"decode_chunk_ids": c}) for c in (1 << 21, 3 << 20, 1 << 22)]
toks = []
for name, o in sets:
    tt = Tokenizer.from_pretrained(vocab)
    for k, v in o.items(): assert L.spl_set_option(tt.handle, k.encode(), int(v)) == 0
    toks.append((name, tt))
def call(tt):
    ob, oo = ctypes.POINTER(ctypes.c_uint8)(), ctypes.POINTER(ctypes.c_uint64)()
    assert L.spl_decode_batch(tt.handle, p, off.ctypes.data, len(texts), ctypes.byref(ob), ctypes.byref(oo)) == 0, _ffi.last_error()
    tot = oo[len(texts)]
    L.spl_free(ob); L.spl_free(oo)
    return tot
res = {k: [] for k, _ in toks}
for rep in range(4):
    for name, tt in toks:
        assert call(tt) == nbytes
        for _ in range(3): call(tt)
        for _ in range(10):
            t0 = time.perf_counter(); call(tt); res[name].append(time.perf_counter() - t0)
print(f"{cfg}: {len(ids)} ids -> {nbytes} B: " + " | ".join(f"{k} {np.median(v)*1e3:.3f} ms = {nbytes/np.median(v)/1e9:.2f} GB/s" for k, v in res.items()))
