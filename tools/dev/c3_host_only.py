"""Dev aid: C-ABI host -> host rate of the C3 batch (multi-chunk pipeline), pinned input."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
texts = corpus.c3(10000)
bs = [t.encode() for t in texts]
off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
blob = b"".join(bs); nb = len(blob)
p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
t = Tokenizer.from_pretrained("o200k_base")
for k, v in [kv.split("=") for kv in sys.argv[1:]]:
    assert L.spl_set_option(t.handle, k.encode(), int(v)) == 0
def call():
    r = ctypes.c_void_p()
    assert L.spl_encode_batch(t.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0, _ffi.last_error()
    L.spl_result_free(r)
for _ in range(5): call()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"C3 {nb} B: median {ts[15]*1e3:.3f} ms = {nb/ts[15]/1e9:.2f} GB/s (best {nb/ts[0]/1e9:.2f})  env: " + " ".join(f"{k}={os.environ[k]}" for k in ("HSA_ENABLE_SDMA","GPU_FORCE_BLIT_COPY_SIZE","ROC_GLOBAL_CU_MASK") if k in os.environ))
