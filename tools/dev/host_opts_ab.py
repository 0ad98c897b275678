"""Dev aid: C-ABI host -> host rate of one-chunk batches with the direct_read / spin_done options (VERDICT r04 #5b), same process, interleaved."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
for vocab, gen, n in (("cl100k_base", "c2", 1000), ("cl100k_base", "c2", 3000), ("o200k_base", "c3", 800)):
    texts = getattr(corpus, gen)(n)
    bs = [t.encode() for t in texts]
    off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
    blob = b"".join(bs); nb = len(blob)
    p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
    toks = {}
    for name, opts in (("default", {}), ("no_direct_read", {"direct_read": 0})):
        t = Tokenizer.from_pretrained(vocab)
        for k, v in opts.items(): assert L.spl_set_option(t.handle, k.encode(), v) == 0
        toks[name] = t
    def call(t):
        r = ctypes.c_void_p()
        assert L.spl_encode_batch(t.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0, _ffi.last_error()
        nt = L.spl_result_n_tokens(r); L.spl_result_free(r); return nt
    ref = call(toks["default"])
    res = {k: [] for k in toks}
    for rep in range(5):
        for name, t in toks.items():
            assert call(t) == ref
            for _ in range(20): call(t)
            t0 = time.perf_counter()
            for _ in range(200): call(t)
            res[name].append((time.perf_counter() - t0) / 200)
    print(f"{gen} x{n} ({nb} B): " + " | ".join(f"{k} {sorted(v)[2]*1e6:.1f} us = {nb/sorted(v)[2]/1e9:.2f} GB/s" for k, v in res.items()))
