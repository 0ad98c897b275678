"""GPU box: spl_encode_batch on a custom-pattern handle in a loop (the C2 batch, GPT-2's pattern, pinned input) -- a target for a kernel + copy trace."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from splintr_amd import Tokenizer, corpus, _ffi
from test_host_regex import GPT2_PATTERN
L = _ffi.lib()
tok = Tokenizer(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), GPT2_PATTERN)
bs = [t.encode("utf-8") for t in corpus.c2(1000)]
off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
blob = b"".join(bs); nb = len(blob)
p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    r = ctypes.c_void_p()
    assert L.spl_encode_batch(tok.handle, p, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0
    L.spl_result_free(r)
