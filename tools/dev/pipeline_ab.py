"""Dev aid: C-ABI host -> host rate of multi-chunk batches (the pipeline) under option sets, one handle per set, interleaved rounds.
usage: pipeline_ab.py c3|c4|c5 [docs|-] name:opt=v,opt=v ..."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
CONFIGS = {"c3": ("o200k_base", "c3", 10000), "c4": ("llama3", "c4", 250000), "c5": ("deepseek_v3", "c5", 25), "c2x8": ("cl100k_base", "c2", 8000), "c2": ("cl100k_base", "c2", 1000)}
cfg = sys.argv[1]
vocab, gen, n = CONFIGS[cfg]
if len(sys.argv) > 2 and sys.argv[2] != "-": n = int(sys.argv[2])
sets = []
for a in sys.argv[3:]:
    name, _, o = a.partition(":")
    sets.append((name, dict(kv.split("=") for kv in o.split(",") if kv)))
if not sets: sets = [("default", {})]
texts = getattr(corpus, gen)(n)
bs = [t.encode() for t in texts]
off = np.zeros(len(bs) + 1, dtype=np.uint64); np.cumsum([len(b) for b in bs], out=off[1:])
blob = b"".join(bs); nb = len(blob)
p = L.spl_host_alloc(nb + 64); ctypes.memmove(p, blob, nb)
_perturb = int(os.environ.get("PERTURB", "0"))     # that many live torch streams first: shifts which hardware queues the library's streams get
if _perturb:
    import torch
    _keep = [torch.cuda.Stream() for _ in range(_perturb)]
    for s_ in _keep:
        with torch.cuda.stream(s_): torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
toks = []
for name, o in sets:
    t = Tokenizer.from_pretrained(vocab)
    for k, v in o.items(): assert L.spl_set_option(t.handle, k.encode(), int(v)) == 0, _ffi.last_error()
    toks.append((name, t))
def call(t, src):
    r = ctypes.c_void_p()
    assert L.spl_encode_batch(t.handle, src, off.ctypes.data, len(bs), 0, ctypes.byref(r)) == 0, _ffi.last_error()
    nt = L.spl_result_n_tokens(r); L.spl_result_free(r); return nt
ref = None
for kind, src in (("pinned", p), ("pageable", blob)):
    res = {name: [] for name, _ in toks}
    for rep in range(4):
        for name, t in toks:
            nt = call(t, src)
            ref = nt if ref is None else ref
            assert nt == ref, (name, nt, ref)
            for _ in range(3): call(t, src)
            for _ in range(10):
                t0 = time.perf_counter(); call(t, src); res[name].append(time.perf_counter() - t0)
    print(f"{cfg} x{n} ({nb} B) {kind}: " + " | ".join(f"{k} {np.median(v)*1e3:.3f} ms = {nb/np.median(v)/1e9:.2f} GB/s" for k, v in res.items()), flush=True)
