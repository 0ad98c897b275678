"""Dev aid (-DSPL_DEBUG_STAMPS -DSPL_STAMP_ALL build via SPL_LIB_PATH): every workgroup's wall clock at k_pretok's phase
boundaries on one bench batch -- where a typical tile's life goes with all tiles resident, and where the last ones lose."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
gen = sys.argv[1] if len(sys.argv) > 1 else "c2"
ndocs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TB = 800
PASS = int(os.environ.get("SPL_WALLS_PASS", "0"))
L = _ffi.lib()
tok = Tokenizer.from_pretrained(os.environ.get("SPL_WALLS_VOCAB", "cl100k_base"))
if gen == "purecjk":
    import random
    _r = random.Random(7)
    texts = [corpus.cjk(_r, 4000) for _ in range(ndocs)]
else:
    texts = getattr(corpus, gen)(ndocs)
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
for k_, v_ in (("memo", "SPL_WALLS_MEMO"), ("fuse", "SPL_WALLS_FUSE")):
    if os.environ.get(v_) is not None: L.spl_set_option(tok.handle, k_.encode(), int(os.environ[v_]))
for _ in range(6): encode_device(tok, batch)          # (the chunk memo fills)
torch.cuda.synchronize()
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
nt = min((batch.n_bytes + TB - 1) // TB, 2048)
acc = None
for rep in range(12):
    encode_device(tok, batch); torch.cuda.synchronize()
    L.spl_debug_phases(tok.handle, 1, st)
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    A = np.ctypeslib.as_array(rec).reshape(2048, 8).astype(np.int64)
    R = A[:nt] if not PASS else A[1024:1024 + nt]      # (SPL_PASSES builds: the second pass's records sit 1024 workgroups on)
    R = R - (R[:, 0].min() if not PASS else A[:nt, 0].min())
    if rep >= 2: acc = R if acc is None else acc + R
R = acc / 10.0 / 100.0          # us
names = ["start", "staged (barrier)", "classified", "masks + starts", "enumerated", "probe done", "merge done", "end"]
print(f"{gen} x{ndocs}{' (second pass of every workgroup)' if PASS else ''}: {nt} tiles; wall clock in us since the first workgroup started (mean of 10 launches)")
print(f"  {'boundary':18s} {'p50':>6s} {'p90':>6s} {'max':>6s}    phase duration p50 / p90 / max")
for i, nm in enumerate(names):
    c = R[:, i]
    line = f"  {nm:18s} {np.percentile(c,50):6.1f} {np.percentile(c,90):6.1f} {c.max():6.1f}"
    if i:
        d = R[:, i] - R[:, i - 1]
        line += f"    {np.percentile(d,50):5.1f} / {np.percentile(d,90):5.1f} / {d.max():5.1f}"
    print(line)
last = np.argsort(-R[:, 7])[:20]
d = np.diff(R, axis=1)
print("  the 20 tiles that end last, phase durations (mean):", " ".join(f"{x:5.1f}" for x in d[last].mean(axis=0)), " start", f"{R[last,0].mean():.1f}")
print("  all tiles, phase durations (mean):                 ", " ".join(f"{x:5.1f}" for x in d.mean(axis=0)))
