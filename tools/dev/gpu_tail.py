"""Dev aid (-DSPL_DEBUG_STAMPS build via SPL_LIB_PATH): which tiles end the launch?  Per-workgroup wall clock of k_pretok on one
bench batch, grouped by what the tile holds (short misses, 17..64-byte misses; from the Python oracle's split)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
from oracle import pyoracle as O
gen = sys.argv[1] if len(sys.argv) > 1 else "c2"
TB = 800
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
texts = getattr(corpus, gen)(1000)
o = O.Oracle.from_pretrained("cl100k_base")
pos = 0; short = {}; med = {}; lng = {}; mlen = {}
for t in texts:
    b = t.encode()
    for a, e in o.split(b):
        k = b[a:e]
        if k not in o.encoder and len(k) > 1:
            tl = (pos + a) // TB
            d = short if len(k) <= 16 else med if len(k) <= 64 else lng
            d[tl] = d.get(tl, 0) + 1
            if 16 < len(k) <= 64: mlen[tl] = max(mlen.get(tl, 0), len(k))
    pos += len(b)
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
acc = None
for rep in range(12):
    encode_device(tok, batch); torch.cuda.synchronize()
    L.spl_debug_phases(tok.handle, 1, st)
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    R = np.ctypeslib.as_array(rec).reshape(4096, 4).astype(np.int64)
    nt = (batch.n_bytes + TB - 1) // TB
    R = R[:nt] - int(st[14])
    if rep >= 2: acc = R if acc is None else acc + R
R = acc / 10.0 / 100.0          # us (100 MHz wall clock)
nt = len(R)
print(f"{gen}: {nt} tiles; per-tile wall clock in us since the kernel's first workgroup started (mean of 10 launches)")
for name, col in (("start", 0), ("merge done", 1), ("end", 3)):
    print(f"  {name:11s} p50 {np.percentile(R[:, col], 50):6.1f}  p90 {np.percentile(R[:, col], 90):6.1f}  p99 {np.percentile(R[:, col], 99):6.1f}  max {R[:, col].max():6.1f}")
sh = np.array([short.get(i, 0) for i in range(nt)]); md = np.array([med.get(i, 0) for i in range(nt)]); ml = np.array([mlen.get(i, 0) for i in range(nt)])
dur = R[:, 3] - R[:, 0]
print("  tile life (end - start) by content:")
for lab, sel in (("no 17..64-byte miss, <= 16 short", (md == 0) & (sh <= 16)), ("no 17..64-byte miss, > 16 short", (md == 0) & (sh > 16)),
                 ("1 medium miss", md == 1), ("2 medium misses", md == 2), (">= 3 medium misses", md >= 3)):
    if sel.any(): print(f"    {lab:36s} {sel.sum():5d} tiles  life mean {dur[sel].mean():5.1f} p90 {np.percentile(dur[sel], 90):5.1f} max {dur[sel].max():5.1f}   end mean {R[sel, 3].mean():5.1f} max {R[sel, 3].max():5.1f}")
order = np.argsort(-R[:, 3])[:15]
print("  the 15 tiles that end last: tile, start, merge done, end, short misses, medium misses (longest)")
for i in order: print(f"    {i:5d} {R[i,0]:6.1f} {R[i,1]:6.1f} {R[i,3]:6.1f}   {sh[i]:3d} {md[i]:2d} ({ml[i]})")
print("  correlation of tile life with: short misses %.2f, medium misses %.2f, longest medium %.2f, start time %.2f" % (
    np.corrcoef(dur, sh)[0, 1], np.corrcoef(dur, md)[0, 1], np.corrcoef(dur, ml)[0, 1], np.corrcoef(dur, R[:, 0])[0, 1]))
