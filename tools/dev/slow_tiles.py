"""Dev aid (-DSPL_DEBUG_STAMPS -DSPL_STAMP_ALL build via SPL_LIB_PATH): which tiles of the bench batch are slow in which phase, and what their text holds."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
TB = 800
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
texts = corpus.c2(1000)
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
for _ in range(6): encode_device(tok, batch)
torch.cuda.synchronize()
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
nt = (batch.n_bytes + TB - 1) // TB
acc = None
for rep in range(12):
    encode_device(tok, batch); torch.cuda.synchronize()
    L.spl_debug_phases(tok.handle, 1, st)
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    A = np.ctypeslib.as_array(rec).reshape(2048, 8).astype(np.int64)[:nt]
    A = A - A[:, 0].min()
    if rep >= 2: acc = A if acc is None else acc + A
R = acc / 10.0 / 100.0
D = np.diff(R, axis=1)          # staged, classified, masks, enumerated, probe, merge, end
blob = b"".join(t.encode() for t in texts)
offs = np.cumsum([0] + [len(t.encode()) for t in texts])
names = ["stage", "classify", "masks", "enum", "probe", "merge", "end"]
feat = []
for t in range(nt):
    w = blob[max(0, t * TB - 32): t * TB + 992]
    hi = sum(1 for c in w if c >= 0x80)
    nd = int(((offs >= t * TB) & (offs < (t + 1) * TB)).sum())
    feat.append((hi, nd))
feat = np.array(feat)
print("correlation of a tile's phase duration with (bytes >= 0x80 in its window, documents that start in it):")
for k, nm in enumerate(names):
    print(f"  {nm:9s} p50 {np.percentile(D[:,k],50):5.1f} p90 {np.percentile(D[:,k],90):5.1f} max {D[:,k].max():5.1f}   corr hi {np.corrcoef(D[:,k], feat[:,0])[0,1]:+.2f}  docs {np.corrcoef(D[:,k], feat[:,1])[0,1]:+.2f}  tile index {np.corrcoef(D[:,k], np.arange(nt))[0,1]:+.2f}")
md = R[:, 6]
print(f"merge done: corr hi {np.corrcoef(md, feat[:,0])[0,1]:+.2f} docs {np.corrcoef(md, feat[:,1])[0,1]:+.2f} index {np.corrcoef(md, np.arange(nt))[0,1]:+.2f} start {np.corrcoef(md, R[:,0])[0,1]:+.2f}")
last = np.argsort(-md)[:15]
print("the 15 tiles that finish their merges last: tile, start, phase durations, hi bytes, docs")
for t in last: print(f"  {t:5d} start {R[t,0]:4.1f} " + " ".join(f"{x:5.1f}" for x in D[t,:6]) + f"  merge done {md[t]:5.1f}  hi {feat[t,0]:3d} docs {feat[t,1]}")
print("tiles with NO byte >= 0x80: classify p50 %.1f max %.1f (%d tiles); with some: p50 %.1f max %.1f" % (np.percentile(D[feat[:,0]==0,1],50), D[feat[:,0]==0,1].max(), (feat[:,0]==0).sum(), np.percentile(D[feat[:,0]>0,1],50), D[feat[:,0]>0,1].max()))
