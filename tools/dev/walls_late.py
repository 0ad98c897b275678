"""Dev aid (-DSPL_DEBUG_STAMPS -DSPL_STAMP_ALL build): phase durations of the workgroups that start LATE in a large launch
(the steady state: every CU already holds tiles in all phases) against those of the first wave of workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(corpus.c2(8000), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
acc = None
for rep in range(8):
    encode_device(tok, batch); torch.cuda.synchronize()
    L.spl_debug_phases(tok.handle, 1, st)
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    A = np.ctypeslib.as_array(rec).reshape(2048, 8).astype(np.int64)
    A = A - A[:, 0].min()
    if rep >= 2: acc = A if acc is None else acc + A
R = acc / 6.0 / 100.0
names = ["start", "staged", "classified", "starts", "enumerated", "probe", "merge", "end"]
d = np.diff(R, axis=1)
order = np.argsort(R[:, 0])
first, late = order[:1000], order[-400:]
print("start time of the late group: p50 %.1f us" % np.percentile(R[late, 0], 50))
print("phase            first 1000 workgroups (p50)   last 400 recorded (p50)")
for i in range(7):
    print(f"  {names[i+1]:12s} {np.percentile(d[first, i], 50):8.2f} {np.percentile(d[late, i], 50):22.2f}")
print("  life         %8.2f %22.2f" % (np.percentile(R[first, 7] - R[first, 0], 50), np.percentile(R[late, 7] - R[late, 0], 50)))
