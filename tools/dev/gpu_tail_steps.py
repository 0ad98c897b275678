"""Dev aid (-DSPL_DEBUG_STAMPS -DSPL_STAMP_TAIL build via SPL_LIB_PATH): where the passes of the tile-owned tail
(bpe_tail_segments: multi-byte text, chunks beyond 64 bytes) spend their time -- wall clock of thread 0 between the steps of
a pass, summed over all workgroups and passes of one launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
vocab = sys.argv[1] if len(sys.argv) > 1 else "o200k_base"
gen = sys.argv[2] if len(sys.argv) > 2 else "c3"
ndocs = int(sys.argv[3]) if len(sys.argv) > 3 else 250
L = _ffi.lib()
tok = Tokenizer.from_pretrained(vocab)
if gen == "purecjk":
    import random
    _r = random.Random(7)
    _texts = [corpus.cjk(_r, 4000) for _ in range(ndocs)]
else:
    _texts = getattr(corpus, gen)(ndocs)
batch = DeviceBatch(_texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
names = ["pack (+ wait for stragglers)", "step 1: heads, light rows", "step 2: heavy rows tabulated", "boundaries", "classification of segments",
         "segments <= 8 B", "segments 9..16 B", "17..64 B, wait"]
NW = 24
tot = np.zeros(NW)
prev = None
for rep in range(6):
    L.spl_debug_phases(tok.handle, 1, st)
    encode_device(tok, batch); torch.cuda.synchronize()
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    cur = np.ctypeslib.as_array(rec).astype(np.float64)[4 * (4096 - 32): 4 * (4096 - 32) + NW].copy()
    if rep: tot += cur - prev              # (the counters only ever grow: differences between launches)
    prev = cur
tot /= 5
import time
torch.cuda.synchronize(); _t0 = time.perf_counter()
for _ in range(20): encode_device(tok, batch)
torch.cuda.synchronize(); print(f"  ({(time.perf_counter() - _t0) / 20 * 1e6:.0f} us per launch of this (stamps) build)")
np_ = max(tot[12], 1)
print(f"{vocab} {gen} x{ndocs}: {batch.n_bytes} bytes; per launch: {tot[12]:.0f} passes over {tot[13]:.0f} rows ({tot[13] / np_:.0f} rows per pass, "
      f"{tot[14] / np_:.0f} heavy, {tot[15] / np_:.0f} short segments, {tot[16] / np_:.1f} of 9..16 B, {tot[17] / np_:.1f} of 17..64 B)")
for k in range(8):
    print(f"  {names[k]:36s} {tot[k] / 100 / np_:6.2f} us per pass")
print(f"  {'sum':36s} {tot[:8].sum() / 100 / np_:6.2f} us per pass")
