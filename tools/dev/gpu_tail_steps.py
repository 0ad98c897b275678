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
batch = DeviceBatch(getattr(corpus, gen)(ndocs), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
names = ["pack", "rows filled (3 round trips)", "boundaries", "segments <= 8 B + classification", "segments 9..16 B", "17..64 B, wait", "rows", "passes"]
tot = np.zeros(8)
prev = None
for rep in range(6):
    L.spl_debug_phases(tok.handle, 1, st)
    encode_device(tok, batch); torch.cuda.synchronize()
    rec = (ctypes.c_uint64 * (4 * 4096))()
    L.spl_debug_blocks(tok.handle, rec, 4096)
    cur = np.ctypeslib.as_array(rec).astype(np.float64)[4 * (4096 - 32): 4 * (4096 - 32) + 8].copy()
    if rep: tot += cur - prev              # (the counters only ever grow: differences between launches)
    prev = cur
tot /= 5
print(f"{vocab} {gen} x{ndocs}: {batch.n_bytes} bytes; per launch: {tot[7]:.0f} passes over {tot[6]:.0f} rows ({tot[6] / max(tot[7], 1):.0f} rows per pass)")
for k in range(6):
    print(f"  {names[k]:36s} {tot[k] / 100 / max(tot[7], 1):6.2f} us per pass")
print(f"  {'sum':36s} {tot[:6].sum() / 100 / max(tot[7], 1):6.2f} us per pass")
