"""Dev aid: phase stamps of k_pretok (debug build) on a pure-CJK / pure-JSON batch."""
import ctypes, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "cjk"
rng = random.Random(7)
texts = [getattr(corpus, kind)(rng, 4200)[:4096] for _ in range(250)]
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(texts, torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, 1, st)
for _ in range(5): encode_device(tok, batch)
torch.cuda.synchronize()
L.spl_debug_phases(tok.handle, 0, st)
names = ["stage text", "barrier", "classify", "sync flags", "chains", "enumerate", "probe", "merge", "flush"]
for i in range(8): print(f"  {names[i + 1]:12s} {st[i + 1] - st[i]:8d}")
print("  total        %8d" % (st[8] - st[0]), " medium(w0) %d short(w0) %d wait %d" % (st[9] - st[6], st[10] - st[9], st[7] - st[10]))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): encode_device(tok, batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"{kind}: {batch.n_bytes} B step {dt * 1e6:.1f} us -> {batch.n_bytes / dt / 1e6:.0f} MB/s")
