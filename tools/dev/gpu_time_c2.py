"""Dev aid: per-kernel times of the bench batch for the library in place, WITHOUT verification (timing experiments)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
bs = [DeviceBatch(corpus.c2(1000, seed=1002 + k), torch.device("cuda", 0)) for k in range(8)]
reserve(tok, max(b.n_bytes for b in bs), 1000)
for i in range(40): encode_device(tok, bs[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(400): encode_device(tok, bs[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
for i in range(80): encode_device(tok, bs[i % 8])
torch.cuda.synchronize()
ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)()
L.spl_profile_read(tok.handle, ms, cnt)
print(sys.argv[1] if len(sys.argv) > 1 else "", f"step {dt * 1e6:.1f} us", {L.spl_kernel_name(i).decode(): round(ms[i] / cnt[i] * 1e3, 2) for i in range(16) if L.spl_kernel_name(i) and cnt[i]})
