"""Dev aid: kernel times of the tile-owned path against the batch size (is the 1 MB batch latency- or throughput-bound?)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
for n in (64, 125, 250, 500, 1000, 1500, 2000, 3000, 4000, 8000):
    bs = [DeviceBatch(corpus.c2(n, seed=1002 + k), torch.device("cuda", 0)) for k in range(4)]
    reserve(tok, max(b.n_bytes for b in bs), n)
    for i in range(20): encode_device(tok, bs[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): encode_device(tok, bs[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
    for i in range(40): encode_device(tok, bs[i % 4])
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)()
    L.spl_profile_read(tok.handle, ms, cnt)
    L.spl_profile_enable(tok.handle, 0)
    k = {L.spl_kernel_name(i).decode(): round(ms[i] / cnt[i] * 1e3, 2) for i in range(16) if L.spl_kernel_name(i) and cnt[i]}
    nb = bs[0].n_bytes
    print(f"docs {n:5d}  bytes {nb:8d}  tiles {(nb + 767) // 768:5d}  step {dt * 1e6:7.1f} us  {nb / dt / 1e9:6.2f} GB/s  {k}")
