"""Dev aid: step time of the bench batch (and optionally C3-like batches) for the library in SPL_LIB_PATH."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
L = _ffi.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "cl100k_base"
gen = sys.argv[2] if len(sys.argv) > 2 else "c2"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
tok = Tokenizer.from_pretrained(name)
force = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # 0 auto, 1 small tiles, 2 large tiles, 3 small tiles + multi-pass
batch = DeviceBatch(getattr(corpus, gen)(n), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
if force:
    L.spl_debug_phases(tok.handle, force << 1, (ctypes.c_uint64 * 16)())
for _ in range(20):
    encode_device(tok, batch)
torch.cuda.synchronize()
ids, off = result_csr(batch)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): encode_device(tok, batch)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
for _ in range(100): encode_device(tok, batch)
torch.cuda.synchronize()
ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)()
L.spl_profile_read(tok.handle, ms, cnt); L.spl_profile_enable(tok.handle, 0)
ks = " ".join(f"{L.spl_kernel_name(i).decode()}={ms[i] / cnt[i] * 1e3:.1f}" for i in range(16) if L.spl_kernel_name(i) and cnt[i])
import zlib
print(f"{os.path.basename(os.environ.get('SPL_LIB_PATH', 'default')):18s} step {best * 1e6:7.1f} us {batch.n_bytes / best / 1e6:9.1f} MB/s  crc {zlib.crc32(ids.tobytes()):08x} | {ks}")
