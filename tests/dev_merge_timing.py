import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splintr_amd import _ffi
_ffi.LIB_PATH = os.path.join(ROOT, "splintr_amd", "libsplintr_timing.so")
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(corpus.c2(1000), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
for _ in range(3): encode_device(tok, batch)
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 8)()
L.spl_debug_merge_timing(out, 1)
encode_device(tok, batch); torch.cuda.synchronize()
L.spl_debug_merge_timing(out, 0)
names = ["init", "reduce", "neighbours", "shuffle", "lookup+wait", "-", "iterations", "-"]
for n, v in zip(names, out): print(f"{n:12s} {v}")
it = max(out[6], 1)
print("per iteration: reduce %.0f nbr %.0f shfl %.0f lookup %.0f" % (out[1]/it, out[2]/it, out[3]/it, out[4]/it))
