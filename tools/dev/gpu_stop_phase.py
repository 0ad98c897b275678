"""Dev aid for counter passes: run the bench batch with k_pretok cut off at a phase boundary
(results are garbage by construction; only the instruction mix up to that phase is of interest)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
stop = int(sys.argv[1])
tok = Tokenizer.from_pretrained("cl100k_base")
batch = DeviceBatch(corpus.c2(1000), torch.device("cuda", 0))
reserve(tok, batch.n_bytes, batch.n_docs)
st = (ctypes.c_uint64 * 16)()
L.spl_debug_phases(tok.handle, stop << 4, st)
for _ in range(20):
    try:
        encode_device(tok, batch)
    except Exception as e:      # token-capacity errors of the garbage result are fine here
        print("encode:", e); break
torch.cuda.synchronize()
