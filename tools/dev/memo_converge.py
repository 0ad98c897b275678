import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib(); L.spl_memo_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
dev = torch.device("cuda", 0)
for vocab, gen, n in (("deepseek_v3", "c5", 100), ("llama3", "c4", 1000000)):
    b = DeviceBatch(getattr(corpus, gen)(n), dev)
    tok = Tokenizer.from_pretrained(vocab); reserve(tok, b.n_bytes + (1 << 20), b.n_docs + 16)
    hist = []
    for k in range(14):
        encode_device(tok, b); torch.cuda.synchronize()
        o = (ctypes.c_uint64 * 4)(); L.spl_memo_stats(tok.handle, o); hist.append((int(o[0]), int(o[1])))
    print(gen, "fills / chunks put in after each pass:", hist, flush=True)
    del tok, b
