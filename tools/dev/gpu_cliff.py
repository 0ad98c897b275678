"""Dev aid: cost of one pathological document (a 64 KB single-class run) inside the bench batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve
tok = Tokenizer.from_pretrained("cl100k_base")
import ctypes
from splintr_amd import _ffi
force = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_ffi.lib().spl_debug_phases(tok.handle, force << 1, (ctypes.c_uint64 * 16)())
for label, extra in (("plain", []), ("+ 64 KB of spaces", [" " * 65536 + "x"]), ("+ 64 KB of 'a'", ["a" * 65536]),
                     ("+ 64 KB of CJK", ["你" * 21845]), ("+ 300 B runs x 50", [("b" * 300 + " ") * 50]),
                     ("+ 300 B words x 50", [" ".join("".join(__import__("random").Random(k).choice("etaoinshrdlucmfw") for _ in range(300)) for k in range(50))])):
    batch = DeviceBatch(corpus.c2(1000) + extra, torch.device("cuda", 0))
    reserve(tok, batch.n_bytes, batch.n_docs)
    for _ in range(2): encode_device(tok, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): encode_device(tok, batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{label:22s} {batch.n_bytes:9d} B  {dt * 1e6:10.1f} us/step")
