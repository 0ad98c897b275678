import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from splintr_amd import Tokenizer, _ffi
from splintr_amd.tokenizer import _read, _DATA, CL100K_BASE_PATTERN
L = _ffi.lib()
tok = Tokenizer.from_bytes(_read(os.path.join(_DATA, "cl100k_base.splv")), CL100K_BASE_PATTERN, {"<|x|>": 100257})
L.spl_profile_enable(tok.handle, 1)
print(tok.encode_with_special("a<|x|>b"), tok.encode("a<|x|>b"))
ms = (ctypes.c_double * 16)(); n = (ctypes.c_uint64 * 16)()
L.spl_profile_read(tok.handle, ms, n)
for i in range(16):
    nm = L.spl_kernel_name(i)
    if nm: print(nm.decode(), n[i], "%.3f ms" % ms[i])
tok2 = Tokenizer.from_pretrained("cl100k_base")
print(tok2.encode_with_special("Hello<|endoftext|>World"))
print(tok2.encode_batch_with_special(["Hello<|endoftext|>World", "x<|think|>y"]))
