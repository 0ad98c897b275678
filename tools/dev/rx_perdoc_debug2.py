import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import numpy as np
import test_gpu_device_split as D
from test_host_regex import GPT2_PATTERN
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
t = Tokenizer.from_bytes(D._blob("o200k_base"), GPT2_PATTERN)
h = Tokenizer.from_bytes(D._blob("o200k_base"), GPT2_PATTERN)
L.spl_set_option(h.handle, b"device_split", 0)
docs = corpus.c3(20)
small = list(docs)
small[10] = small[10][:300] + "=" * 3000 + small[10][300:]
want = h.encode_batch_csr(small)
got = t.encode_batch_csr(small)
a = got[0][int(got[1][10]):int(got[1][11])]; b = want[0][int(want[1][10]):int(want[1][11])]
k = next(i for i in range(min(len(a), len(b))) if a[i] != b[i])
print("doc 10 tokens", len(a), len(b), "first diff at token", k, a[k-2:k+6], b[k-2:k+6])
print("decoded got :", repr(t.decode(a[max(k-3,0):k+4].tolist())))
print("decoded want:", repr(t.decode(b[max(k-3,0):k+4].tolist())))
# bitmaps: device (status) vs host
st, gp, dst, dgp, status = D._both(t, small)
print("split_device status", status, "start words differ:", int((st != dst).sum()), "gap words differ:", int((gp != dgp).sum()))
