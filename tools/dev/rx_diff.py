import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_device_split import _both, _blob, _first_diff
from test_host_regex import GPT2_PATTERN
from splintr_amd import Tokenizer
from fuzzgen import fuzz_corpus
t = Tokenizer.from_bytes(_blob("cl100k_base"), GPT2_PATTERN)
texts = fuzz_corpus(515, 50, 40)
st, gp, dst, dgp, status = _both(t, texts)
blob = "".join(texts).encode()
print(sys.argv[1], "status", status, "equal", np.array_equal(st, dst), np.array_equal(gp, dgp))
if not np.array_equal(st, dst):
    d = _first_diff(st, dst)
    print(" first diff at", d, repr(blob[max(0, d - 12):d + 12]), "host bits", [i for i in range(max(0,d-12), d+12) if (st[i>>5]>>(i&31))&1], "dev", [i for i in range(max(0,d-12), d+12) if (dst[i>>5]>>(i&31))&1])
