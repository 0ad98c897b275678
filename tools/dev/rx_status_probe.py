"""Dev aid: the device splitter's status word for a few texts under one pattern."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import test_gpu_device_split as D
from test_host_regex import VARIANT_B, GPT2_PATTERN
from splintr_amd import Tokenizer
for name, pat in (("variant_b", VARIANT_B), ("gpt2", GPT2_PATTERN)):
    t = Tokenizer.from_bytes(D._blob("cl100k_base"), pat)
    for n in (10, 40, 60, 100, 200, 300, 600):
        st = D._both(t, [" " * n])[4]
        st2 = D._both(t, [" " * n + "x"])[4]
        print(name, n, "status", st, "with x behind:", st2)
