"""Dev aid: which documents of the custom-pattern test corpus make the device splitter give up (one document per call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import test_gpu_custom_pattern as T
from test_host_regex import VARIANT_B
from splintr_amd import _ffi
key = sys.argv[1] if len(sys.argv) > 1 else "variant_b"
t, orc = T._pair("cl100k_base", VARIANT_B)
texts = T._texts(100 + len(key))
L = _ffi.lib()
short = [x for x in texts if all(e - s < 1000 for s, e in orc.split(x.encode("utf-8")))]
for i, x in enumerate(short):
    b = L.spl_device_split_fallbacks(t.handle)
    t.encode_batch([x])
    if L.spl_device_split_fallbacks(t.handle) != b:
        sp = orc.split(x.encode("utf-8"))
        print(i, len(x.encode()), "longest match", max(e - s for s, e in sp), repr(x[:120]))
