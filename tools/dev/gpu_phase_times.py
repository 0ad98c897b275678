"""Dev aid (-DSPL_DEBUG_STAMPS builds): duration of k_pretok when it is cut off after each phase -- the
cumulative cost of the phases under real concurrency (all workgroups resident, 5 per CU)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
bs = [DeviceBatch(corpus.c2(1000, seed=1002 + k), torch.device("cuda", 0)) for k in range(8)]
reserve(tok, max(b.n_bytes for b in bs), 1000)
names = {1: "stage + document search", 2: "+ classify", 3: "+ masks, sync", 4: "+ chains", 5: "+ enumerate", 6: "+ whole-chunk probe", 7: "+ merge loops", 0: "+ tile record (whole kernel)"}
st = (ctypes.c_uint64 * 16)()
for stop in (1, 2, 3, 4, 5, 6, 7, 0):
    L.spl_debug_phases(tok.handle, stop << 4, st)
    for i in range(16): encode_device(tok, bs[i % 8])
    torch.cuda.synchronize()
    L.spl_profile_enable(tok.handle, 1); L.spl_profile_reset(tok.handle)
    for i in range(64): encode_device(tok, bs[i % 8])
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 16)(); cnt = (ctypes.c_uint64 * 16)()
    L.spl_profile_read(tok.handle, ms, cnt); L.spl_profile_enable(tok.handle, 0)
    print(f"cut after phase {stop}: k_pretok {ms[2] / cnt[2] * 1e3:7.2f} us   {names[stop]}")
