"""Dev aid: kernel-only rate of the large configurations for one library build (SPL_LIB_PATH) -- the timing-only cut builds of the tile-owned
tail (-DSPL_TAIL_CUT=1..3, -DSPL_SKIP_TAIL=1: tokens missing, nothing is compared) beside the shipped one; chunk memo on and off.
   python tools/dev/tail_cuts.py <label>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
label = sys.argv[1] if len(sys.argv) > 1 else "default"
L = _ffi.lib(); dev = torch.device("cuda", 0)
out = []
with torch.cuda.stream(torch.cuda.Stream(dev)):
    for name, vocab, gen, seed, ndocs in (("c3", "o200k_base", "c3", 7003, 2500), ("c4", "llama3", "c4", 8004, 100000), ("c5", "deepseek_v3", "c5", 9005, 8), ("c2x8", "cl100k_base", "c2", 5002, 8000)):
        b = DeviceBatch(getattr(corpus, gen)(ndocs, seed=seed), dev)
        for memo in (1, 0):
            tok = Tokenizer.from_pretrained(vocab)
            reserve(tok, b.n_bytes + (1 << 20), 200000)
            assert L.spl_set_option(tok.handle, b"memo", memo) == 0
            for _ in range(8): encode_device(tok, b)
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(20): encode_device(tok, b)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
            ts.sort()
            out.append(f"{name}{'' if memo else '-'} {b.n_bytes / ts[1] / 1e9:.2f}")
            del tok
print(f"{label:7s} " + " | ".join(out), flush=True)
