"""Dev aid: latency of Tokenizer.encode(text) for texts of several sizes (median of 2000 calls), small path on / off."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from splintr_amd import Tokenizer, corpus, _ffi
L = _ffi.lib()
tok = Tokenizer.from_pretrained("cl100k_base")
docs = corpus.c2(64)
for on, solo in ((1, 1), (1, 0), (1, 1), (1, 0), (0, 0)):
    L.spl_set_option(tok.handle, b"small_path", on)
    L.spl_set_option(tok.handle, b"fuse", solo)
    for text in ("Hello, world!", docs[0][:100], docs[0][:500], docs[0], docs[0] + docs[1] + docs[2], (docs[0] + docs[1] + docs[2] + docs[3] + docs[4])[:4000]):
        for _ in range(200): tok.encode(text)
        ts = []
        for _ in range(2000):
            t0 = time.perf_counter(); ids = tok.encode(text); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"small_path={on} fuse={solo} {len(text.encode()):5d} B  {len(ids):4d} tokens  p10 {ts[200]*1e6:6.1f}  p50 {ts[1000]*1e6:6.1f}  p90 {ts[1800]*1e6:6.1f} us")
