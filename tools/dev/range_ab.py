"""Dev aid: a large device-resident batch as ONE launch pair (option range_tiles 0) against ranges of its tiles on one / two streams: the whole CSR
of every form compared with the one-pair form's, a sample of documents with the oracle, ms per step.   python tools/dev/range_ab.py [c3|c4|c5 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch, encode_device, reserve
from oracle.coracle import COracle
L = _ffi.lib(); dev = torch.device("cuda", 0)
CFG = {"c3": ("o200k_base", "c3", 10000), "c4": ("llama3", "c4", 1000000), "c5": ("deepseek_v3", "c5", 100)}
def opt(t, k, v): assert L.spl_set_option(t.handle, k.encode(), int(v)) == 0, _ffi.last_error()
with torch.cuda.stream(torch.cuda.Stream(dev)):
    for name in (sys.argv[1:] or ["c3", "c4", "c5"]):
        vocab, gen, n = CFG[name]
        texts = getattr(corpus, gen)(n)
        b = DeviceBatch(texts, dev)
        orc = COracle(vocab)
        ref = None
        for label, rt, rs, memo, gsm in (("one pair, sums added by every tile", 0, 1, 1, 0), ("one pair", 0, 1, 1, 256), ("one pair, sums added by every tile", 0, 1, 1, 0), ("one pair", 0, 1, 1, 256),
                                         ("ranges, 2 streams", 32768, 2, 1, 256), ("one pair, memo off", 0, 1, 0, 256)):
            tok = Tokenizer.from_pretrained(vocab)
            reserve(tok, b.n_bytes + (1 << 20), b.n_docs + 16)
            opt(tok, "range_tiles", rt); opt(tok, "range_streams", rs); opt(tok, "memo", memo); opt(tok, "group_scan_min", gsm)
            for _ in range(4): encode_device(tok, b)
            torch.cuda.synchronize()
            off = b.out_off.clone(); T = int(off[-1].item()); ids = b.ids[:T].clone()
            if ref is None:
                ref = (ids, off)
                k = min(2000, b.n_docs)
                want = orc.encode_batch(texts[:k])
                o = off.cpu().numpy(); i_ = ids.cpu().numpy().view(np.uint32)
                ok_o = all(i_[int(o[d]):int(o[d + 1])].tolist() == want[d] for d in range(k))
            same = torch.equal(off, ref[1]) and torch.equal(ids, ref[0])
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(5): encode_device(tok, b)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5)
            ts.sort()
            print(f"{name} {b.n_bytes/1e6:.0f} MB {label:36s}: {ts[1]*1e3:7.3f} ms {b.n_bytes/ts[1]/1e9:6.2f} GB/s  CSR == one pair's: {same}  first {k} docs == oracle: {ok_o}", flush=True)
            del tok
