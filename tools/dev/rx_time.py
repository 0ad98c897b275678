"""Dev aid: the device splitter alone (spl_split_device: text and offsets in HBM) -- us per call for a few patterns and corpora, in both
forms (option device_split_walk 1: k_rxw_walk / k_rxw_mark, 0: k_rx_match / k_rx_mark), bitmaps compared with the host splitter's.
   python tools/dev/rx_time.py [once]        (once: one call per case, for a rocprofv3 run)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from splintr_amd import Tokenizer, corpus, _ffi
from test_host_regex import GPT2_PATTERN, TIKTOKEN_CL100K, TIKTOKEN_O200K
L = _ffi.lib(); dev = torch.device("cuda", 0)
once = len(sys.argv) > 1 and sys.argv[1] == "once"
blob_v = open(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), "rb").read()
def pack(texts):
    parts = [x.encode() for x in texts]; blob = b"".join(parts)
    off = np.zeros(len(parts) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(x) for x in parts], dtype=np.uint64)
    return blob, off
cases = [("c2", corpus.c2(1000)), ("c3x400", corpus.c3(400)), ("c2x8000", corpus.c2(8000))]
if once or os.environ.get("RX_TIME_QUICK") == "1": cases = cases[:1]
WALKS = (1,)
ALL = (("gpt2", GPT2_PATTERN), ("tk_cl100k", TIKTOKEN_CL100K), ("tk_o200k", TIKTOKEN_O200K), ("p3", r" ?\p{L}+| ?[^\s\p{L}]+|\s+"), ("p2", r"\S+|\s+"))
PATS = [x for x in ALL if x[0] in os.environ.get("RX_TIME_PAT", "gpt2,tk_cl100k,tk_o200k").split(",")]
for pname, pat in PATS:
    t = Tokenizer.from_bytes(blob_v, pat)
    for cname, texts in cases:
        blob, off = pack(texts)
        words = len(blob) // 32 + 2
        st, gp = np.zeros(words, dtype=np.uint32), np.zeros(words, dtype=np.uint32)
        assert L.spl_split_host(t.handle, blob, off.ctypes.data, len(texts), st.ctypes.data, gp.ctypes.data) == 0
        d_text = torch.from_numpy(np.frombuffer(blob + b"\0" * ((-len(blob)) % 16 + 16), dtype=np.uint8).copy()).to(dev)
        d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        stream = torch.cuda.current_stream().cuda_stream
        for walk in WALKS:
            d_st = torch.full((words + 2,), -1, dtype=torch.int32, device=dev); d_gp = torch.full((words + 2,), -1, dtype=torch.int32, device=dev)
            d_status = torch.zeros(4, dtype=torch.int32, device=dev)
            def call():
                rc = L.spl_split_device(t.handle, d_text.data_ptr(), len(blob), d_off.data_ptr(), len(texts), d_st.data_ptr(), d_gp.data_ptr(), d_status.data_ptr(), stream)
                assert rc == 0, _ffi.last_error()
            call(); torch.cuda.synchronize()
            ok = np.array_equal(d_st[:words].cpu().numpy().view(np.uint32), st) and np.array_equal(d_gp[:words].cpu().numpy().view(np.uint32), gp)
            status = int(d_status[0].item()); dbg = d_status[1:4].tolist(); d_status.zero_()
            if once: print(f"{pname}/{cname} ok={ok} status={status}", flush=True); continue
            best = 1e9
            for rep in range(5):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(20): call()
                ev1.record(); torch.cuda.synchronize()
                best = min(best, ev0.elapsed_time(ev1) / 20 * 1e3)
            t0 = time.perf_counter()
            for _ in range(20): call(); torch.cuda.synchronize()
            sync_us = (time.perf_counter() - t0) / 20 * 1e6
            d = ""
            if not ok:
                a_, b_ = d_st[:words].cpu().numpy().view(np.uint32), st
                nz = np.nonzero(a_ != b_)[0]
                a2, b2 = d_gp[:words].cpu().numpy().view(np.uint32), gp
                nz2 = np.nonzero(a2 != b2)[0]
                d = f" starts differ in {len(nz)} words (first {nz[:3]}), gaps in {len(nz2)} (first {nz2[:3]})"
            print(f"{pname}/{cname} {len(blob)/1e6:.2f} MB: {best:7.1f} us/call back to back ({len(blob)/best/1e3:6.2f} GB/s), {sync_us:7.1f} us with a synchronisation per call; "
                  f"status {status} dbg {dbg} bitmaps {'==' if ok else '!='} host{d}", flush=True)
