"""GPU box: time spl_split_device alone (us per call, min of rounds) for a corpus config and a pattern.  python tools/dev/rx_time.py [label]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch
from test_host_regex import GPT2_PATTERN, TIKTOKEN_CL100K, TIKTOKEN_O200K
label = sys.argv[1] if len(sys.argv) > 1 else "-"
L = _ffi.lib()
dev = torch.device("cuda", 0)
out = []
for pname, pat in (("gpt2", GPT2_PATTERN), ("tk_cl100k", TIKTOKEN_CL100K), ("tk_o200k", TIKTOKEN_O200K)):
    tok = Tokenizer(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), pat)
    for cname, texts in (("c2", corpus.c2(1000)), ("c3", corpus.c3(2000)[:400])):
        b = DeviceBatch(texts, dev)
        words = b.n_bytes // 32 + 4
        d_st, d_gp = torch.zeros(words, dtype=torch.int32, device=dev), torch.zeros(words, dtype=torch.int32, device=dev)
        d_status = torch.zeros(4, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        def f():
            assert L.spl_split_device(tok.handle, b.text.data_ptr(), b.n_bytes, b.doc_off.data_ptr(), b.n_docs, d_st.data_ptr(), d_gp.data_ptr(), d_status.data_ptr(), s) == 0
        for _ in range(3): f()
        d_status.zero_(); f()
        torch.cuda.synchronize(); cnt = d_status.cpu().tolist()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        out.append(f"{pname}/{cname} {b.n_bytes/1e6:.2f}MB {best:.0f}us {b.n_bytes/best/1e3:.2f}GB/s st={cnt}")
print(f"[{label}] " + " | ".join(out))
