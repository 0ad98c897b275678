"""GPU box: spl_split_device in a loop for ONE pattern / corpus (profiling target).  python tools/dev/rx_time1.py gpt2|tk_cl100k|tk_o200k c2|c3 [n]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from splintr_amd import Tokenizer, corpus, _ffi
from splintr_amd.device import DeviceBatch
from test_host_regex import GPT2_PATTERN, TIKTOKEN_CL100K, TIKTOKEN_O200K
pat = {"gpt2": GPT2_PATTERN, "tk_cl100k": TIKTOKEN_CL100K, "tk_o200k": TIKTOKEN_O200K}[sys.argv[1]]
texts = corpus.c2(1000) if sys.argv[2] == "c2" else corpus.c3(2000)[:400]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = _ffi.lib(); dev = torch.device("cuda", 0)
tok = Tokenizer(os.path.join(ROOT, "splintr_amd", "data", "cl100k_base.splv"), pat)
b = DeviceBatch(texts, dev)
words = b.n_bytes // 32 + 4
d_st, d_gp = torch.zeros(words, dtype=torch.int32, device=dev), torch.zeros(words, dtype=torch.int32, device=dev)
d_status = torch.zeros(4, dtype=torch.int32, device=dev)
for _ in range(n):
    assert L.spl_split_device(tok.handle, b.text.data_ptr(), b.n_bytes, b.doc_off.data_ptr(), b.n_docs, d_st.data_ptr(), d_gp.data_ptr(), d_status.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
