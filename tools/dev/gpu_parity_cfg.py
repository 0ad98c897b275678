"""Dev aid: parity of the o200k-family configurations against the C oracle for the library in SPL_LIB_PATH (a slice of each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from splintr_amd import Tokenizer, corpus
from splintr_amd.device import DeviceBatch, encode_device, reserve, result_csr
from oracle.coracle import COracle
ok = True
for name, vocab, texts in (("c3", "o200k_base", corpus.c3(1500)), ("c5", "deepseek_v3", corpus.c5(3)), ("c4", "llama3", corpus.c4(30000)),
                           ("c3-cl100k", "cl100k_base", corpus.c3(600)), ("c3-mistral", "mistral_v3", corpus.c3(600))):
    tok = Tokenizer.from_pretrained(vocab); orc = COracle(vocab)
    b = DeviceBatch(texts, torch.device("cuda", 0)); reserve(tok, b.n_bytes, b.n_docs)
    encode_device(tok, b); torch.cuda.synchronize()
    ids, off = result_csr(b)
    bs = [t.encode() for t in texts]
    o_ids, o_off = orc.encode_packed(np.frombuffer(b"".join(bs), dtype=np.uint8), b.host_offsets, threads=64)
    good = np.array_equal(ids, o_ids) and np.array_equal(off, o_off)
    ok = ok and good
    print(sys.argv[1] if len(sys.argv) > 1 else "", name, vocab, b.n_bytes, "ok" if good else "MISMATCH")
print("ALL OK" if ok else "FAILED")
