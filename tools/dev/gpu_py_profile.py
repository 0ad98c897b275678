"""Dev aid: where the Python surface spends its time (pack / C ABI / list building / deallocation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from splintr_amd import Tokenizer, corpus, _ffi
for cfg, vocab, gen, n in (("c2", "cl100k_base", corpus.c2, 1000), ("c4", "llama3", corpus.c4, 250000)):
    texts = gen(n)
    tok = Tokenizer.from_pretrained(vocab)
    sh = _ffi.shim()
    tok.encode_batch(texts[:100])
    def T(f, reps=3):
        f(); t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    ascii_frac = sum(t.isascii() for t in texts) / len(texts)
    t_pack = T(lambda: sh.pack(texts))
    t_csr = T(lambda: tok.encode_batch_csr(texts))
    res = [None]
    def full(): res[0] = tok.encode_batch(texts)
    t_full_keep = T(full)            # previous result freed inside the next call's assignment
    t0 = time.perf_counter(); res[0] = None; t_free = (time.perf_counter() - t0) * 1e3
    ntok = sum(len(x) for x in tok.encode_batch(texts))
    print(f"{cfg}: docs {n} tokens {ntok} ascii docs {ascii_frac:.2f} | pack {t_pack:.2f} ms, pack+C ABI+numpy {t_csr:.2f} ms, encode_batch {t_full_keep:.2f} ms (incl. freeing the previous result), free alone {t_free:.2f} ms")
