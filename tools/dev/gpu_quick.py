"""Ad-hoc GPU bring-up script (run through gpurun); the real suite is tests/test_gpu_*.py."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from splintr_amd import Tokenizer, corpus
from oracle.coracle import COracle
from fuzzgen import fuzz_corpus

g = json.load(open(os.path.join(ROOT, "tests/golden/reference_vectors.json")))
bad = 0
for name in ("cl100k_base", "o200k_base", "llama3", "deepseek_v3"):
    t0 = time.time()
    tok = Tokenizer.from_pretrained(name)
    orc = COracle(name)
    print(name, "create %.2fs" % (time.time() - t0), tok)
    for text, ids in g[name]:
        got = tok.encode(text)
        if got != ids:
            bad += 1; print("GOLDEN MISMATCH", name, repr(text), got, ids)
    texts = fuzz_corpus(4242, 3000) + corpus.c2(50) + corpus.c3(20) + corpus.worst_case(3000)
    got = tok.encode_batch(texts)
    want = orc.encode_batch(texts, threads=8)
    nb = 0
    for i, (a, b) in enumerate(zip(got, want)):
        if a != b:
            nb += 1
            if nb <= 3:
                print("MISMATCH", name, i, repr(texts[i][:80]), a[:20], b[:20])
    print(name, "batch mismatches", nb, "of", len(texts))
    bad += nb
    gs = tok.encode_batch_with_special(["Hello<|endoftext|>World<|think|>x", "<think>a</think>", "no specials"])
    ws = [orc.encode_with_special(x) for x in ["Hello<|endoftext|>World<|think|>x", "<think>a</think>", "no specials"]]
    if gs != ws:
        bad += 1; print("SPECIAL MISMATCH", name, gs, ws)
print("TOTAL BAD", bad)
