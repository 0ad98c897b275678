"""Dev aid: open-ended randomized parity stress of the custom-pattern path (host splitter + external chunk boundaries in the tile
kernel) against the Python oracle running the same pattern on PCRE2.   python tools/dev/gpu_custom_stress.py [seconds] [first seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
from fuzzgen import cased_corpus, fuzz_corpus, latin_corpus
from stressgen import edge_batch, literals
from test_host_regex import GPT2_PATTERN, MIXED, SPARSE, VARIANT_A, VARIANT_B
from splintr_amd import Tokenizer, _ffi
from oracle import pyoracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DATA = os.path.join(ROOT, "splintr_amd", "data")
PATS = [GPT2_PATTERN, VARIANT_A, VARIANT_B, SPARSE, MIXED, r"\p{L}+(?:'\p{L}+)?|\p{N}{1,4}|\s+|.", r" ?[A-Za-z]+| ?[0-9]+|\s*[\r\n]+|\s+(?!\S)|\s+|[^\sA-Za-z0-9]+"]
VOC = [("cl100k_base", False), ("o200k_base", False), ("llama3", False), ("mistral_v3", True), ("deepseek_v3", True)]
cache = {}
def pair(vocab, bl, pat, sp):
    key = (vocab, pat, tuple(sorted(sp.items())))
    if key not in cache:
        if len(cache) > 12: cache.clear()
        with open(os.path.join(DATA, vocab + ".splv"), "rb") as f: blob = f.read()
        enc, _ = O.load_splv(os.path.join(DATA, vocab + ".splv"))
        t = (Tokenizer.from_bytes_byte_level if bl else Tokenizer.from_bytes)(blob, pat, sp)
        cache[key] = (t, O.Oracle(enc, pat, bl, sp, "pcre2"))
    return cache[key]
t0 = time.time(); runs = 0; bad = 0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    vocab, bl = rng.choice(VOC); pat = rng.choice(PATS)
    special = rng.random() < 0.3
    sp = {}
    if special:
        lits = literals(vocab)
        sp = {l: 300000 + i for i, l in enumerate(rng.sample(lits, 4))}
        if rng.random() < 0.3: sp["<|a|>"] = 300100; sp["<|a|>x"] = 300101; sp["|>"] = 300102
    kind = rng.random()
    if kind < 0.4: texts = fuzz_corpus(seed, rng.randint(30, 400), rng.choice([10, 40, 120]))
    elif kind < 0.6: texts = latin_corpus(seed, rng.randint(30, 300), 80) + cased_corpus(seed, rng.randint(30, 200), 60)
    else: texts = edge_batch(seed)[3][:rng.choice([40, 120, 300])]
    if special:
        ls = list(sp)
        for i in range(0, len(texts), 2):
            x = texts[i]; c = rng.randrange(len(x) + 1); texts[i] = x[:c] + rng.choice(ls) + x[c:]
    if rng.random() < 0.25: texts = ["".join(texts)]
    t, orc = pair(vocab, bl, pat, sp)
    L = _ffi.lib()
    if rng.random() < 0.3:
        L.spl_set_option(t.handle, b"chunk_bytes", rng.choice([16, 64, 256]) << 10); L.spl_set_option(t.handle, b"single_chunk_max_bytes", 0)
    else:
        L.spl_set_option(t.handle, b"chunk_bytes", 8 << 20); L.spl_set_option(t.handle, b"single_chunk_max_bytes", 4 << 20)
    got = t.encode_batch_with_special(texts) if special else t.encode_batch(texts)
    for i, x in enumerate(texts):
        want = orc.encode_with_special(x) if special else orc.encode(x)
        if got[i] != want:
            bad += 1; print("MISMATCH seed", seed, vocab, "special", special, ascii(pat[:40]), ascii(x[:80]), got[i][:12], want[:12], flush=True); break
    runs += 1; seed += 1
print(f"{runs} custom-pattern batches, {bad} mismatches")
