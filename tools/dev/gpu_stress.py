"""Dev aid: randomized parity stress beyond the fixed seeds of the test suite (every mode, every vocabulary)."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as tg
from fuzzgen import cased_corpus, fuzz_corpus, latin_corpus
from oracle.coracle import COracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
orcs = {}
def coracle(name):
    if name not in orcs: orcs[name] = COracle(name)
    return orcs[name]
lits = json.load(open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8"))
t0 = time.time(); runs = 0; bad = 0; seed = seed0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    name = rng.choice(tg.VOCABS); geom = rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 5]); special = rng.random() < 0.35
    kind = rng.random()
    if kind < 0.25: texts = fuzz_corpus(seed, rng.randint(50, 1500), rng.choice([10, 40, 120]))
    elif kind < 0.4: texts = latin_corpus(seed, rng.randint(50, 1500), rng.choice([20, 120, 600])) + cased_corpus(seed, rng.randint(50, 800), rng.choice([20, 80, 400]))
    elif kind < 0.8: texts = tg._multibyte_texts(seed, rng.randint(5, 120), rng.choice([150, 700, 3000, 12000]))
    else: texts = fuzz_corpus(seed, 300, 60) + tg._multibyte_texts(seed + 1, 60, 2000)
    if rng.random() < 0.5:                                   # runs of one character (or a short period) of any length
        for _ in range(rng.randint(1, 6)):
            unit = rng.choice(["a", " ", "\n", "=", "-", "0", "\u00e9", "\u4f60", "\ud55c", "\U0001F642", "ab", " \n", "\u4f60\u597d", "x'", "\t"])
            run = unit * rng.choice([5, 40, 70, 130, 260, 520, 800, 2100, 5000])
            i = rng.randrange(len(texts)); t = texts[i]; c = rng.randrange(len(t) + 1)
            texts[i] = t[:c] + run + t[c:]
    if rng.random() < 0.3:                                   # documents that end around tile and window edges
        for _ in range(rng.randint(1, 20)):
            n = rng.choice([767, 768, 769, 799, 800, 801, 863, 864, 865, 991, 992, 993, 1535, 1536, 1537, 1599, 1600, 1601, 1727, 1728, 1729]) + rng.randint(-2, 2)
            texts.insert(rng.randrange(len(texts) + 1), ("lorem ipsum 12 " * 200)[:n])
    if special:
        ls = list(lits[name])
        for i in range(0, len(texts), 2):
            t = texts[i]; c = rng.randrange(len(t) + 1); texts[i] = t[:c] + rng.choice(ls) + t[c:]
    if rng.random() < 0.3: texts = ["".join(texts)]
    tg._force_tiles(name, geom)
    try:
        tg.assert_batch_equal(name, texts, coracle, special=special)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, name, "geom", geom, "special", special, str(e)[:300], flush=True)
    finally:
        tg._force_tiles(name, 0)
    runs += 1; seed += 1
print(f"{runs} random batches, {bad} mismatches (seeds {seed0}..{seed - 1})")
