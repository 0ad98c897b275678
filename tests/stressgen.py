"""Batch generators of the randomized parity stress (one seed -> one batch, vocabulary, execution mode).

random_batch: fuzz / Latin / cased / multi-byte text with runs of one character spliced in, documents that
end around tile and window edges, special-token literals, everything joined into one document.
edge_batch:   runs of ONE class that outgrow a tile's halo and end in multi-byte characters of the same
class at 300 consecutive alignments against the window edges, optionally with special tokens and empty
texts right behind the edge -- the family that found both window-edge bugs of round 2.
Used by tests/test_gpu_stress.py (fixed seeds, in the driver-run suite) and tools/gpu_stress.py /
gpu_edge_sweep.py (open-ended runs)."""
import json
import os
import random

from conftest import ROOT, VOCABS
from fuzzgen import cased_corpus, fuzz_corpus, latin_corpus

MODES_ALL = [0, 0, 0, 1, 4, 4, 5, 5]          # (2 and 3 were the multi-pass pipeline, removed in round 4)
_LITS = None


def literals(name):
    global _LITS
    if _LITS is None:
        with open(os.path.join(ROOT, "splintr_amd", "data", "special_tokens.json"), encoding="utf-8") as f:
            _LITS = json.load(f)
    return list(_LITS[name])


def multibyte_texts(seed, n, size):
    import test_gpu_parity as tg
    return tg._multibyte_texts(seed, n, size)


_RUN_UNITS = ["a", " ", "\n", "=", "-", "0", "é", "你", "한", "\U0001F642", "ab", " \n", "你好", "x'", "\t"]
_EDGE_SIZES = [767, 768, 769, 799, 800, 801, 863, 864, 865, 991, 992, 993, 1535, 1536, 1537, 1599, 1600, 1601, 1727, 1728, 1729]


def random_batch(seed, modes=MODES_ALL):
    """-> (vocabulary, mode, with_special, texts)"""
    rng = random.Random(seed)
    name = rng.choice(VOCABS)
    geom = rng.choice(modes)
    special = rng.random() < 0.35
    kind = rng.random()
    if kind < 0.25:
        texts = fuzz_corpus(seed, rng.randint(50, 1500), rng.choice([10, 40, 120]))
    elif kind < 0.4:
        texts = (latin_corpus(seed, rng.randint(50, 1500), rng.choice([20, 120, 600]))
                 + cased_corpus(seed, rng.randint(50, 800), rng.choice([20, 80, 400])))
    elif kind < 0.8:
        texts = multibyte_texts(seed, rng.randint(5, 120), rng.choice([150, 700, 3000, 12000]))
    else:
        texts = fuzz_corpus(seed, 300, 60) + multibyte_texts(seed + 1, 60, 2000)
    if rng.random() < 0.5:                                   # runs of one character (or a short period) of any length
        for _ in range(rng.randint(1, 6)):
            unit = rng.choice(_RUN_UNITS)
            run = unit * rng.choice([5, 40, 70, 130, 260, 520, 800, 2100, 5000])
            i = rng.randrange(len(texts))
            t = texts[i]
            c = rng.randrange(len(t) + 1)
            texts[i] = t[:c] + run + t[c:]
    if rng.random() < 0.3:                                   # documents that end around tile and window edges
        for _ in range(rng.randint(1, 20)):
            n = rng.choice(_EDGE_SIZES) + rng.randint(-2, 2)
            texts.insert(rng.randrange(len(texts) + 1), ("lorem ipsum 12 " * 200)[:n])
    if special:
        ls = literals(name)
        for i in range(0, len(texts), 2):
            t = texts[i]
            c = rng.randrange(len(t) + 1)
            texts[i] = t[:c] + rng.choice(ls) + t[c:]
    if rng.random() < 0.3:
        texts = ["".join(texts)]
    return name, geom, special, texts


# (run unit, multi-byte characters of the same class that may end the run)
FAMILIES = [("0", ["\u2167", "\u0663", "\xbd", "\U0001d7d8"]),            # numbers: Nl, Nd, No, astral Nd
            ("a", ["\xe9", "\u4f60", "\u01c5", "\U00010400"]),            # letters: Ll, Lo, Lt, astral Lu
            ("A", ["\xc9", "\u0416", "\u4f60"]),
            (" ", ["\u2003", "\xa0", "\u3000"]),                           # whitespace: EM SPACE, NBSP, IDEOGRAPHIC SPACE
            ("-", ["\u2014", "\u3002", "\U0001f642", "\xa7"]),            # "other"
            ("\n", ["\u2003", "\x85"]),                                    # newline run ending in other whitespace
            ("1a", ["\u2167\xe9"]),
            ("x'", ["'\u017f", "\u2019s"]),
            ("\u4f60", ["\u597d", "\u3002"])]
EDGE_MODES = [0, 0, 1, 4, 4, 5, 5]


def _pad(n):
    return ("lorem ipsum " * 400)[:n]


def edge_batch(seed, modes=EDGE_MODES):
    """-> (vocabulary, mode, with_special, texts)"""
    rng = random.Random(seed)
    name = rng.choice(VOCABS)
    geom = rng.choice(modes)
    unit, tails = rng.choice(FAMILIES)
    L = rng.choice([90, 120, 130, 160, 190, 200, 230, 240, 260, 400, 700, 1100])
    head = rng.choice(["", rng.choice(tails), rng.choice(tails) * 3])
    tail = "".join(rng.choice(tails) for _ in range(rng.randint(1, 3)))
    after = rng.choice([" and the end", "\nx", "", "!", "9", "Z"])
    special = rng.random() < 0.3                              # a special token right behind (or inside) the run's end
    if special:
        lit = rng.choice(literals(name))
        after = rng.choice([lit, lit + after, after + lit])
        tail = rng.choice([tail, tail + lit + tail])
    k0 = rng.randrange(0, 900)
    texts = [_pad(k) + head + unit * (L // len(unit)) + tail + after for k in range(k0, k0 + 300)]
    mix = rng.random()
    if mix < 0.3:
        texts = ["".join(texts)]
    elif mix < 0.5:
        texts = [x for t in texts for x in (t, rng.choice(["", "", "a", "é"]))]   # empty / tiny texts behind the edge
    return name, geom, special, texts


# ---- custom split patterns (host splitter + external chunk boundaries) ------------------------------------------------
CUSTOM_VOCABS = [("cl100k_base", False), ("o200k_base", False), ("llama3", False), ("mistral_v3", True), ("deepseek_v3", True)]


def custom_patterns():
    from test_host_regex import GPT2_PATTERN, MIXED, SPARSE, VARIANT_A, VARIANT_B
    return [GPT2_PATTERN, VARIANT_A, VARIANT_B, SPARSE, MIXED, r"\p{L}+(?:'\p{L}+)?|\p{N}{1,4}|\s+|.",
            r" ?[A-Za-z]+| ?[0-9]+|\s*[\r\n]+|\s+(?!\S)|\s+|[^\sA-Za-z0-9]+"]


def custom_batch(seed):
    """-> (vocabulary, byte_level, pattern, special-token map, with_special, texts, (chunk_bytes, single_chunk_max_bytes))
    Patterns that tile the text and patterns that do not (gaps), special-token sets with and without overlaps, fuzz /
    Latin / edge-run texts, one-chunk and many-chunk host pipelines."""
    rng = random.Random(seed)
    vocab, bl = rng.choice(CUSTOM_VOCABS)
    pat = rng.choice(custom_patterns())
    special = rng.random() < 0.3
    sp = {}
    if special:
        sp = {lit: 300000 + i for i, lit in enumerate(rng.sample(literals(vocab), 4))}
        if rng.random() < 0.3:
            sp.update({"<|a|>": 300100, "<|a|>x": 300101, "|>": 300102})
    kind = rng.random()
    if kind < 0.4:
        texts = fuzz_corpus(seed, rng.randint(30, 400), rng.choice([10, 40, 120]))
    elif kind < 0.6:
        texts = latin_corpus(seed, rng.randint(30, 300), 80) + cased_corpus(seed, rng.randint(30, 200), 60)
    else:
        texts = edge_batch(seed)[3][:rng.choice([40, 120, 300])]
    if special:
        ls = list(sp)
        for i in range(0, len(texts), 2):
            x = texts[i]
            c = rng.randrange(len(x) + 1)
            texts[i] = x[:c] + rng.choice(ls) + x[c:]
    if rng.random() < 0.25:
        texts = ["".join(texts)]
    if seed % 23 == 0:                                       # now and then a batch beyond 1.25 MiB: the second tile geometry
        from splintr_amd import corpus
        texts = texts + corpus.c2(1400, seed=seed) + corpus.c3(40, seed=seed)
    opts = (rng.choice([16, 64, 256]) << 10, 0) if rng.random() < 0.3 else (8 << 20, 4 << 20)
    return vocab, bl, pat, sp, special, texts, opts
