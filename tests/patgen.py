"""Random split patterns inside the construct set the product's splitter accepts (splintr_amd/csrc/spl_regex.cpp), and small texts over an
alphabet that makes their class sets overlap -- for the parity tests of the host splitter (against PCRE2) and of the device splitter (against
the host splitter): alternations of short sequences of one-character items with greedy / lazy / possessive / counted quantifiers, groups,
caseless groups, atomic groups, look-aheads and assertions in tail position.  A pattern that can match the empty string (or uses something
the compiler refuses) is skipped by the caller: the constructor says so.
Reference: Tokenizer::new compiles whatever it is given (src/core/tokenizer.rs:410-456)."""
import random
from typing import List

ITEMS = ["a", "b", "c", " ", r"\n", "1", r"\d", r"\s", r"\S", r"\p{L}", r"\p{N}", r"\p{Lu}", r"\p{Ll}", "[ab]", r"[^a\s]", "[a-c1-3]", r"\w", ".",
         "é", r"[^\r\n\p{L}\p{N}]", r"[\p{L}\p{M}]", r"[^\s\p{L}\p{N}]", r"\p{P}", r"[\r\n]", "'", r"\x{6f22}"]
QUANTS = ["", "", "", "?", "*", "+", "+", "{1,2}", "{2}", "{1,3}", "?+", "*+", "++", "{1,3}+", "+?", "*?", "??"]
TAILS = ["", "", "", "", r"(?!\S)", r"(?=a)", r"(?!\d)", "$", r"\z", r"\b", r"(?=\s)"]


def _seq(rng: random.Random) -> str:
    n = rng.choice([1, 1, 2, 2, 3])
    out = []
    for _ in range(n):
        it = rng.choice(ITEMS)
        r = rng.random()
        if r < 0.08:
            it = "(?:" + rng.choice(ITEMS) + "|" + rng.choice(ITEMS) + rng.choice(["", "+"]) + ")"
        elif r < 0.12:
            it = "(?i:" + rng.choice(["a", "b", "s", "ab"]) + ")"
        elif r < 0.16:
            it = "(?>" + rng.choice(ITEMS) + rng.choice(["+", "*", ""]) + ")"
        out.append(it + rng.choice(QUANTS))
    return "".join(out) + rng.choice(TAILS)


def random_pattern(rng: random.Random) -> str:
    return "|".join(_seq(rng) for _ in range(rng.choice([1, 2, 2, 3, 3, 4, 5])))


ALPHABET = ["a", "a", "b", "b", "c", "A", "B", " ", " ", " ", "  ", "\n", "\r\n", "\t", "1", "2", "12", "3", "é", "漢", "字", "'", ".", ",", "-", "_", "ab", "ba",
            "abc", "́", " ", "x"]


def random_texts(rng: random.Random, count: int, max_atoms: int = 24) -> List[str]:
    return ["".join(rng.choice(ALPHABET) for _ in range(rng.randint(0, max_atoms))) for _ in range(count)]
