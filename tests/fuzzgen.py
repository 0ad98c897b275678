"""Deterministic adversarial string generator shared by the parity tests.

The alphabet concentrates on everything the split patterns (src/core/tokenizer.rs:39, :42 in
the reference) distinguish: letter case classes (Lu/Ll/Lt/Lm/Lo), marks, three kinds of
numbers, every flavour of whitespace, CR/LF, apostrophes with (caseless) contraction letters,
punctuation, emoji/ZWJ, and multi-byte forms of each.  Nothing here was assigned after
Unicode 13, so PCRE2 (Unicode 14), Python `regex` and unicodedata agree on every class.
"""
import random
from typing import List

ATOMS = [
    # ASCII letters, incl. all contraction letters in both cases
    "a", "b", "s", "t", "r", "e", "v", "m", "l", "d", "x", "S", "T", "R", "E", "V", "M", "L", "D", "A", "Z",
    "the", "Hello", "HTTP", "camelCase", "don", "re", "ll", "ve", "LL", "Re",
    # digits and other numbers (Nd / No / Nl), incl. non-ASCII
    "0", "1", "7", "12", "123", "1234", "²", "½", "Ⅷ", "٣", "５",
    # whitespace of every kind (\s under UCP)
    " ", " ", " ", "  ", "\t", "\n", "\r", "\r\n", "\n\n", "\x0b", "\x0c", "\x85", " ", " ",
    "᠎", " ", " ", " ", " ", "　",
    # apostrophes and look-alikes
    "'", "'", "'s", "'t", "'re", "'ve", "'m", "'ll", "'d", "'S", "'RE", "'ſ", "’", "`",
    # punctuation / symbols / controls / format chars (all class "other")
    ".", ",", "!", "?", "-", "_", "(", ")", "{", "}", "[", "]", "\"", "/", "\\", "=", "==", "+", "#", "$", "%",
    ":", ";", "<", ">", "|", "<|", "|>", "//", "/\n", "\n/", "\x00", "\x1f", "\x7f", "‍", "​", "﻿", "。",
    "，", "§", "€", "—",
    # letters: Lu/Ll non-ASCII, Lt, Lm, Lo (several scripts), caseless oddities
    "É", "é", "ß", "ſ", "K", "İ", "ı", "ǅ", "ǈ", "ʰ", "ˠ",
    "々", "你", "好", "世", "界", "あ", "ア", "한", "א", "ا", "ก",
    "Α", "α", "Ж", "ж", "\U00020000", "\U0001d400",
    # marks (Mn/Mc/Me)
    "́", "̈", "ः", "⃝", "゙",
    # emoji and friends
    "\U0001f30d", "\U0001f600", "❤️", "\U0001f468‍\U0001f469",
]

WORDS = ["the", "of", "and", "to", "in", "is", "that", "for", "it", "as", "was", "with", "be", "by", "on",
         "not", "he", "this", "are", "or", "his", "from", "at", "which", "but", "have", "an", "had", "they",
         "you", "were", "their", "one", "all", "we", "can", "her", "has", "there", "been", "if", "more",
         "when", "will", "would", "who", "so", "no", "tokenizer", "wavefront", "bandwidth", "parallel",
         "International", "HTTPServer", "getElementById", "snake_case_name", "don't", "I'm", "they'll",
         "we've", "it's", "you'd", "they're", "DON'T", "x86_64", "utf8", "3.14159", "1,000,000", "2024-01-02"]


def fuzz_string(rng: random.Random, max_atoms: int = 40) -> str:
    n = rng.randint(0, max_atoms)
    mode = rng.random()
    out: List[str] = []
    for _ in range(n):
        r = rng.random()
        if mode < 0.3:  # word-ish text with occasional oddities
            if r < 0.6:
                out.append(rng.choice(WORDS))
                out.append(rng.choice([" ", " ", " ", "  ", "\n", ", ", ". ", "\t", ""]))
            else:
                out.append(rng.choice(ATOMS))
        elif mode < 0.5:  # runs: repeat the same atom several times
            out.append(rng.choice(ATOMS) * rng.randint(1, 6))
        else:
            out.append(rng.choice(ATOMS))
    return "".join(out)


def fuzz_corpus(seed: int, count: int, max_atoms: int = 40) -> List[str]:
    rng = random.Random(seed)
    return [fuzz_string(rng, max_atoms) for _ in range(count)]


def invalid_utf8_corpus(seed: int, count: int) -> List[bytes]:
    """Byte strings that are NOT valid UTF-8: fuzz text with stray continuation bytes, truncated and
    over-long sequences, lead bytes of every length (incl. 0xF8..0xFF), and runs of each."""
    rng = random.Random(seed)
    junk = [b"\x80", b"\xbf", b"\x80\x80\x80\x80\x80", b"\xc0", b"\xc1\x81", b"\xc3", b"\xe4", b"\xe4\xb8", b"\xe4\xb8\x96\x96",
            b"\xf0", b"\xf0\x9f", b"\xf0\x9f\x8c", b"\xf0\x9f\x8c\x8d\x8d", b"\xf8\x88\x80\x80\x80", b"\xff", b"\xfe\xff",
            b"\xed\xa0\x80", b"\xc0\x80", b"\xe0\x80\x80", b"\xf4\x90\x80\x80", b"\xc3\x28", b"\xa0\xa1", b"\xe2\x28\xa1"]
    out = []
    for _ in range(count):
        parts = []
        for _ in range(rng.randint(1, 12)):
            r = rng.random()
            if r < 0.45:
                parts.append(fuzz_string(rng, 6).encode("utf-8"))
            elif r < 0.85:
                parts.append(rng.choice(junk) * rng.randint(1, 3))
            else:
                parts.append(bytes(rng.randrange(256) for _ in range(rng.randint(1, 6))))
        b = b"".join(parts)
        if rng.random() < 0.3 and b:
            b = b[:rng.randrange(len(b)) + 1]                 # cut anywhere, also inside a character
        out.append(b)
    return out


_LATIN_ATOMS = [" ", " ", "  ", "\n", "\n\n", "\r\n", "\t", " \n", "\n ", "   ", "\x0b", "\x0c", "a", "b", "Hello", "world", "é", "ü", "ſ",
                "Ünï", "日本", "'", "'s", "'S", "'t", "'re", "'RE", "'ve", "'ll", "'lL", "'d", "'m", "'ſ", "'l", "'r", "'x", "1", "12", "123",
                "1234", "12345678", "!", "!!", ".", ",", "()", "{", "}\n", ";\n", "/", "//", "#", "x=1", "->", "_", "__", "-", "\"", "`",
                "~", "\x00", "\x7f", "\x1f"]


def latin_corpus(seed: int, count: int, max_atoms: int = 120) -> List[str]:
    """ASCII text with letters beyond ASCII and nothing else multi-byte: the windows the bit-vector start
    computation (spl_scan_starts.h) takes -- whitespace runs with and without newlines, contractions in
    either case and with U+017F, number runs of every length mod 3, "other" runs before letters."""
    rng = random.Random(seed)
    return ["".join(rng.choice(_LATIN_ATOMS) for _ in range(rng.randint(0, max_atoms))) for _ in range(count)]


_CASE_ATOMS = ["HelloWorld", "ABCdef", "aB", "XMLParser", "it's's", "don't", "DON'T", "we'Re", "x'rE", "'s", "a'", "É", "éÉ", "Ünï", "ǅ",
               "!\n/", "/\n/", "//\n//x", "\n/", "a/\n/b", ";\n/?", "}\n\n/", "ſ", "'ſ", "A'ſB", "1", "12", "1234567", " ", "\n", "a", "B", "'", "/", "!"]


def cased_corpus(seed: int, count: int, max_atoms: int = 80) -> List[str]:
    """What the o200k family's start masks must get right: case changes inside a letter run, contraction
    suffixes (also back to back, in either case, with U+017F), newlines and '/' behind "other" runs."""
    rng = random.Random(seed)
    return ["".join(rng.choice(_CASE_ATOMS) for _ in range(rng.randint(0, max_atoms))) for _ in range(count)]
