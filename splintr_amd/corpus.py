"""Deterministic synthetic corpora for BASELINE.json's configs.

Self-contained (stdlib `random.Random(seed)` only), no identical documents, natural Zipf-like
word reuse.  Nothing here uses code points assigned after Unicode 13, so every Unicode table in
play (PCRE2/Unicode 14, Python `regex`, the class table) agrees on all of them.

    c1(n)  cl100k   English prose, ~1 KB docs                    seed 1001
    c2(n)  cl100k   50 % prose / 50 % code, ~1 KB docs           seed 1002   (the bench workload)
    c3(n)  o200k    40 % prose / 30 % JSON / 30 % CJK, ~4 KB     seed 1003
    c4(n)  llama3   short chat prompts, 64-512 B log-uniform     seed 1004
    c5(n)  deepseek long documents (default 2 MiB) of mixed paragraphs   seed 1005
    worst_case()    long single-class runs and emoji ZWJ sequences
"""
from __future__ import annotations

import math
import random
from typing import List

_COMMON = ("the of and to in is that for it as was with be by on not he this are or his from at which but have an "
           "had they you were their one all we can her has there been if more when will would who so no out up "
           "into than them only some could these two may then do first any my now such like our over man me even "
           "most made after also did many before must through back years where much your way well down should "
           "because each just those people how too little state good very make world still own see men work long "
           "get here between both life being under never day same another know while last might us great old "
           "year off come since against go came right used take three").split()
_RARE = ("tokenizer wavefront bandwidth parallel throughput latency compiler accelerator heterogeneous "
         "quantization vectorized asynchronous infrastructure characterization internationalization "
         "photosynthesis thermodynamics electromagnetic pharmaceutical jurisdiction entrepreneurship "
         "Mississippi Kubernetes PostgreSQL JavaScript TensorFlow Wikipedia Massachusetts Schwarzenegger "
         "counterintuitive misunderstanding unconstitutional disproportionately xylophone zeitgeist "
         "naïve café résumé Zürich São Paulo señor jalapeño Ångström façade coöperate").split()
_CONTR = ["don't", "I'm", "they'll", "we've", "it's", "you'd", "they're", "can't", "won't", "she's", "I'll",
          "DON'T", "We'Re", "o'clock", "rock'n'roll"]
_PUNCT = [". ", ", ", "; ", ": ", "! ", "? ", " - ", " (", ") ", "\"", "... ", ".\n", ".\n\n"]
_IDENT = ("index value result buffer count offset length table entry token chunk rank merge pair state "
          "config options handler request response stream kernel device host thread block grid tile "
          "getValue setName parseInput encodeBatch HTTPServer XMLParser userId maxLen numTokens "
          "snake_case_name load_table byte_pair_encode __init__ self cls args kwargs").split()
_KEYW = "def return if else elif for while in not and or import from class try except with as lambda None True False".split()
_CKEYW = "int void char const static struct return if else for while unsigned size_t uint32_t".split()
_CJK_COMMON = "的一是不了人我在有他这为之大来以个中上们到说国和地也子时道出而要于就下得可你年生自会那后能对着事其里所去行过家十用发天如然作方成者多日都三小军二无同么经法当起与好看学进种将还分此心前面又定见只主没公从"
_KANA = "あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんアイウエオカキクケコサシスセソタチツテトナニヌネノ"
_HANGUL = "가나다라마바사아자차카타파하한국어서울대학교사람시간문제생각"
_FW_PUNCT = "，。！？；：「」（）、"
_EMOJI = ["\U0001f600", "\U0001f30d", "\U0001f680", "❤️", "\U0001f468‍\U0001f469‍\U0001f467", "\U0001f44d\U0001f3fd"]


def _word(rng: random.Random) -> str:
    r = rng.random()
    if r < 0.80:
        # Zipf-ish: low indices far more likely
        return _COMMON[min(int(rng.paretovariate(1.1)) - 1, len(_COMMON) - 1)]
    if r < 0.92:
        return rng.choice(_RARE)
    if r < 0.96:
        return rng.choice(_CONTR)
    if r < 0.98:
        return str(rng.randint(0, 10 ** rng.randint(1, 7)))
    return rng.choice(_COMMON).capitalize()


def prose(rng: random.Random, nbytes: int) -> str:
    out: List[str] = []
    size = 0
    cap = True
    while size < nbytes:
        w = _word(rng)
        if cap:
            w = w[:1].upper() + w[1:]
            cap = False
        sep = " "
        if rng.random() < 0.14:
            sep = rng.choice(_PUNCT)
            cap = sep[0] in ".!?"
        out.append(w)
        out.append(sep)
        size += len(w) + len(sep)
    return "".join(out)


def code(rng: random.Random, nbytes: int) -> str:
    out: List[str] = []
    size = 0
    indent = 0
    style = rng.choice(["py", "c", "json"])
    while size < nbytes:
        ind = ("\t" * indent) if rng.random() < 0.2 else ("    " * indent)
        a, b, c = rng.choice(_IDENT), rng.choice(_IDENT), rng.choice(_IDENT)
        k = rng.random()
        if style == "py":
            if k < 0.2:
                line = f"{ind}def {a}({b}, {c}=None):"
                indent = min(indent + 1, 4)
            elif k < 0.4:
                line = f"{ind}{a} = {b}[{rng.randint(0, 4096)}] + {c}.{rng.choice(_IDENT)}({rng.random():.4f})"
            elif k < 0.55:
                line = f"{ind}{rng.choice(_KEYW)} {a} {rng.choice(['==', '!=', '<=', 'in', 'is not'])} {b}:"
                indent = min(indent + 1, 4)
            elif k < 0.7:
                line = f"{ind}return {a} if {b} else '{c}_{rng.randint(0, 99)}'  # {prose(rng, 20).strip()}"
                indent = max(indent - 1, 0)
            elif k < 0.8:
                line = f'{ind}print(f"{{{a}}}: {{{b}:>8.3f}}\\n")'
            else:
                line = ""
                indent = max(indent - 1, 0)
        elif style == "c":
            if k < 0.2:
                line = f"{ind}{rng.choice(_CKEYW)} {a}({rng.choice(_CKEYW)} *{b}, size_t {c}) {{"
                indent = min(indent + 1, 4)
            elif k < 0.5:
                line = f"{ind}{a}[{b}++] = ({rng.choice(_CKEYW)})({c} >> {rng.randint(1, 31)}) & 0x{rng.randint(0, 2 ** 32 - 1):08X};"
            elif k < 0.65:
                line = f"{ind}for (int {a} = 0; {a} < {b}; ++{a}) {{"
                indent = min(indent + 1, 4)
            elif k < 0.8:
                line = f"{ind}}}"
                indent = max(indent - 1, 0)
            else:
                line = f"{ind}/* {prose(rng, 30).strip()} */"
        else:
            if k < 0.5:
                line = f'{ind}"{a}": {rng.choice([str(rng.randint(-999, 99999)), f"{rng.random() * 1000:.3f}", "true", "null", chr(34) + b + chr(34)])},'
            elif k < 0.7:
                line = f'{ind}"{a}_{b}": {{'
                indent = min(indent + 1, 4)
            elif k < 0.85:
                line = f'{ind}"{a}": [{", ".join(str(rng.randint(0, 255)) for _ in range(rng.randint(1, 8)))}],'
            else:
                line = f"{ind}}},"
                indent = max(indent - 1, 0)
        out.append(line + "\n")
        size += len(line) + 1
    return "".join(out)


def json_doc(rng: random.Random, nbytes: int) -> str:
    out: List[str] = ["{"]
    size = 1
    while size < nbytes:
        k = rng.choice(_IDENT)
        r = rng.random()
        if r < 0.3:
            v = str(rng.randint(-10 ** 6, 10 ** 9))
        elif r < 0.5:
            v = f"{rng.uniform(-1e4, 1e4):.6f}"
        elif r < 0.8:
            v = '"' + prose(rng, rng.randint(5, 60)).strip().replace('"', '\\"') + '\\n"'
        elif r < 0.9:
            v = "[" + ", ".join(str(rng.randint(0, 999)) for _ in range(rng.randint(0, 10))) + "]"
        else:
            v = '{"' + rng.choice(_IDENT) + '": ' + rng.choice(["true", "false", "null"]) + "}"
        s = f'"{k}": {v}, '
        out.append(s)
        size += len(s)
    out.append('"end": 0}')
    return "".join(out)


def cjk(rng: random.Random, nbytes: int) -> str:
    out: List[str] = []
    size = 0
    while size < nbytes:
        r = rng.random()
        if r < 0.7:
            s = "".join(rng.choice(_CJK_COMMON) for _ in range(rng.randint(2, 18)))
        elif r < 0.85:
            s = "".join(rng.choice(_KANA) for _ in range(rng.randint(2, 12)))
        elif r < 0.95:
            s = "".join(rng.choice(_HANGUL) for _ in range(rng.randint(2, 8))) + " "
        else:
            s = rng.choice(_EMOJI)
        s += rng.choice(_FW_PUNCT) if rng.random() < 0.5 else ""
        out.append(s)
        size += len(s.encode("utf-8"))
    return "".join(out)


def _trim(s: str, nbytes: int) -> str:
    """Cut to at most nbytes UTF-8 bytes on a character boundary."""
    b = s.encode("utf-8")[:nbytes]
    return b.decode("utf-8", "ignore")


def c1(n: int = 1000, seed: int = 1001) -> List[str]:
    rng = random.Random(seed)
    return [_trim(prose(rng, 1200), rng.randint(900, 1100)) for _ in range(n)]


def c2(n: int = 1000, seed: int = 1002) -> List[str]:
    rng = random.Random(seed)
    docs = []
    for i in range(n):
        size = rng.randint(900, 1100)
        docs.append(_trim(prose(rng, size + 100) if i % 2 == 0 else code(rng, size + 100), size))
    return docs


def c3(n: int = 10000, seed: int = 1003, doc_bytes: int = 4096) -> List[str]:
    rng = random.Random(seed)
    docs = []
    for _ in range(n):
        r = rng.random()
        size = rng.randint(int(doc_bytes * 0.9), int(doc_bytes * 1.1))
        gen = prose if r < 0.4 else json_doc if r < 0.7 else cjk
        docs.append(_trim(gen(rng, size + 100), size))
    return docs


def c4(n: int = 1_000_000, seed: int = 1004) -> List[str]:
    rng = random.Random(seed)
    docs = []
    lo, hi = math.log(64), math.log(512)
    for _ in range(n):
        size = int(math.exp(rng.uniform(lo, hi)))
        r = rng.random()
        body = prose(rng, size + 40) if r < 0.8 else code(rng, size + 40) if r < 0.9 else cjk(rng, size + 40)
        docs.append(_trim(body, size))
    return docs


def c5(n: int = 100, seed: int = 1005, doc_bytes: int = 2 << 20) -> List[str]:
    rng = random.Random(seed)
    docs = []
    for _ in range(n):
        paras: List[str] = []
        size = 0
        while size < doc_bytes:
            r = rng.random()
            psz = rng.randint(200, 3000)
            gen = prose if r < 0.4 else json_doc if r < 0.7 else cjk
            p = gen(rng, psz) + "\n\n"
            paras.append(p)
            size += len(p.encode("utf-8"))
        docs.append(_trim("".join(paras), doc_bytes))
    return docs


def worst_case(run_bytes: int = 65536) -> List[str]:
    return [
        "a" * run_bytes,
        " " * run_bytes,
        "".join(_CJK_COMMON[i % len(_CJK_COMMON)] for i in range(run_bytes // 3)),
        "\U0001f468‍\U0001f469‍\U0001f467‍\U0001f466" * (run_bytes // 25),
        "1234567890" * (run_bytes // 10),
        "\n" * run_bytes,
        "=" * run_bytes,
        ("word " * 5 + "\n") * (run_bytes // 26),
    ]
