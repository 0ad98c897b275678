"""Deterministic synthetic corpora for BASELINE.json's configs.

Self-contained (stdlib `random.Random(seed)` only), no identical documents, natural Zipf-like
word reuse.  Nothing here uses code points assigned after Unicode 13, so every Unicode table in
play (PCRE2/Unicode 14, Python `regex`, the class table) agrees on all of them.

    c1(n)  cl100k   English prose, ~1 KB docs                    seed 1001
    c2(n)  cl100k   50 % prose / 50 % code, ~1 KB docs           seed 1002   (the bench workload)
    c2_wide(n)      the same mix over a WIDE lexicon: >= 20 000 distinct words from a syllable grammar,
                    Zipf-distributed (natural 1 MB English/code has ~10x the distinct words of c2)  seed 2002
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


# ---- wide lexicon (c2_wide): words from a syllable grammar, rank-frequency ~ 1 / rank -----------------
_ONSETS = ("b c d f g h j k l m n p r s t v w y z bl br ch cl cr dr fl fr gl gr pl pr qu sc sh sk sl sm sn sp st "
           "str sw th tr tw wh wr").split() + [""]
_VOWELS = "a e i o u ai au ea ee ie io oa oo ou ue y".split()
_CODAS = ("b d f g k l m n p r s t x ck ct ft ld lf lk ll lm lp lt mp nd ng nk nt pt rd rk rm rn rt sk sp ss st "
          "th").split() + ["", "", "", ""]
_STEMS = ("act add age air allow answer appear apply arm art ask back ball bank base bear beat bed begin believe bill "
          "bit black blood blue board boat body book born box boy break bring brother build burn business buy call camp "
          "car card care carry case catch cause cell center chance change charge check child choose church circle city "
          "claim class clean clear close cloud coast cold color common company compare complete condition connect consider "
          "contain control cook copy corner cost count country course cover create cross crowd cry cut dance dark data "
          "deal decide deep design detail develop differ direct discover divide doctor door double draw dream dress drink "
          "drive drop dry early earth ease east eat edge effect eight electric end enemy energy engine enter equal even "
          "event ever exact example except excite exercise expect experience explain express face fact fair fall family "
          "farm fast father fear feed feel field fight figure fill final find fine finger finish fire fish fit five flat "
          "floor flow flower fly follow food foot force forest form forward found four free fresh friend front fruit full "
          "game garden gather general gentle girl give glass gold govern grand grass gray green ground group grow guess "
          "guide hair half hand happen hard hat head hear heart heat heavy help high hill history hit hold hole home hope "
          "horse hot hour house human hundred hunt hurry ice idea include industry inform insect interest iron island join "
          "joy jump keep key kill kind king knew land language large late laugh law lay lead learn leave left leg length "
          "level lie lift light line list listen live locate lock log lone look lost loud love low machine magnet main "
          "major map mark market mass master match material matter mean measure meat meet melody metal method middle milk "
          "million mind mine minute miss mix modern moment money month moon morning mother motion mount mouth move music "
          "name nation nature near need neighbor new night nine noise noon north nose note nothing notice noun number "
          "object observe ocean offer office oil open operate opposite order organ original paint paper paragraph parent "
          "part party pass past path pattern pay people perhaps person phrase pick picture piece place plain plan plane "
          "plant play please plural poem point poor populate port pose position possible post pound power practice prepare "
          "present press pretty print probable problem process produce product proper protect prove provide pull push "
          "quart question quick quiet quite race radio rail rain raise range rather reach read ready real reason receive "
          "record red region remember repeat reply represent require rest result rich ride ring rise river road rock roll "
          "room root rope rose round row rule run safe sail salt sand save scale school science score sea search season "
          "seat second section seed seem segment select self sell send sense sentence separate serve set settle seven "
          "shape share sharp shell shine ship shoe shop shore short shoulder shout show side sight sign silent silver "
          "similar simple sing single sister sit size skill skin sky sleep slip slow small smell smile snow soft soil "
          "soldier solution solve son song soon sound south space speak special speech speed spell spend spoke spot spread "
          "spring square stand star start station stay stead steam steel step stick stone stood stop store story straight "
          "strange stream street stretch string strong student study subject substance subtract success sudden suffix sugar "
          "suggest suit summer sun supply support sure surface surprise swim syllable symbol system table tail talk tall "
          "teach team teeth tell temperature term test thank thick thin thing think third thought thousand thus tie time "
          "tiny tire together tone tool top total touch toward town track trade train travel tree triangle trip trouble "
          "truck true try tube turn twenty type unit until usual valley value vary verb view village visit voice vowel "
          "wait walk wall want war warm wash watch water wave wear weather week weight west wheel white whole wide wife "
          "wild win wind window wing winter wire wish woman wonder wood word write wrong yard yellow young").split()
_PREFIXES = "un re pre dis over under inter mis non out sub super anti co de en fore mid semi trans up".split()
_SUFFIXES = ("s ed ing er ers ly ness ment ments tion tions able ful less ist ists ism ize ized ity al ic ous ive "
             "ship hood ward wise like").split()
_WIDE_N = 32768


class _Lexicon:
    """_WIDE_N distinct words, drawn with probability ~ 1 / (rank + 2.7) (Zipf-Mandelbrot): the head is real
    English (the c2 word list, then common stems), the body is derived forms and compounds of those stems
    (prefix + stem + suffix: words a BPE vocabulary covers in two or three tokens, as it does natural derived
    words), the tail pseudo-words from a syllable grammar (names, jargon)."""

    def __init__(self, seed: int = 20020):
        import bisect
        rng = random.Random(seed)
        seen, words = set(), []

        def add(w):
            if len(w) >= 2 and w not in seen:
                seen.add(w)
                words.append(w)
        for w in _COMMON:
            add(w)
        stems = list(_STEMS)
        rng.shuffle(stems)
        for w in stems:
            add(w)
        while len(words) < _WIDE_N:
            r = rng.random()
            if r < 0.55:                               # derived form
                w = rng.choice(_STEMS)
                if rng.random() < 0.35:
                    w = rng.choice(_PREFIXES) + w
                if rng.random() < 0.8:
                    sfx = rng.choice(_SUFFIXES)
                    if w.endswith("e") and sfx[0] in "aeiou":
                        w = w[:-1]
                    w += sfx
            elif r < 0.8:                              # compound
                w = rng.choice(_STEMS) + rng.choice(_STEMS)
            else:                                      # pseudo-word
                nsyl = 1 + min(int(rng.paretovariate(1.6)) - 1, 2)
                w = "".join(rng.choice(_ONSETS) + rng.choice(_VOWELS) + rng.choice(_CODAS) for _ in range(nsyl))
            add(w)
        self.words = words
        acc, cum = 0.0, []
        for r in range(len(words)):
            acc += 1.0 / (r + 2.7)
            cum.append(acc)
        self.cum, self.total, self._bisect = cum, acc, bisect.bisect_left

    def draw(self, rng: random.Random) -> str:
        return self.words[min(self._bisect(self.cum, rng.random() * self.total), len(self.words) - 1)]


_LEX = None


def _lexicon() -> "_Lexicon":
    global _LEX
    if _LEX is None:
        _LEX = _Lexicon()
    return _LEX


def _word_wide(rng: random.Random) -> str:
    r = rng.random()
    if r < 0.92:
        return _lexicon().draw(rng)
    if r < 0.96:
        return rng.choice(_CONTR)
    if r < 0.98:
        return str(rng.randint(0, 10 ** rng.randint(1, 7)))
    return _lexicon().draw(rng).capitalize()


def _ident_wide(rng: random.Random) -> str:
    """identifiers composed from lexicon words: snake_case, camelCase, PascalCase, plain"""
    lex = _lexicon()
    k = 1 + min(int(rng.paretovariate(1.3)) - 1, 3)
    parts = [lex.draw(rng) for _ in range(k)]
    r = rng.random()
    if r < 0.4:
        return "_".join(parts)
    if r < 0.75:
        return parts[0] + "".join(p.capitalize() for p in parts[1:])
    if r < 0.9:
        return "".join(p.capitalize() for p in parts)
    return parts[0]


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


def prose(rng: random.Random, nbytes: int, word=None) -> str:
    word = word or _word
    out: List[str] = []
    size = 0
    cap = True
    while size < nbytes:
        w = word(rng)
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


def code(rng: random.Random, nbytes: int, ident=None, word=None) -> str:
    out: List[str] = []
    size = 0
    indent = 0
    style = rng.choice(["py", "c", "json"])
    while size < nbytes:
        ind = ("\t" * indent) if rng.random() < 0.2 else ("    " * indent)
        if ident is None:
            a, b, c = rng.choice(_IDENT), rng.choice(_IDENT), rng.choice(_IDENT)
        else:
            a, b, c = ident(rng), ident(rng), ident(rng)
        k = rng.random()
        if style == "py":
            if k < 0.2:
                line = f"{ind}def {a}({b}, {c}=None):"
                indent = min(indent + 1, 4)
            elif k < 0.4:
                line = f"{ind}{a} = {b}[{rng.randint(0, 4096)}] + {c}.{(ident(rng) if ident else rng.choice(_IDENT))}({rng.random():.4f})"
            elif k < 0.55:
                line = f"{ind}{rng.choice(_KEYW)} {a} {rng.choice(['==', '!=', '<=', 'in', 'is not'])} {b}:"
                indent = min(indent + 1, 4)
            elif k < 0.7:
                line = f"{ind}return {a} if {b} else '{c}_{rng.randint(0, 99)}'  # {prose(rng, 20, word).strip()}"
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
                line = f"{ind}/* {prose(rng, 30, word).strip()} */"
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


def c2_wide(n: int = 1000, seed: int = 2002) -> List[str]:
    """C2's mix (50 % prose / 50 % code, ~1 KB documents) over the wide lexicon: the whole-chunk hit rate, the
    misses per tile and the table traffic of natural text instead of a 2 k-word vocabulary's."""
    rng = random.Random(seed)
    docs = []
    for i in range(n):
        size = rng.randint(900, 1100)
        docs.append(_trim(prose(rng, size + 100, _word_wide) if i % 2 == 0
                          else code(rng, size + 100, _ident_wide, _word_wide), size))
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
