#!/usr/bin/env python3
"""Build-time generator of the code-point class table used by the pre-tokeniser kernels.

The classes are PROBED from the system libpcre2-8 (the reference's own optional regex backend,
``src/core/tokenizer.rs:474-481``, flags UTF|UCP) so the table is, by construction, the exact
character-class semantics of ``\\p{Lu} \\p{Ll} \\p{Lt} \\p{Lm} \\p{Lo} \\p{M} \\p{N} \\s`` that the
oracle's split engine applies (PCRE2 10.39 => Unicode 14.0.0).  Also probes which code points the
caseless contraction letters ``(?i:s|t|r|e|v|m|l|d)`` match under UTF|UCP.

Output: ``splintr_amd/data/unicode_classes.bin``
    char[4] "SPLU" | u32 version=2 | u32 block_shift | u32 n_blocks | char[16] unicode_version
    u16 stage1[0x110000 >> block_shift]      block index per code-point block
    u8  stage2[n_blocks << block_shift]      class code per code point
    u32 n_fold | n_fold x { u32 code_point | u32 ascii_lower }   caseless partners of s,t,r,e,v,m,l,d
    (version 2) u32 gc_n_blocks | u16 gc_stage1[0x110000 >> block_shift] | u8 gc_stage2[gc_n_blocks << block_shift]
                the GENERAL CATEGORY of every code point (GC_NAMES), for the host splitter's \p{P} \p{S} \p{Z}
                \p{Nd} ... \d (csrc/spl_regex.cpp); the GPU scanner only needs the classes above
    (version 3) u32 n_scripts | n_scripts x { char[32] name | u32 n_ranges | n_ranges x { u32 first | u32 last } }
                the SCRIPT property as the engine classes it (\p{Han}, \p{Hiragana}, \p{Latin} ... -- every script name the engine
                accepts of SCRIPT_NAMES), as inclusive code-point ranges: custom split patterns of CJK-aware tokenizers
Class codes: see CLASS_NAMES (shared with splintr_amd/csrc/spl_scan.h).

``--engine regex``: the same tables probed from the Python ``regex`` module instead (its Unicode version is newer than
PCRE2 10.39's 14.0) -> ``unicode_classes_regex.bin``: the second table VERDICT r03 #8 asks for; the reference's default
engine (regexr, Cargo.toml:41) ships tables of an unknown version, so a caller who knows better can choose.
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Pcre2Pattern, pcre2_versions  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splintr_amd", "data")

CLASS_NAMES = ["P", "AP", "SP", "WS", "NL", "N", "Lu", "Ll", "Lt", "Lm", "Lo", "M"]
# general categories, code = index (csrc/spl_regex.cpp GC_*); Cn = everything no other category claims
GC_NAMES = ["Cn", "Lu", "Ll", "Lt", "Lm", "Lo", "Mn", "Mc", "Me", "Nd", "Nl", "No", "Pc", "Pd", "Ps", "Pe", "Pi", "Pf", "Po",
            "Sm", "Sc", "Sk", "So", "Zs", "Zl", "Zp", "Cc", "Cf", "Cs", "Co"]
# Unicode script names (UAX #24 through Unicode 15.1); an engine that does not know one is skipped for that name
SCRIPT_NAMES = """Adlam Ahom Anatolian_Hieroglyphs Arabic Armenian Avestan Balinese Bamum Bassa_Vah Batak Bengali Bhaiksuki Bopomofo Brahmi Braille
Buginese Buhid Canadian_Aboriginal Carian Caucasian_Albanian Chakma Cham Cherokee Chorasmian Common Coptic Cuneiform Cypriot Cypro_Minoan Cyrillic
Deseret Devanagari Dives_Akuru Dogra Duployan Egyptian_Hieroglyphs Elbasan Elymaic Ethiopic Georgian Glagolitic Gothic Grantha Greek Gujarati
Gunjala_Gondi Gurmukhi Han Hangul Hanifi_Rohingya Hanunoo Hatran Hebrew Hiragana Imperial_Aramaic Inherited Inscriptional_Pahlavi
Inscriptional_Parthian Javanese Kaithi Kannada Katakana Kawi Kayah_Li Kharoshthi Khitan_Small_Script Khmer Khojki Khudawadi Lao Latin Lepcha Limbu
Linear_A Linear_B Lisu Lycian Lydian Mahajani Makasar Malayalam Mandaic Manichaean Marchen Masaram_Gondi Medefaidrin Meetei_Mayek Mende_Kikakui
Meroitic_Cursive Meroitic_Hieroglyphs Miao Modi Mongolian Mro Multani Myanmar Nabataean Nag_Mundari Nandinagari New_Tai_Lue Newa Nko Nushu
Nyiakeng_Puachue_Hmong Ogham Ol_Chiki Old_Hungarian Old_Italic Old_North_Arabian Old_Permic Old_Persian Old_Sogdian Old_South_Arabian Old_Turkic
Old_Uyghur Oriya Osage Osmanya Pahawh_Hmong Palmyrene Pau_Cin_Hau Phags_Pa Phoenician Psalter_Pahlavi Rejang Runic Samaritan Saurashtra Sharada
Shavian Siddham SignWriting Sinhala Sogdian Sora_Sompeng Soyombo Sundanese Syloti_Nagri Syriac Tagalog Tagbanwa Tai_Le Tai_Tham Tai_Viet Takri
Tamil Tangsa Tangut Telugu Thaana Thai Tibetan Tifinagh Tirhuta Toto Ugaritic Vai Vithkuqi Wancho Warang_Citi Yezidi Yi Zanabazar_Square""".split()
C = {n: i for i, n in enumerate(CLASS_NAMES)}
BLOCK_SHIFT = 7


def all_codepoints_utf8():
    cps = [cp for cp in range(0x110000) if not (0xD800 <= cp <= 0xDFFF)]
    data = "".join(map(chr, cps)).encode("utf-8")
    # byte offset -> cp
    offs = {}
    o = 0
    for cp in cps:
        offs[o] = cp
        o += 1 if cp < 0x80 else 2 if cp < 0x800 else 3 if cp < 0x10000 else 4
    offs[o] = 0x110000
    return data, offs


ENGINE = "pcre2"


def members(pattern, data, offs):
    s = set()
    if ENGINE == "regex":
        import regex
        text = data.decode("utf-8")
        for m in regex.finditer(pattern, text):
            s.update(map(ord, m.group(0)))
        return s
    for a, b in Pcre2Pattern(pattern).find_iter(data):
        o = a
        while o < b:
            cp = offs[o]
            s.add(cp)
            o += 1 if cp < 0x80 else 2 if cp < 0x800 else 3 if cp < 0x10000 else 4
    return s


def two_stage(cls):
    bs = 1 << BLOCK_SHIFT
    blocks, stage1, stage2 = {}, [], bytearray()
    for b in range(0x110000 >> BLOCK_SHIFT):
        blk = bytes(cls[b * bs:(b + 1) * bs])
        idx = blocks.get(blk)
        if idx is None:
            idx = blocks[blk] = len(blocks)
            stage2 += blk
        stage1.append(idx)
    return stage1, stage2, len(blocks)


def main():
    global ENGINE
    if "--engine" in sys.argv:
        ENGINE = sys.argv[sys.argv.index("--engine") + 1]
    if ENGINE == "regex":
        import regex
        import unicodedata
        ver, uver = regex.__version__, "regex-" + regex.__version__[:9]
        print("Python regex", ver, "(unicodedata of this interpreter:", unicodedata.unidata_version + ")")
    else:
        ver, uver = pcre2_versions()
        print("PCRE2", ver, "Unicode", uver)
    data, offs = all_codepoints_utf8()
    sets = {k: members(p, data, offs) for k, p in {
        "Lu": r"\p{Lu}", "Ll": r"\p{Ll}", "Lt": r"\p{Lt}", "Lm": r"\p{Lm}", "Lo": r"\p{Lo}",
        "L": r"\p{L}", "M": r"\p{M}", "N": r"\p{N}", "S": r"\s",
        "notS": r"\S", "X": r"[^\r\n\p{L}\p{N}]", "Pset": r"[^\s\p{L}\p{N}]",
    }.items()}
    L = sets["Lu"] | sets["Ll"] | sets["Lt"] | sets["Lm"] | sets["Lo"]
    assert L == sets["L"], "L != Lu|Ll|Lt|Lm|Lo"
    for a in ("Lu", "Ll", "Lt", "Lm", "Lo", "M", "N", "S"):
        for b in ("Lu", "Ll", "Lt", "Lm", "Lo", "M", "N", "S"):
            if a < b:
                assert not (sets[a] & sets[b]), (a, b)
    allcp = set(offs.values()) - {0x110000}
    assert sets["notS"] == allcp - sets["S"]
    assert sets["X"] == allcp - L - sets["N"] - {0x0A, 0x0D}
    assert sets["Pset"] == allcp - L - sets["N"] - sets["S"]
    assert 0x0A in sets["S"] and 0x0D in sets["S"] and 0x20 in sets["S"]
    print({k: len(v) for k, v in sets.items()})
    print("whitespace:", " ".join(f"U+{c:04X}" for c in sorted(sets["S"])))

    cls = bytearray(0x110000)  # default P (also for surrogates: never seen in valid UTF-8)
    for name in ("Lu", "Ll", "Lt", "Lm", "Lo", "M", "N"):
        code = C[name]
        for cp in sets[name]:
            cls[cp] = code
    for cp in sets["S"]:
        cls[cp] = C["WS"]
    cls[0x20] = C["SP"]
    cls[0x0A] = C["NL"]
    cls[0x0D] = C["NL"]
    cls[0x27] = C["AP"]

    stage1, stage2, nblocks = two_stage(cls)
    print(f"blocks: {nblocks} x {1 << BLOCK_SHIFT} B = {len(stage2)} B; stage1 {len(stage1) * 2} B")

    # general categories: one probe per category, runs of members per match
    gc = bytearray(0x110000)
    seen = set()
    for code, name in enumerate(GC_NAMES):
        if name in ("Cn", "Cs"):
            continue                                   # (surrogates never occur in UTF-8; unassigned = the rest)
        m = members(r"\p{%s}+" % name, data, offs)
        assert not (m & seen), name
        seen |= m
        for cp in m:
            gc[cp] = code
    for cp in range(0xD800, 0xE000):
        gc[cp] = GC_NAMES.index("Cs")
    for grp, parts in (("L", ("Lu", "Ll", "Lt", "Lm", "Lo")), ("M", ("Mn", "Mc", "Me")), ("N", ("Nd", "Nl", "No"))):
        u = set()
        for nm in parts:
            u |= {cp for cp in sets[grp] if gc[cp] == GC_NAMES.index(nm)}
        assert u == sets[grp], grp                     # the class table and the categories agree
    g1, g2, gnb = two_stage(gc)
    print(f"general categories: {gnb} blocks = {len(g2)} B; unassigned (Cn): {sum(1 for cp in allcp if gc[cp] == 0)}")

    folds = []
    for ch in "stremvld":
        m = members(f"(?i:{ch})", data, offs)
        for cp in sorted(m):
            folds.append((cp, ord(ch)))
        print(f"(?i:{ch}) ->", [f"U+{c:04X}" for c in sorted(m)])

    # scripts: the runs of \p{Name}+ over all code points in order ARE its ranges
    scripts = []
    for name in SCRIPT_NAMES:
        try:
            m = members(r"\p{%s}+" % name, data, offs)
        except Exception:
            continue
        if not m:
            continue
        rs, cps = [], sorted(m)
        a = b = cps[0]
        for cp in cps[1:]:
            if cp == b + 1 or (b == 0xD7FF and cp == 0xE000):
                b = cp
            else:
                rs.append((a, b)); a = b = cp
        rs.append((a, b))
        scripts.append((name, rs))
    print(f"scripts: {len(scripts)} of {len(SCRIPT_NAMES)} names known to the engine, {sum(len(r) for _, r in scripts)} ranges; "
          f"Han {sum(b - a + 1 for a, b in dict(scripts).get('Han', []))} code points")

    out = bytearray(struct.pack("<4sIII16s", b"SPLU", 3, BLOCK_SHIFT, nblocks, uver.encode()[:16]))
    out += struct.pack(f"<{len(stage1)}H", *stage1)
    out += stage2
    out += struct.pack("<I", len(folds))
    for cp, lo in folds:
        out += struct.pack("<II", cp, lo)
    out += struct.pack("<I", gnb)
    out += struct.pack(f"<{len(g1)}H", *g1)
    out += g2
    out += struct.pack("<I", len(scripts))
    for name, rs in scripts:
        out += struct.pack("<32sI", name.encode(), len(rs))
        for a, b in rs:
            out += struct.pack("<II", a, b)
    path = os.path.join(OUT, "unicode_classes.bin" if ENGINE == "pcre2" else "unicode_classes_regex.bin")
    with open(path, "wb") as f:
        f.write(out)
    print("wrote", path, len(out), "B")


if __name__ == "__main__":
    main()
