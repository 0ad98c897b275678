#!/usr/bin/env python3
"""Build-time data converter (runs in the build container only; reads /root/reference DATA).

Converts the reference's rank tables (``python/splintr/vocabs/*.tiktoken``: ``base64 SP rank``
lines, parsed with the semantics of ``src/core/vocab.rs:57-89``) into this repo's own binary
container ``splintr_amd/data/<name>.splv``, and extracts the special-token id tables
(``src/core/pretrained.rs:238-335, 402-547``: pure name->id data) into
``splintr_amd/data/special_tokens.json``.

SPLV v1 layout (little endian):
    char[4] "SPLV" | u32 version=1 | u32 n_records | u32 flags | u32 max_key_len
    n_records x { u32 rank | u16 key_len | u8 key[key_len] }      (file order preserved)
flags bit0 = keys are in ByteLevel space (deepseek_v3, mistral_v3; src/python/bindings.rs:122-129, 152-158).
"""
import base64
import json
import os
import re
import struct
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splintr_amd", "data")

VOCABS = {  # name -> (file, byte_level)
    "cl100k_base": ("cl100k_base.tiktoken", False),
    "o200k_base": ("o200k_base.tiktoken", False),
    "llama3": ("llama3.tiktoken", False),
    "deepseek_v3": ("deepseek_v3.tiktoken", True),
    "mistral_v3": ("mistral_v3_tekken.tiktoken", True),
}


def parse_tiktoken(data: bytes):
    recs = []
    for line in data.split(b"\n"):
        if not line:
            continue
        sp = line.rfind(b" ")
        if sp < 0:
            raise ValueError("Missing space separator")
        key = base64.b64decode(line[:sp], validate=True)
        rank = int(line[sp + 1:].strip())
        recs.append((key, rank))
    return recs


def pack(name: str):
    fn, bl = VOCABS[name]
    with open(os.path.join(REF, "python/splintr/vocabs", fn), "rb") as f:
        recs = parse_tiktoken(f.read())
    maxlen = max(len(k) for k, _ in recs)
    out = bytearray(struct.pack("<4sIIII", b"SPLV", 1, len(recs), 1 if bl else 0, maxlen))
    for key, rank in recs:
        out += struct.pack("<IH", rank, len(key))
        out += key
    path = os.path.join(OUT, name + ".splv")
    with open(path, "wb") as f:
        f.write(out)
    print(f"{name}: {len(recs)} records, max key {maxlen} B -> {path} ({len(out)} B)")


def eval_expr(expr: str, base: int) -> int:
    expr = expr.strip()
    m = re.fullmatch(r"base(?:\s*\+\s*(\d+))?", expr)
    if m:
        return base + int(m.group(1) or 0)
    return int(expr.replace("_", ""))


def special_tokens():
    src = open(os.path.join(REF, "src/core/pretrained.rs"), encoding="utf-8").read()
    # split into fn bodies
    fns = {}
    for m in re.finditer(r"fn (\w+)\(([^)]*)\)[^{]*\{", src):
        name = m.group(1)
        depth, i = 1, m.end()
        while depth:
            c = src[i]
            depth += c == "{"
            depth -= c == "}"
            i += 1
        fns[name] = src[m.end():i - 1]

    def run(fn: str, base: int = 0):
        out = {}
        body = fns[fn]
        for line in body.splitlines():
            line = line.strip()
            m = re.match(r'special\.insert\("((?:[^"\\]|\\.)*)"\.to_string\(\),\s*([^)]+)\);', line)
            if m:
                out[m.group(1)] = eval_expr(m.group(2), base)
                continue
            m = re.match(r"(insert_agent_tokens\w*)\(&mut special,\s*(\d+)\);", line)
            if m:
                out.update(run(m.group(1), int(m.group(2))))
        return out

    table = {
        "cl100k_base": run("cl100k_base_special_tokens"),
        "o200k_base": run("o200k_base_special_tokens"),
        "llama3": run("llama3_special_tokens"),
        "deepseek_v3": run("deepseek_v3_special_tokens"),
        "mistral_v3": run("mistral_v3_special_tokens"),
    }
    for k, v in table.items():
        print(f"special[{k}]: {len(v)} literals, ids {min(v.values())}..{max(v.values())}")
    with open(os.path.join(OUT, "special_tokens.json"), "w", encoding="utf-8") as f:
        json.dump(table, f, ensure_ascii=False, indent=0, sort_keys=False)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in (sys.argv[1:] or VOCABS):
        pack(n)
    special_tokens()
