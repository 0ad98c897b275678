#!/usr/bin/env python3
"""Generates tests/golden/reference_test_strings.json: the INPUT strings of the reference's own Python
tests that pin backend equality (regexr == PCRE2 == regexr without JIT) and UTF-8 boundary handling
(python/tests/test_cl100k.py:436-570 and the o200k / llama3 / deepseek_v3 siblings use the same sets),
with the ids the PCRE2-backed Python oracle gives them for every in-scope vocabulary.  The reference
asserts round trips and backend equality on these, not ids; PCRE2 (UTF|UCP) is one of the backends it
requires to agree, so these ids are what its PCRE2 build produces -- modulo the oracle's BPE
restatement, which the 18 reference-held id vectors pin.  Run in the build container; the file is committed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

PLAIN = [
    "The quick brown fox jumps over the lazy dog. 你好世界 🦀",                        # test_backend_consistency
    "I'm sorry you're hurting—breakups suck, but you'll get through it.",            # test_em_dash
    "He said, ‘Hello’ and she replied, “Goodbye”.",              # test_curly_quotes
    "Check if you're using valid credentials—API key, token—in headers.",            # test_mixed_multibyte
    "word—word", "a—b", "test—", "—start", "one—two—three",                           # test_em_dash_at_boundaries
    "Check your brake pads or rotors—they might be worn out.",
    "I'm sorry you're hurting—breakups suck.",                                        # test_batch_encode_multibyte
    "Check if you're using valid credentials.",
    "That weird noise could hint at a few things!",
    "Grinding while braking? Check your brake pads—they might be worn.",
    "Check credentials—API key—in headers.",                                          # test_backend_consistency_multibyte
    "That weird noise could hint at a few things—grinding, rattling, knocking.",     # test_large_batch_multibyte_parallel
    "Grinding while braking? Check your brake pads or rotors—they might be worn out.",
    "word—word—word—word—word",
    "A 403 Forbidden error means your API request is authenticated but lacks permission.",
    "Hello, world!", "   \n\t  ", "Multi-line\ntext\nwith\nnewlines", "def hello():\n    print('Hello')",
    " hello world ", "hello  world",
]
SPECIAL = [                                                                           # test_batch_encode_with_special_multibyte
    "<|user|>I'm hurting—help me<|assistant|>Here's how—step by step:",
    "<|system|>You're a helpful assistant<|user|>What's this—a bug?",
]
# the seven base texts that test_large_batch_multibyte_parallel repeats 100 times
LARGE_BATCH_BASE = [PLAIN[1], PLAIN[3], PLAIN[15], PLAIN[16], PLAIN[2], PLAIN[17], PLAIN[18]]

out = {"_source": "input strings: /root/reference/python/tests/test_cl100k.py:436-570 (data only); ids: oracle/pyoracle.py with "
                  "libpcre2-8 %s (Unicode %s), tools/make_pin_fixtures.py" % O.pcre2_versions(),
       "plain": PLAIN, "special": SPECIAL, "large_batch_base": LARGE_BATCH_BASE, "ids": {}, "ids_with_special": {}}
for name in ("cl100k_base", "o200k_base", "llama3", "deepseek_v3", "mistral_v3"):
    t = O.Oracle.from_pretrained(name, engine="pcre2")
    out["ids"][name] = [t.encode(s) for s in PLAIN]
    out["ids_with_special"][name] = [t.encode_with_special(s) for s in SPECIAL]
    for s, ids in zip(PLAIN, out["ids"][name]):
        assert t.decode_bytes(ids).decode("utf-8") == s
with open(os.path.join(ROOT, "tests", "golden", "reference_test_strings.json"), "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False, indent=0)
print("ok", len(PLAIN), len(SPECIAL))
