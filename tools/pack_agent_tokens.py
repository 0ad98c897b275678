#!/usr/bin/env python3
"""Build-time data converter (runs in the build container only; reads /root/reference DATA).

Extracts the agent-token constant tables -- pure NAME -> id data of the reference's generated
classes CL100K_AGENT_TOKENS ... MISTRAL_V3_AGENT_TOKENS (src/python/agent_tokens_generated.rs) --
into splintr_amd/data/agent_tokens.json, from which splintr_amd/agent_tokens.py builds its classes."""
import json
import os
import re

REF = "/root/reference/src/python/agent_tokens_generated.rs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splintr_amd", "data", "agent_tokens.json")

src = open(REF, encoding="utf-8").read()
tables = {}
for m in re.finditer(r'#\[pyclass\(name = "(\w+)"[^\]]*\)\]', src):
    name = m.group(1)
    nxt = src.find("#[pyclass(", m.end())
    body = src[m.end(): nxt if nxt >= 0 else len(src)]
    tables[name] = {c.group(1): int(c.group(2)) for c in re.finditer(r"const (\w+): u32 = (\d+);", body)}
    print(name, len(tables[name]), "constants", min(tables[name].values()), "..", max(tables[name].values()))
with open(OUT, "w") as f:
    json.dump(tables, f, indent=0, sort_keys=False)
