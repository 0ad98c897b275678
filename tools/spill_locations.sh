#!/bin/bash
# Where k_pretok<800,192>'s scratch accesses sit, by source line (static count): tools/spill_locations.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -std=c++17 -Wno-unused-value --cuda-device-only -S -g1 "$@" -o /tmp/spl_dev.s splintr_amd/csrc/spl_api.hip 2>/dev/null
python3 - <<'PY'
import re, collections
files = {}
cur = None
infn = False
cnt = collections.Counter()
n_inst = 0
for line in open('/tmp/spl_dev.s'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if line.startswith('_ZN3spl8k_pretokILi800ELi192'):
        infn = True
        continue
    if infn and line.startswith('.Lfunc_end'):
        infn = False
    if not infn:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    s = line.strip()
    if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
        continue
    n_inst += 1
    if s.startswith('scratch_'):
        cnt[(cur, s.split()[0])] += 1
print(n_inst, 'machine instructions in k_pretok<800,192>;', sum(cnt.values()), 'scratch accesses')
for (loc, op), c in sorted(cnt.items(), key=lambda kv: (kv[0][0] or ('', 0))):
    print(f'  {c:3d} {op:24s} {loc[0] if loc else "?"}:{loc[1] if loc else 0}')
PY
