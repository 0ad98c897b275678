cd ${GRAFT_REPO_ROOT:-/root/repo}
for q in 4 6 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-c5 2>/dev/null > /tmp/b_$q.json
python - $q <<'PY'
import json,sys
q=sys.argv[1]
d=json.loads(open(f"/tmp/b_{q}.json").read().strip().splitlines()[-1])
t=d["throughputs"]
print("Q", q, d["value"], d["c2_wide_rotation"]["value"], d["pipelined"]["value"], [(k, t[k]["c_abi_host"], t[k]["c_abi_host_pageable"], t[k]["encode_one_call_us"]) for k in ("c2","c3")], t["c2_custom_pattern"]["c_abi_host"])
PY
done
