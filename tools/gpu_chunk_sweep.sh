cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/chunk_sweep.txt
for cb in ${CBS:-2097152 4194304 8388608 16777216}; do
  for c in ${CFGS:-c3 c5}; do
    timeout 300 python - "$c" "$cb" >> gpurun_out/chunk_sweep.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, "tools")
import host_path_bench as H
out = H.measure(sys.argv[1], python_surface=False, options={"chunk_bytes": int(sys.argv[2])})
print(json.dumps({k: out[k] for k in ("config", "c_abi_host", "c_abi_host_pageable", "kernel_hbm")} | {"chunk_bytes": int(sys.argv[2])}))
PY
  done
done
