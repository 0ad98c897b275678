cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so; touch splintr_amd/libsplintr_hip.so
  echo "=== $v" >> gpurun_out/phases.log
  timeout 200 python tools/dev/gpu_phases.py >> gpurun_out/phases.log 2>&1
done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
