cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/time_ab.log
for rep in 1 2; do for v in "$@"; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so; touch splintr_amd/libsplintr_hip.so
  timeout 200 python tools/dev/gpu_time_c2.py $v 2>/dev/null >> gpurun_out/time_ab.log
done; done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
