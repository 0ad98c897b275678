# phase stamps of one workgroup (default 1053: ends last on the bench batch) for the given stamp builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/phases_slow.log
WG=${WG:-1053}
for v in "$@"; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so
  echo "=== $v workgroup $WG" >> gpurun_out/phases_slow.log
  SPL_DEBUG_WG=$WG timeout 200 python tools/dev/gpu_phases.py >> gpurun_out/phases_slow.log 2>&1
done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
