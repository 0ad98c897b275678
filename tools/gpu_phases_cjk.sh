cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cp _ab/lib_stamps.so splintr_amd/libsplintr_hip.so
timeout 300 python tools/dev/gpu_phases_cjk.py > gpurun_out/phases_cjk.log 2>&1
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
