cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/phases64.log
cp _ab/lib_stamps.so splintr_amd/libsplintr_hip.so
timeout 200 python tools/dev/gpu_phases.py cl100k_base c2 64 > gpurun_out/phases64.log 2>&1
timeout 200 python tools/dev/gpu_phases.py cl100k_base c2 500 >> gpurun_out/phases64.log 2>&1
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
