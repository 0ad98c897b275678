cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/phases.log
bash tools/gpu_phases_ab.sh stamps
cp _ab/lib_stamps.so splintr_amd/libsplintr_hip.so
timeout 300 python tools/dev/gpu_phase_times.py >> gpurun_out/phases.log 2>&1
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
