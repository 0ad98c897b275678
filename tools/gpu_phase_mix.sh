cd ${GRAFT_REPO_ROOT:-/root/repo}
cp _ab/lib_stamps.so splintr_amd/libsplintr_hip.so; touch splintr_amd/libsplintr_hip.so
bash tools/pmc_phase_mix.sh > gpurun_out/phase_mix.txt 2>&1
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
rm -rf gpurun_out/mix_*
