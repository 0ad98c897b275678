cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -f gpurun_out/ab.log
bash tools/gpu_ab.sh default segascii
bash tools/gpu_suite.sh
