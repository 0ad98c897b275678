# the driver's GPU tier: every -m gpu test, as the driver runs it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout=600 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "gpu suite rc=$?" >> gpurun_out/t_all.log
