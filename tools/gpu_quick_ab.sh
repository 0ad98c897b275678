# quick loop: cl100k parity tests on the default build, then kernel times of the given variants (default first)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout=300 --timeout-method=thread > gpurun_out/t_quick.log 2>&1
echo "rc=$?" >> gpurun_out/t_quick.log
bash tools/gpu_time_ab.sh default "$@"
