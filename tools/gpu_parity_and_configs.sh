# cl100k/o200k/mistral parity tests on the default build, the bench-batch kernel times and the larger configurations, default against the given variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout=300 --timeout-method=thread > gpurun_out/t_quick.log 2>&1
echo "rc=$?" >> gpurun_out/t_quick.log
bash tools/gpu_time_ab.sh default "$@"
bash tools/gpu_configs_ab.sh default "$@"
