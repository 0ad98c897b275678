# parity tests on the default build, then kernel times of A/B variants: bash tools/gpu_mask_ab.sh variant...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hostpath.py tests/test_gpu_decode.py -x -q -m gpu --timeout=600 --timeout-method=thread > gpurun_out/t_mask.log 2>&1
echo "rc=$?" >> gpurun_out/t_mask.log
bash tools/gpu_time_ab.sh default "$@"
