cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "mistral" --timeout=600 --timeout-method=thread > gpurun_out/t_mistral.log 2>&1
echo "rc=$?" >> gpurun_out/t_mistral.log
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "rc=$?" >> gpurun_out/bench_full.err
