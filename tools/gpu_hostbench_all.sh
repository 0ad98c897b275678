cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/hostbench.log
for c in c2 c3 c4 c5; do timeout 300 python tools/host_path_bench.py $c >> gpurun_out/hostbench.log 2>&1; done
timeout 600 python -m pytest tests/test_gpu_hostpath.py -x -q --timeout=300 --timeout-method=thread > gpurun_out/t_host.log 2>&1; echo "rc=$?" >> gpurun_out/t_host.log
