# round-2 pass B: full gpu suite (as the driver runs it) + the three throughputs per config
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout=600 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "gpu suite rc=$?" >> gpurun_out/t_all.log
rm -f gpurun_out/hostbench.log
for c in c2 c3 c4 c5; do timeout 300 python tools/host_path_bench.py $c >> gpurun_out/hostbench.log 2>&1; done
