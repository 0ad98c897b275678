cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/pysurf.log
for c in c2 c3 c4; do timeout 300 python tools/host_path_bench.py $c >> gpurun_out/pysurf.log 2>&1; done
timeout 300 python -m pytest tests/test_gpu_hostpath.py -x -q -k "python_surface" > gpurun_out/t_py.log 2>&1; echo rc=$? >> gpurun_out/t_py.log
