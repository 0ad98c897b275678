# end-of-round check, as the driver does it: GPU tier of the tests, smoke(), the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu --timeout=600 --timeout-method=thread > gpurun_out/t_all.log 2>&1
echo "gpu suite rc=$?" >> gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.json 2> gpurun_out/bench_driver_cmd.err; echo "rc=$?" >> gpurun_out/bench_driver_cmd.err
grep -o "libsplintr_hip.so\|_spl_py.so" /proc/self/maps | sort -u > /dev/null
