# round-2 validation pass A: host pipeline tests (traced), parity suite on both kernel routings, quick bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
SPL_TRACE=1 timeout 420 python -X faulthandler -m pytest tests/test_gpu_hostpath.py -x -v --timeout=120 --timeout-method=thread 2>&1 | tail -c 200000 > gpurun_out/t_host.log
echo "hostpath rc=$?" >> gpurun_out/t_host.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout=300 --timeout-method=thread > gpurun_out/t_par_A.log 2>&1
echo "parity A rc=$?" >> gpurun_out/t_par_A.log
cp splintr_amd/libsplintr_hip.so _ab/lib_round1.so
cp _ab/lib_tilelist.so splintr_amd/libsplintr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout=300 --timeout-method=thread > gpurun_out/t_par_B.log 2>&1
echo "parity B rc=$?" >> gpurun_out/t_par_B.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughputs --no-c4 > gpurun_out/bench_B.json 2> gpurun_out/bench_B.err
cp _ab/lib_round1.so splintr_amd/libsplintr_hip.so
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughputs --no-c4 > gpurun_out/bench_A.json 2> gpurun_out/bench_A.err
