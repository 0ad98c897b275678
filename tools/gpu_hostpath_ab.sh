# C ABI host path (C2, 1 MB and 3 MB) for the given library variants, interleaved twice; host-path parity on the default
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/hostpath_ab.log
for rep in 1 2; do for v in "$@"; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so
  for n in - 3000; do echo "$v $(timeout 300 python tools/host_path_bench.py c2 $n 2>/dev/null | tail -1)" >> gpurun_out/hostpath_ab.log; done
done; done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
timeout 600 python -m pytest tests/test_gpu_hostpath.py -x -q -m gpu --timeout=300 --timeout-method=thread > gpurun_out/t_hostpath.log 2>&1; echo rc=$? >> gpurun_out/t_hostpath.log
