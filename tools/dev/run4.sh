cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_device_split.py -x -q 2>&1 | tail -3
for v in rx; do SPL_LIB_PATH=$PWD/_abx/lib_$v.so timeout 300 python tools/dev/rx_time.py $v 2>&1 | grep "^\["; done
timeout 600 python tools/host_path_bench.py custom 2>&1 | tail -1
