cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_device_split.py -x -q 2>&1 | tail -2
timeout 600 python tools/host_path_bench.py custom 2>&1 | tail -1
