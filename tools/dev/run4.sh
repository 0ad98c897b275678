cd $GRAFT_REPO_ROOT
timeout 600 python tools/host_path_bench.py custom 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_custom_pattern.py -x -q 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_stress.py -x -q -k custom 2>&1 | tail -2
