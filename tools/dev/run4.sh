cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_custom_pattern.py tests/test_gpu_device_split.py -x -q 2>&1 | tail -8 | cut -c1-300
