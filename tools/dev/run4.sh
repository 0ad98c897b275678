cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_device_split.py -x -q -k special 2>&1 | tail -12 | cut -c1-300
