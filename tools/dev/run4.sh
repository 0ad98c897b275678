cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3
