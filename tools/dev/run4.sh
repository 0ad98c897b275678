cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh toargs clean
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3
