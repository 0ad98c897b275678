cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh tidx akind
AB_DIR=_abx bash tools/dev/run_cfg_ab.sh tidx akind
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_tiktoken_format.py -x -q 2>&1 | tail -3
