cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh r3 new3 lean
SPL_LIB_PATH=$PWD/_abx/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2 1000 2>&1 | grep -v amdgpu.ids
AB_DIR=_abx bash tools/dev/run_cfg_ab.sh new3 lean
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3
