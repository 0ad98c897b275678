cd $GRAFT_REPO_ROOT
timeout 300 python tools/dev/gpu_kbench.py slot 2>&1 | grep "^\["
timeout 300 python tools/dev/gpu_kbench.py slot 2>&1 | grep "^\["
SPL_LIB_PATH=$PWD/_abx/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2 1000 2>&1 | grep -v amdgpu.ids
SPL_LIB_PATH=$PWD/_abx/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2_wide 1000 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/dev/gpu_time_configs.py slot 2>/dev/null | grep GB/s | awk '{printf "%s %s %s %s | ", $1, $2, $6, $7} END {print ""}'
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -3
