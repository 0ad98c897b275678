cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh toargs split
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; tail -4 gpurun_out/r04_gputests.log
