cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh akind toargs
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_gputests.log 2>&1; tail -6 gpurun_out/r04_gputests.log
