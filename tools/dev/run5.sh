cd $GRAFT_REPO_ROOT
tools/profile_round4.sh > gpurun_out/r04_profile.log 2>&1
AB_DIR=_abx tools/r04_mix.sh > gpurun_out/r04_mix.log 2>&1
for v in tidx notail; do SPL_LIB_PATH=$PWD/_abx/lib_$v.so timeout 300 python tools/dev/gpu_time_configs.py $v 2>/dev/null | grep GB/s | awk '{printf "%s %s %s %s | ", $1, $2, $6, $7} END {print ""}'; done > gpurun_out/r04_tail_cost_raw.txt
cat gpurun_out/r04_tail_cost_raw.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; tail -4 gpurun_out/r04_gputests.log
