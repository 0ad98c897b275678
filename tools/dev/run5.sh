cd $GRAFT_REPO_ROOT
tools/profile_round4.sh > gpurun_out/r04_profile.log 2>&1
AB_DIR=_abx tools/r04_mix.sh > gpurun_out/r04_mix.log 2>&1
tail -3 gpurun_out/r04_profile.log
