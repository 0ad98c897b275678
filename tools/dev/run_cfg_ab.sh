# On the GPU box: tools/dev/run_cfg_ab.sh v1 v2 ...  (libraries ${AB_DIR:-_abx}/lib_<v>.so; C3 / C4 / C5 / mistral / C2x8 kernel-only, two interleaved rounds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do
  SPL_LIB_PATH=$PWD/${AB_DIR:-_abx}/lib_$v.so timeout 300 python tools/dev/gpu_time_configs.py $v 2>/dev/null | grep GB/s | awk '{printf "%s %s %s %s | ", $1, $2, $6, $7} END {print ""}'
done; done
