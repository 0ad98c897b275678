cd $GRAFT_REPO_ROOT
AB_DIR=_abx tools/gpu_kbench_ab.sh doc1 na32 na16
for v in doc1 na32 na16; do SPL_LIB_PATH=$PWD/_abx/lib_$v.so timeout 300 python tools/dev/gpu_time_configs.py $v 2>/dev/null | grep GB/s | awk '{printf "%s %s %s %s | ", $1, $2, $6, $7} END {print ""}'; done
