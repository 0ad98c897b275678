# On the GPU box: tools/dev/ab_rot.sh v1 v2 ...  (libraries _aby/lib_<v>.so): the C2 / c2_wide rotations (memo warm), two interleaved rounds each
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; rm -f gpurun_out/ab_rot.log
for rep in 1 2; do for v in "$@"; do MEMO_AB_QUICK=1 SPL_LIB_PATH=$PWD/_aby/lib_$v.so timeout 300 python tools/dev/memo_ab.py $v 2>&1 | grep "^\[" | grep "memo=1" | grep -v "c3" >> gpurun_out/ab_rot.log; done; done
sort -s -k1,4 gpurun_out/ab_rot.log | cut -c1-110
