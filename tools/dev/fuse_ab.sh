cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; rm -f gpurun_out/fuse_ab.log
for rep in 1 2; do for v in "$@"; do FUSE_AB_QUICK=1 SPL_LIB_PATH=$PWD/_aby/lib_$v.so timeout 200 python tools/dev/fuse_ab.py $v 2>&1 | grep "^\[" | grep "fuse=1" >> gpurun_out/fuse_ab.log; done; done
sort -s -k1,3 gpurun_out/fuse_ab.log
