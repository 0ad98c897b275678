#!/bin/bash
# On the GPU box: tools/gpu_kbench_ab.sh v1 v2 ...   (libraries ${AB_DIR:-_ab}/lib_<v>.so; _ab/ is in .gpurunignore, so ship a comparison from another directory, e.g. AB_DIR=_abx, two interleaved rounds each)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/kbench_ab.log
for rep in 1 2; do for v in "$@"; do
  SPL_LIB_PATH=$PWD/${AB_DIR:-_ab}/lib_$v.so timeout 300 python tools/dev/gpu_kbench.py $v 2>/dev/null | grep "^\[" >> gpurun_out/kbench_ab.log
done; done
cat gpurun_out/kbench_ab.log
