# On the GPU box: tools/dev/tail_cuts.sh v1 v2 ...   (libraries _aby/lib_<v>.so; two interleaved rounds; GB/s, "-": chunk memo off)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do SPL_LIB_PATH=$PWD/_aby/lib_$v.so timeout 400 python tools/dev/tail_cuts.py $v 2>/dev/null | grep "|"; done; done
