cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in gnp gpre; do for g in 0 1; do
  SPL_GRAPH_REPLAY=$g SPL_LIB_PATH=$PWD/_abx/lib_$v.so timeout 300 python tools/dev/gpu_kbench.py ${v}_replay$g 2>/dev/null | grep "^\["
done; done; done
