cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do echo "=== $v"; SPL_LIB_PATH=_aby/lib_$v.so RX_TIME_QUICK=1 timeout 200 python tools/dev/rx_time.py 2>&1 | grep -v amdgpu.ids; done
