# kernel / copy timeline of the C2 host path (rocprofv3 kernel + memory-copy trace)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o p -- python tools/host_path_bench.py ${CFG:-c2} > $O/log.txt 2>&1
python tools/rocpd_timeline.py $(find $O/kt -name "*.db" | head -1) ${NEV:-40} > gpurun_out/host_timeline.txt 2>&1
rm -rf $O/kt
