# kernel / copy timeline of the C3 host -> host call (40 MB, five 8 MiB chunks through the pipeline): rocprofv3 kernel + memory-copy trace
#   tools/gpu_c3_timeline.sh [option=value ...]      -> gpurun_out/c3_timeline.txt
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tl3; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o p -- python tools/dev/c3_host_only.py "$@" > $O/log.txt 2>&1
{ tail -1 $O/log.txt; python tools/rocpd_timeline.py $(find $O/kt -name "*.db" | head -1) ${NEV:-44}; } > gpurun_out/c3_timeline${TAG}.txt 2>&1
rm -rf $O/kt
cat gpurun_out/c3_timeline${TAG}.txt
