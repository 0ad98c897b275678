# The first two steps of tools/profile_round2.sh only: the bench line and the kernel trace of the same command
# (for a change that leaves the instruction counts as they are).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r02s; rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughputs --no-c4"
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
