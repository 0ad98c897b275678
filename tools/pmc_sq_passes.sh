# Counter passes over the bench batch (one rocprofv3 run per group; --kernel-trace only).
#   bash tools/pmc_sq_passes.sh "<group 1>" "<group 2>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for c in "$@"; do
  i=$((i+1)); d=$R/gpurun_out/pmcx_$i; rm -rf $d; mkdir -p $d
  (cd $R && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d/log.txt 2>&1); echo "pass $i rc=$?"
  (cd $R && python tools/pmc_summary.py $(find $d -name "*.db" | head -1) --kernel ${PMC_KERNEL:-k_pretok})
done
