#!/bin/bash
# On the GPU box: L2 hit / miss and fetch / write size of k_pretok for one library build.
#   tools/pmc_quick.sh <label> [lib path]      -> gpurun_out/pmcq_<label>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; [ -n "$2" ] && export SPL_LIB_PATH=$2
out=$R/gpurun_out/pmcq_$L.txt; : > $out
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=$R/gpurun_out/pmcq_${L}_$i; rm -rf $d; mkdir -p $d
  (cd $R && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c4 --no-c5 $BENCH_ARGS > $d/log.txt 2>&1)
  (cd $R && python tools/pmc_summary.py $(find $d -name "*.db" | head -1) --kernel k_pretok >> $out 2>&1)
  rm -rf $d
done
echo "[$L]"; cat $out
