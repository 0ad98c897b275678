cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05; mkdir -p $O
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c4 --no-c5 --no-c2-wide --regions 1"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1)); d=$O/pmcw_$i; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py $Q --corpus c2_wide > $d/log.txt 2>&1
  echo "== c2_wide pass $i ($c) rc=$?" >> $O/c2_wide_pmc.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "k_pretok|k_tile_out" >> $O/c2_wide_pmc.txt
done
rm -rf $O/pmcw_*
cat $O/c2_wide_pmc.txt
