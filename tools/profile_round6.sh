# Round-6 profile set (GPU box): the bench line, the rocprofv3 kernel trace of the SAME command, the PMC passes (one
# counter group per run, --kernel-trace only -- never combined with sys / hip traces), the JSON summaries bench.py reads,
# and the same L2 / fetch counters for the lexically wide rotation (--corpus c2_wide).  Everything lands in gpurun_out/r06/;
# copy what is to be judged into profiles/ (gpurun only merges gpurun_out/ back).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c4 --no-c5 --no-c2-wide"
timeout 2000 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c4 --no-c5 --no-c2-wide --regions 1"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$O/pmc_$i; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py $Q > $d/log.txt 2>&1
  echo "== pass $i ($c) rc=$?" >> $O/pmc.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "k_pretok|k_tile_out" >> $O/pmc.txt
done
python tools/pmc_to_json.py $O/pmc.txt $O/kernel_stats.txt $O
# the wide rotation: kernel trace + the memory-side counters + the instruction counts
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktw -o p -- $CMD --corpus c2_wide > $O/ktw.log 2>&1
python tools/rocpd_summary.py $(find $O/ktw -name "*.db" | head -1) > $O/kernel_stats_c2_wide.txt 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1)); d=$O/pmcw_$i; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py $Q --corpus c2_wide > $d/log.txt 2>&1
  echo "== c2_wide pass $i ($c) rc=$?" >> $O/c2_wide_pmc.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "k_pretok|k_tile_out" >> $O/c2_wide_pmc.txt
done
rm -rf $O/kt $O/ktw $O/pmc_* $O/pmcw_*
ls -la $O
