# Round-4 profile set (GPU box): the bench line, the rocprofv3 kernel trace of the SAME command, the PMC passes (one
# counter group per run, --kernel-trace only -- never combined with sys / hip traces), and the JSON summaries bench.py
# reads.  Everything lands in gpurun_out/r04/; copy what is to be judged into profiles/ (tools/profile_round3.sh does
# not touch profiles/ itself: gpurun only merges gpurun_out/ back).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04; rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughputs --no-c4 --no-c5"
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
python tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$O/pmc_$i; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-throughputs --no-c4 --no-c5 > $d/log.txt 2>&1
  echo "== pass $i ($c) rc=$?" >> $O/pmc.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "k_pretok|k_tile_out" >> $O/pmc.txt
done
python tools/pmc_to_json.py $O/pmc.txt $O/kernel_stats.txt $O
rm -rf $O/kt $O/pmc_*
ls -la $O
