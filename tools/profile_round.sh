cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 500 --warmup 50 > gpurun_out/bench_r01d.json 2> gpurun_out/bench_r01d.err; tail -c 600 gpurun_out/bench_r01d.json
rm -rf gpurun_out/kt_d; timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_d -o p -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/kt_d.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/kt_d -name "*.db" | head -1) > gpurun_out/kt_d_summary.txt; cat gpurun_out/kt_d_summary.txt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$R/gpurun_out/pmc_d_$i; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d/log.txt 2>&1; echo "pass $i ($c) rc=$?"
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) | grep -E "k_pretok|k_tile_out"
done
