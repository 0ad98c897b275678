# VALU/SALU/LDS/VMEM instruction counts of k_pretok cut off after each phase (differences = per-phase mix)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for ph in 1 2 3 4 5 6 7 0; do
  d=$R/gpurun_out/mix_$ph; rm -rf $d; mkdir -p $d
  (cd $R && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $d -o p -- python tools/dev/gpu_stop_phase.py $ph > $d/log.txt 2>&1); echo "stop after phase $ph rc=$?"
  (cd $R && python tools/pmc_summary.py $(find $d -name "*.db" | head -1) --kernel k_pretok)
done
