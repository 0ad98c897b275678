cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rx_final2.txt; rm -f $O
timeout 300 python tools/dev/rx_time.py final 2>&1 | grep "^\[" >> $O
cd /tmp && export TMPDIR=/tmp
for pat in gpt2 tk_cl100k tk_o200k; do
d=$R/gpurun_out/rxp_kt; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats -d $d -o p -- python $R/tools/dev/rx_time1.py $pat c2 50 > $d/log.txt 2>&1
echo "== kernel trace, $pat on the C2 batch (50 calls of spl_split_device)" >> $O
python $R/tools/rocpd_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "kernel  |k_rx_|fillBuffer" >> $O
rm -rf $d
done
for pat in gpt2; do
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); d=$R/gpurun_out/rxp_$i; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python $R/tools/dev/rx_time1.py $pat c2 10 > $d/log.txt 2>&1
  echo "== $pat pass $i ($c) rc=$?" >> $O
  python $R/tools/pmc_summary.py $(find $d -name "*.db" | head -1) 2>&1 | grep -E "k_rx_match" >> $O
  rm -rf $d
done
done
cat $O
