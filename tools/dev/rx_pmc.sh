#!/bin/bash
# On the GPU box: instruction mix and lane utilisation of the device splitter's kernels, one call per pattern (tools/dev/rx_time.py once).
#   tools/dev/rx_pmc.sh <label> [lib path]      -> gpurun_out/rx_pmc_<label>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; [ -n "$2" ] && export SPL_LIB_PATH=$R/$2
out=$R/gpurun_out/rx_pmc_$L.txt; : > $out
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); d=$R/gpurun_out/rx_pmc_${L}_$i; rm -rf $d; mkdir -p $d
  (cd $R && RX_TIME_QUICK=${RX_TIME_QUICK:-1} timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python tools/dev/rx_time.py once > $d/log.txt 2>&1)
  for k in k_rxw_walk k_rxw_mark k_rx_match k_rx_mark; do (cd $R && python tools/pmc_summary.py $(find $d -name "*.db" | head -1) --kernel $k >> $out 2>&1); done
  rm -rf $d
done
echo "[$L]"; cat $out
