#!/bin/bash
# Round-5 phase instruction mix of k_pretok (GPU box): counter passes with the kernel cut off after each phase
# (-DSPL_DEBUG_STAMPS build in ${AB_DIR:-_abq}/lib_stamps.so), and every workgroup's wall clock at the phase
# boundaries (-DSPL_STAMP_ALL build, lib_stampall.so), for one tile alone and for the full bench batch.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05mix; rm -rf $O; mkdir -p $O
A=$R/${AB_DIR:-_abq}
for ph in 1 2 3 4 5 6 7 0; do
  d=$O/mix_$ph; mkdir -p $d
  SPL_LIB_PATH=$A/lib_stamps.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $d -o p -- python tools/dev/gpu_stop_phase.py $ph > $d/log.txt 2>&1
  echo "== stop after phase $ph rc=$?" >> $O/phase_mix.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) --kernel k_pretok >> $O/phase_mix.txt 2>&1
  rm -rf $d
done
SPL_LIB_PATH=$A/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2 1000 > $O/walls_full.txt 2>&1
SPL_LIB_PATH=$A/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2 1 > $O/walls_one_doc.txt 2>&1
SPL_LIB_PATH=$A/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c2_wide 1000 > $O/walls_wide.txt 2>&1
timeout 300 python tools/dev/gpu_kbench.py base > $O/kbench.txt 2>&1
cat $O/phase_mix.txt $O/walls_*.txt $O/kbench.txt
# round 5: the tail -- phase walls on multi-byte text, the steps of a pass (-DSPL_STAMP_TAIL build), the cut builds
SPL_LIB_PATH=$A/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py c3 250 > $O/walls_c3.txt 2>&1
SPL_LIB_PATH=$A/lib_stampall.so timeout 300 python tools/dev/gpu_phase_walls.py purecjk 250 > $O/walls_purecjk.txt 2>&1
for v in o200k_base cl100k_base deepseek_v3; do
  SPL_LIB_PATH=$A/lib_tailst.so timeout 300 python tools/dev/gpu_tail_steps.py $v purecjk 250 > $O/tail_steps_purecjk_$v.txt 2>&1
done
SPL_LIB_PATH=$A/lib_tailst.so timeout 300 python tools/dev/gpu_tail_steps.py o200k_base c3 250 > $O/tail_steps_c3.txt 2>&1
AB_DIR=${AB_DIR:-_abq} bash tools/dev/run_cfg_ab.sh full cut1 cut2 cut3 notail > $O/tail_cuts.txt 2>&1
cat $O/walls_c3.txt $O/walls_purecjk.txt $O/tail_steps_*.txt $O/tail_cuts.txt
