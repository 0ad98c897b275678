# Texture-addresser occupancy of the tile kernel (are the scattered table probes the bound?).
# (the *_sum counters of the TA block hung rocprofv3 on this pool -- only the averaged BUSY counter is taken)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
d=$R/gpurun_out/ta_1; rm -rf $d; mkdir -p $d
(cd $R && timeout 300 rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --kernel-trace -d $d -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d/log.txt 2>&1); echo "rc=$?"
(cd $R && python tools/pmc_summary.py $(find $d -name "*.db" | head -1) | grep -E "k_pretok")
