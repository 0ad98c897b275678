cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02cal; rm -rf $O; mkdir -p $O
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); d=$O/cal_$i; mkdir -p $d
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- _ab/calib_fetch > $d/log.txt 2>&1
  echo "== calibration pass $i ($c) rc=$?" >> $O/calib.txt
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) >> $O/calib.txt 2>&1
done
grep "known bytes" $O/cal_1/log.txt >> $O/calib.txt
# host path: direct write on / off, C2 and C3
cp /dev/null $O/host.txt
for o in "direct_write=1" "direct_write=0"; do timeout 200 python tools/host_path_bench.py c2 - $o >> $O/host.txt 2>&1; done
timeout 300 python tools/host_path_bench.py c3 >> $O/host.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_hostpath.py -x -q --timeout=300 --timeout-method=thread > $O/t_host.log 2>&1; echo "rc=$?" >> $O/t_host.log
rm -rf $O/cal_*
