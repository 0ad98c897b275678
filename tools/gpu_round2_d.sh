cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_streaming.py tests/test_gpu_parity.py -x -q -m gpu -k "streaming or reference_test_strings or mistral_v3_special" --timeout=300 --timeout-method=thread > gpurun_out/t_new.log 2>&1
echo "rc=$?" >> gpurun_out/t_new.log
# A/B: segment search for ASCII medium chunks
cp splintr_amd/libsplintr_hip.so _ab/lib_default.so
for v in default segascii default segascii; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so
  timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughputs --no-c4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_us'])" >> gpurun_out/ab_seg.log 2>&1
done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
