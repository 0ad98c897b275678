cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -2
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
t=d['throughputs']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'], 'c4', d['c4_strong']['value'], 'c5', d['c5_strong']['value'], 'c3k', t['c3']['kernel_hbm'], 'c3h', t['c3']['c_abi_host'], 'c2h', t['c2']['c_abi_host'], 'wide', t['c2_wide']['kernel_hbm'])
PY
