cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; tail -3 gpurun_out/r04_gputests.log
tools/profile_round4.sh > gpurun_out/r04_profile.log 2>&1
AB_DIR=_abx tools/r04_mix.sh > gpurun_out/r04_mix.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench.json'))
t=d['throughputs']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'], 'c4', d['c4_strong']['value'], 'c5', d['c5_strong']['value'], 'c3k', t['c3']['kernel_hbm'], 'c3h', t['c3']['c_abi_host'], 'c2h', t['c2']['c_abi_host'], 'wide', t['c2_wide']['kernel_hbm'], 'custom', t['c2_custom_pattern'].get('c_abi_host'))
PY
head -4 gpurun_out/r04/kernel_stats.txt
