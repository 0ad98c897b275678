# usage: bash tools/gpu_ab.sh variant1 variant2 ...   (each measured twice, interleaved; results in gpurun_out/ab.log)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2; do for v in "$@"; do
  cp _ab/lib_$v.so splintr_amd/libsplintr_hip.so; touch splintr_amd/libsplintr_hip.so
  timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-throughputs --no-c4 2> gpurun_out/ab_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_us'])" >> gpurun_out/ab.log 2>&1
done; done
cp _ab/lib_default.so splintr_amd/libsplintr_hip.so
