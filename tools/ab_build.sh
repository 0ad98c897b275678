#!/bin/bash
# Build A/B variants of the library into _ab/:  tools/ab_build.sh name "-DFLAG=1 ..." [name2 "flags2" ...]
# (_ab/ is scratch: git-ignored; it ships to the GPU box with a gpurun call, so delete it when the comparison is done)
set -e
cd "$(dirname "$0")/.."
mkdir -p ${AB_OUT:-_ab}
build() { hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -std=c++17 -fPIC -shared -Wno-unused-value $2 -o ${AB_OUT:-_ab}/lib_$1.so splintr_amd/csrc/spl_api.hip splintr_amd/csrc/spl_tables.cpp splintr_amd/csrc/spl_regex.cpp 2>&1 | grep -E " error" || true; }
while [ $# -ge 2 ]; do build "$1" "$2" & shift 2; done
wait
ls -la ${AB_OUT:-_ab}/*.so
