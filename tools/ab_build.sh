#!/bin/bash
# Build kernel variants for A/B runs on the GPU box:  tools/ab_build.sh name "-DFOO=1 ..." [name flags]...
# Libraries land in _ab/<name>.so (git-ignored, shipped by gpurun); select with SPL_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
mkdir -p _ab
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $f -o _ab/$n.so splintr_amd/csrc/spl_api.hip splintr_amd/csrc/spl_tables.cpp &
done
wait
ls -la _ab
