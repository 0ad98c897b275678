#!/bin/bash
# ThreadSanitizer build of the library's HOST side (device code unchanged) and of the host-pipeline driver, into
# tests/san/_build/ (git-ignored; travels to the GPU box with a gpurun snapshot).  tests/test_gpu_sanitizers.py runs it.
set -e
cd "$(dirname "$0")/.."
B=tests/san/_build; mkdir -p $B
C=splintr_amd/csrc
if [ ! -f $B/libsplintr_hip_tsan.so ] || [ -n "$(find $C include -newer $B/libsplintr_hip_tsan.so -type f | head -1)" ]; then
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -Wno-unused-value -Xarch_host -fsanitize=thread \
        -o $B/libsplintr_hip_tsan.so $C/spl_api.hip $C/spl_tables.cpp $C/spl_regex.cpp 2>&1 | grep -E " error" || true
fi
hipcc -O1 -g -std=c++17 -fsanitize=thread -x c++ tests/san/hostpath_driver.cpp -o $B/hostpath_tsan -L$B -lsplintr_hip_tsan -Wl,-rpath,'$ORIGIN' -pthread
ls -la $B
