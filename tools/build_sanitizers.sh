#!/bin/bash
# ThreadSanitizer build of the library's HOST side (device code unchanged) and of the host-pipeline driver, into
# tests/san/_build/ (git-ignored; travels to the GPU box with a gpurun snapshot).  tests/test_gpu_sanitizers.py runs it.
set -e
cd "$(dirname "$0")/.."
B=tests/san/_build; mkdir -p $B
C=splintr_amd/csrc
# staleness by CONTENT (a modification time proves nothing about a file that travelled or was checked out)
H=$(cat $C/* include/splintr_hip.h tests/san/hostpath_driver.cpp | sha256sum | cut -d' ' -f1)
if [ ! -f $B/libsplintr_hip_tsan.so ] || [ "$(cat $B/tsan.srchash 2>/dev/null)" != "$H" ]; then
  # a failed build fails the script (and with it the test): no filter that could swallow the compiler's exit code
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -Wno-unused-value -Xarch_host -fsanitize=thread \
        -o $B/libsplintr_hip_tsan.so $C/spl_api.hip $C/spl_tables.cpp $C/spl_regex.cpp > $B/tsan_build.log 2>&1 \
        || { tail -40 $B/tsan_build.log; exit 1; }
  echo "$H" > $B/tsan.srchash
fi
hipcc -O1 -g -std=c++17 -fsanitize=thread -x c++ tests/san/hostpath_driver.cpp -o $B/hostpath_tsan -L$B -lsplintr_hip_tsan -Wl,-rpath,'$ORIGIN' -pthread
ls -la $B
