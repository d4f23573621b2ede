#!/bin/bash
# The library and the plain C++ callers built a second time with the HOST side under AddressSanitizer (device code is
# compiled as usual: the GPU sanitizer is not used), then run on the GPU box.  Catches host-side memory errors in the
# workspace carving, plan building and two-phase calls of the C ABI.
#   build (CPU container):  bash tools/host_asan.sh build      -> ab/asan/
#   run   (GPU box):        bash tools/host_asan.sh run        -> gpurun_out/host_asan.txt
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
D=ab/asan
if [ "${1:-build}" = build ]; then
  mkdir -p $D
  for f in lap_build radix_sort cspmm aux halo exchange; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
        -c magnet_amd/csrc/$f.hip -o $D/$f.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address $D/*.o -o $D/libmagnet_hip_asan.so -ldl || exit 1
  for t in abi_client abi_sweep; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -Xarch_host -fsanitize=address -DABI_SWEEP_SKIP_TEARDOWN tests/abi/$t.cpp -o $D/${t}_asan -L$D -lmagnet_hip_asan -Wl,-rpath,'$ORIGIN' || exit 1
  done
  rm -f $D/*.o
  ls -la $D
else
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
  O=gpurun_out/host_asan.txt
  mkdir -p gpurun_out
  : > $O
  ( cd $D
    timeout -k 10 100 ./abi_client_asan 300 5000 8 0.25 > /dev/null 2>> ../../$O; echo "abi_client rc=$?" >> ../../$O
    timeout -k 10 100 ./abi_client_asan 300 5000 8 0.25 exchange 2>> ../../$O | tail -1 >> ../../$O; echo "abi_client exchange rc=${PIPESTATUS[0]}" >> ../../$O
    timeout -k 10 300 ./abi_sweep_asan >> ../../$O 2>&1; echo "abi_sweep rc=$?" >> ../../$O )
  grep -c "ERROR: AddressSanitizer" $O | sed 's/^/AddressSanitizer reports: /' >> $O
  tail -15 $O
fi
