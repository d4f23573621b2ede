#!/bin/bash
# GPU box: whole GPU test suite, then the fuzzers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/run
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/run/gpu_tests3.log 2>&1
rc=$?
tail -4 gpurun_out/run/gpu_tests3.log
[ $rc -eq 0 ] || exit $rc
for kind in graph dist aux spmm; do
  timeout -k 10 400 python3 -u tests/fuzz_parity.py --kind $kind --cases ${CASES:-300} --seed ${SEED:-31} > gpurun_out/run/fuzz_$kind.log 2>&1
  echo "fuzz $kind rc=$? $(grep -c FAILED gpurun_out/run/fuzz_$kind.log) failed; $(tail -1 gpurun_out/run/fuzz_$kind.log)"
done
