#!/bin/bash
# GPU box: a long randomized parity sweep (tests/fuzz_parity.py), several seeds per kind
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run
for seed in ${SEEDS:-201 202 203}; do
  for kind in graph dist aux spmm; do
    timeout -k 10 600 python3 -u tests/fuzz_parity.py --kind $kind --cases ${CASES:-600} --seed $seed > gpurun_out/run/fuzzc_${kind}_$seed.log 2>&1
    echo "seed $seed $kind rc=$? $(grep -c FAILED gpurun_out/run/fuzzc_${kind}_$seed.log) failed; $(tail -1 gpurun_out/run/fuzzc_${kind}_$seed.log)"
  done
done
