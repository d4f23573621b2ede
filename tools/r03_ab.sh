#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
for rep in 1 2; do
for v in $VARIANTS; do
  echo "== $ENVNAME=$v" 
  env $ENVNAME=$v timeout -k 10 120 python3 tools/builder_bench.py 2>&1 | grep "sort=" | head -2 || exit 1
done
done
