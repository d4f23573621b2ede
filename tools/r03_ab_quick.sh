#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for v in ${VARIANTS:-A B}; do
    if [ $v = B ]; then r=$PWD; else r=$PWD/ab/$v; fi
    echo -n "$v: "; MAGNET_ROOT=$r ONLY_C5=${ONLY_C5-1} timeout -k 10 150 python3 tools/builder_bench.py 2>&1 | grep "sort=" || exit 1
  done
done
