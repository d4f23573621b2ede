#!/bin/bash
# GPU box: the GPU test suite, then the per-kernel table of the config-4 epoch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r03/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/ep_single -- python3 tools/epoch_profile.py > gpurun_out/r03/ep_single.log 2>&1 || { tail -5 gpurun_out/r03/ep_single.log; exit 1; }
tail -2 gpurun_out/r03/ep_single.log
timeout -k 10 120 python3 tools/epoch_profile.py --gather-loss > gpurun_out/r03/ep_gather.log 2>&1 && tail -1 gpurun_out/r03/ep_gather.log
