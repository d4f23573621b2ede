#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/gpu_tests2.log 2>&1
rc=$?
tail -5 gpurun_out/r03/gpu_tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/ep_single2 -- python3 tools/epoch_profile.py > gpurun_out/r03/ep_single2.log 2>&1 || { tail -5 gpurun_out/r03/ep_single2.log; exit 1; }
grep "^epoch" gpurun_out/r03/ep_single2.log
timeout -k 10 120 python3 tools/epoch_profile.py > gpurun_out/r03/ep_plain2.log 2>&1 && grep "^epoch" gpurun_out/r03/ep_plain2.log
timeout -k 10 500 python3 bench.py > gpurun_out/r03/bench_default_a.json 2> gpurun_out/r03/bench_default_a.err || { tail -5 gpurun_out/r03/bench_default_a.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_default_a.json'))
print('value %.4g ms/step %.1f frac %.4f builder %s epoch %s cpu %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['builder_roofline'], d['epoch'], d['cpu_baseline'].get('value')))"
