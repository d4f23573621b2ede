#!/bin/bash
# GPU box: builder tests, then builder throughput and its per-kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "builder or row_blocks or generator or full_size" > gpurun_out/r03/builder_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r03/builder_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 tools/builder_bench.py > gpurun_out/r03/builder_bench_${TAG:-x}.txt 2>&1 || { tail -5 gpurun_out/r03/builder_bench_${TAG:-x}.txt; exit 1; }
cat gpurun_out/r03/builder_bench_${TAG:-x}.txt | grep sort=
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/bt_${TAG:-x} -- python3 tools/builder_trace.py > gpurun_out/r03/bt_${TAG:-x}.log 2>&1 || { tail -5 gpurun_out/r03/bt_${TAG:-x}.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
tag=os.environ.get('TAG','x')
f=sorted(glob.glob('gpurun_out/r03/bt_%s/*/*_kernel_stats.csv'%tag))[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    print('%-70s %5s calls %9.3f ms avg %8.3f' % (n[:70], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
