#!/bin/bash
# GPU box: builder of the working tree (B) against the copy under ab/A, same GPU, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/run
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "builder or row_blocks or generator or full_size" > gpurun_out/run/builder_tests.log 2>&1
rc=$?
tail -4 gpurun_out/run/builder_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  echo "== A"; MAGNET_ROOT=$PWD/ab/A timeout -k 10 150 python3 tools/builder_bench.py 2>&1 | grep "sort=" || exit 1
  echo "== B"; timeout -k 10 150 python3 tools/builder_bench.py 2>&1 | grep "sort=" || exit 1
done
for v in A B; do
  if [ $v = A ]; then export MAGNET_ROOT=$PWD/ab/A; else unset MAGNET_ROOT; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/run/ab_$v -- python3 tools/builder_trace.py > gpurun_out/run/ab_$v.log 2>&1 || { tail -5 gpurun_out/run/ab_$v.log; exit 1; }
  echo "== $v kernels"
  python3 - $v <<'PY'
import csv, glob, sys
f=sorted(glob.glob('gpurun_out/run/ab_%s/*/*_kernel_stats.csv'%sys.argv[1]))[-1]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    if 'powerlaw' in n: continue
    tot+=int(r['TotalDurationNs'])
    if int(r['TotalDurationNs'])>2e5: print('%-64s %4s calls %9.3f ms avg %8.3f' % (n[:64], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
print('kernel time per build: %.2f ms' % (tot/4e6))
PY
done
