#!/bin/bash
# GPU box: the round's evidence files (copied into profiles/ afterwards)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/run/collect
mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep -- python3 tools/epoch_profile.py > $O/ep.log 2>&1 || { tail -5 $O/ep.log; exit 1; }
grep "^epoch" $O/ep.log
timeout -k 10 120 python3 tools/epoch_profile.py > $O/ep_plain.log 2>&1 && grep "^epoch" $O/ep_plain.log
timeout -k 10 120 python3 tools/epoch_profile.py --cached > $O/ep_cached.log 2>&1 && grep "^epoch" $O/ep_cached.log
timeout -k 10 200 python3 tools/builder_bench.py > $O/builder_bench.txt 2>&1 || { tail -5 $O/builder_bench.txt; exit 1; }
grep sort= $O/builder_bench.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bt -- python3 tools/builder_trace.py > $O/bt.log 2>&1 || { tail -5 $O/bt.log; exit 1; }
echo "trace done"
