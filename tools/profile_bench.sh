#!/bin/bash
# Runs on the GPU box (through gpurun): per-kernel time of the default bench.py workload, then
# HBM read / write counters in separate passes.  Summaries land in gpurun_out/; copy the ones to be
# kept into profiles/ with tools/summarize_profile.py.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="${BENCH_ARGS:---no-cpu-baseline}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_stats.json" 2> "$OUT/bench_stats.err" || { echo "stats pass failed"; tail -5 "$OUT/bench_stats.err"; exit 1; }
echo "stats pass done"
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS --no-epoch --steps 1 --warmup 0 > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/bench_fetch.err"; exit 1; }
echo "fetch pass done"
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS --no-epoch --steps 1 --warmup 0 > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || { echo "write pass failed"; tail -5 "$OUT/bench_write.err"; exit 1; }
echo "write pass done"
find "$OUT" -name "*.csv" | head -20
