#!/bin/bash
# Runs on the GPU box (through gpurun): hardware counters of the SpMM kernel of the default bench workload,
# one rocprofv3 --pmc pass per counter group (cache hits, address translation, read latency, issue).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="--no-cpu-baseline --no-epoch --steps 1 --warmup 0"
run_pass() {
    name="$1"; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" \
        || { echo "$name pass failed"; tail -5 "$OUT/$name.err"; exit 1; }
    echo "$name pass done"
}
run_pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run_pass tlb TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run_pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
