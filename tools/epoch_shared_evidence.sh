#!/bin/bash
# GPU box: per-kernel tables of the config-4 epoch with the shared forward (appended to profiles/r03_epoch_kernels.txt by hand)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s6
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep_shared -- python3 tools/epoch_profile.py --shared > $O/ep_shared.log 2>&1 || { tail -5 $O/ep_shared.log; exit 1; }
grep "^epoch" $O/ep_shared.log
python3 tools/epoch_table.py $O/ep_shared/*/*_kernel_trace.csv 1 > $O/epoch_table_shared.txt || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep_shared_cached -- python3 tools/epoch_profile.py --shared --cached > $O/ep_shared_cached.log 2>&1 || { tail -5 $O/ep_shared_cached.log; exit 1; }
grep "^epoch" $O/ep_shared_cached.log
python3 tools/epoch_table.py $O/ep_shared_cached/*/*_kernel_trace.csv 1 nll_loss2d_backward > $O/epoch_table_shared_cached.txt || exit 1
rm -rf $O/ep_shared $O/ep_shared_cached
timeout -k 10 120 python3 tools/epoch_compare.py > $O/epoch_compare.txt 2>&1; tail -4 $O/epoch_compare.txt
