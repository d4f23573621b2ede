#!/usr/bin/env python3
"""Per-kernel table of ONE training epoch from a rocprofv3 kernel trace of tools/epoch_profile.py:

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/epoch_profile.py
    python3 tools/epoch_table.py DIR/*/*_kernel_trace.csv > profiles/<tag>_epoch_kernels.txt

The last epoch of the run is cut out of the trace (from one input pack to the next) and its launches are summed by
kernel; the rows add up to the epoch."""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    if name.startswith('Cijk'):
        return 'hipBLASLt fp32 GEMM, macro tile ' + name.split('_MT')[1].split('_')[0]
    n = name.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(?:<[^(]*?>)?)\(', n)
    n = m.group(1) if m else n
    if 'elementwise_kernel_manual_unroll' in name or 'direct_copy' in name:
        return 'torch copy (direct_copy_kernel)'
    if 'reduce_kernel' in name:
        return 'torch reduce_kernel (' + ('max' if 'MaxOps' in name else 'sum') + ')'
    if 'multi_tensor_apply' in name:
        return 'torch multi_tensor_apply (Adam)'
    if 'vectorized_elementwise_kernel' in name or 'unrolled_elementwise_kernel' in name:
        return 'torch elementwise (small)'
    return n[:100]


def main(path, packs_per_epoch=2, marker='pack_kernel'):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    packs = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    # an epoch packs the input twice (training forward, evaluation forward; once with the shared forward of
    # tools/epoch_profile.py --shared: pass 1 as the second argument; when the first layer's orders are kept nothing is
    # packed: pass a kernel that runs once per epoch as the third argument, e.g. nll_loss2d_backward — the window then
    # runs from one training backward to the next, which is one epoch's launches all the same): the last-but-one epoch is cut out
    if len(packs) < 2 * packs_per_epoch:
        raise SystemExit('the trace holds fewer than two epochs')
    win = rows[packs[-2 * packs_per_epoch]:packs[-packs_per_epoch]]
    agg = OrderedDict()
    for r in win:
        a = agg.setdefault(short(r['Kernel_Name']), [0, 0.0])
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot = sum(v[1] for v in agg.values())
    span = (int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])) / 1e6
    print('one epoch (train forward + backward + Adam, evaluation forward): %d launches, %.2f ms of kernel time over a span of '
          '%.2f ms' % (len(win), tot, span))
    print('%-100s %6s %10s %7s' % ('kernel', 'calls', 'ms', 'share'))
    small = [0, 0.0]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if v[1] < 0.05:
            small[0] += v[0]; small[1] += v[1]
            continue
        print('%-100s %6d %10.3f %6.1f%%' % (k, v[0], v[1], 100 * v[1] / tot))
    print('%-100s %6d %10.3f %6.1f%%' % ('(kernels below 0.05 ms in total)', small[0], small[1], 100 * small[1] / tot))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, sys.argv[3] if len(sys.argv) > 3 else 'pack_kernel')
