#!/usr/bin/env python3
"""Launch-by-launch table of ONE Laplacian build from a rocprofv3 kernel trace of tools/builder_trace.py:

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/builder_trace.py
    python3 tools/builder_table.py DIR/*/*_kernel_trace.csv > profiles/<tag>_builder_kernels.txt

The last build of the trace is cut out (from the first record-generating histogram of the last sort to the end)."""
import csv
import re
import sys


def short(name):
    n = name.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(?:<[^(]*?>)?)\(', n)
    n = m.group(1) if m else n
    return re.sub(r'<(unsigned )?(int|long)( long)?>', '', n.split('.kd')[0])[:90]


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    firsts = [i for i, r in enumerate(rows) if 'tile_histogram<1' in r['Kernel_Name'] or 'tile_histogram<2' in r['Kernel_Name']]
    if not firsts:
        raise SystemExit('no record-generating histogram in the trace')
    start = firsts[-1]
    while start > 0 and 'fillBuffer' in rows[start - 1]['Kernel_Name']:
        start -= 1
    win = rows[start:]
    t0 = int(win[0]['Start_Timestamp'])
    total = 0.0
    print(' start ms       ms  kernel')
    for r in win:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        total += (e - s) * 1e-6
        print('%9.3f %8.3f  %s' % ((s - t0) * 1e-6, (e - s) * 1e-6, short(r['Kernel_Name'])))
    span = (int(win[-1]['End_Timestamp']) - t0) * 1e-6
    print('kernel time %.2f ms over a span of %.2f ms' % (total, span))


if __name__ == '__main__':
    main(sys.argv[1])
