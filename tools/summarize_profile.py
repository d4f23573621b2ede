#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof) into the summaries kept
under profiles/:  <tag>_kernel_stats.csv (per-kernel time of the bench run), <tag>_hbm_counters.csv
(FETCH_SIZE / WRITE_SIZE per launch of the kernels that matter) and spmm_traffic.json (HBM bytes
per SpMM launch, which bench.py reports as roofline.traffic).

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950
FETCH_SIZE reports half the bytes of 16-byte-per-lane coalesced reads (checked here on two kernels whose
traffic is known exactly: the read-only probe stream_read_kernel and the edge generator powerlaw_kernel).
spmm_traffic.json records a hash of the SpMM kernel source; bench.py refuses the file when the source has changed since.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
src = os.path.join(ROOT, 'gpurun_out', 'prof')
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(src, 'stats', '*', '*_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, tag + '_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        name = r['Name']
        if name.startswith('Cijk_'):
            name = name[:60] + '... (hipBLASLt fp32 GEMM)'
        elif len(name) > 160:
            name = name[:160] + '...'
        w.writerow([name, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])

bench = json.loads(open(os.path.join(src, 'bench_stats.json')).read().strip().splitlines()[-1])
per = {}
for kind in ('fetch', 'write'):
    path = newest(os.path.join(src, kind, '*', '*_counter_collection.csv'))
    recs = list(csv.DictReader(open(path)))
    # the counter passes run bench.py with --no-epoch: every launch belongs to the headline workload
    for r in recs:
        n = r['Kernel_Name']
        key = ('cspmm' if 'cspmm_vec4<' in n else 'cspmm_chunks' if 'cspmm_vec4_chunks' in n else 'cspmm_combine' if 'cspmm_vec4_combine' in n
               else 'powerlaw' if 'powerlaw_kernel' in n else 'stream_read' if 'stream_read_kernel' in n
               else 'gemm' if n.startswith('Cijk_') else None)
        if key:
            per.setdefault((key, kind), []).append(
                (float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
with open(os.path.join(dst, tag + '_hbm_counters.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'counter', 'launch', 'raw_KiB', 'bytes_corrected', 'duration_ms_under_pmc'])
    for (key, kind), vals in sorted(per.items()):
        for i, (v, ms) in enumerate(vals):
            w.writerow([key, 'FETCH_SIZE' if kind == 'fetch' else 'WRITE_SIZE', i, v,
                        v * 1024 * (2 if kind == 'fetch' else 1), '%.3f' % ms])
mean = lambda xs: sum(xs) / len(xs)
fetch = mean([v for v, _ in per[('cspmm', 'fetch')]]) * 1024 * 2
write = mean([v for v, _ in per[('cspmm', 'write')]]) * 1024
# kernels whose traffic is known exactly: the read-only probe of bench.py's ceiling measurement (8 GiB read per launch) and
# the edge generator (8 bytes written per edge, 50M edges per launch, nothing read)
cal_f = mean([v for v, _ in per[('stream_read', 'fetch')]]) * 1024 * 2
cal_w = mean([v for v, _ in per[('powerlaw', 'write')]]) * 1024
for extra in ('cspmm_chunks', 'cspmm_combine'):       # the hub-row kernels belong to the same SpMM order
    if (extra, 'fetch') in per:
        fetch += mean([v for v, _ in per[(extra, 'fetch')]]) * 1024 * 2
        write += mean([v for v, _ in per[(extra, 'write')]]) * 1024
nnz = bench['config']['nnz']
import re
m = re.search(r'N=(\d+) E=(\d+).*?, (\d+) complex features', bench['config']['workload'])
out = {'nodes': int(m.group(1)), 'edges': int(m.group(2)), 'features': int(m.group(3)), 'nnz': nnz,
       'hbm_bytes_per_launch': fetch + write, 'read_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
       'algorithmic_bytes_per_launch': bench['roofline']['algorithmic_bytes_per_launch'],
       'calibration': {'kernels': 'stream_read_kernel (reads 8 GiB per launch, writes nothing) for FETCH_SIZE; powerlaw_kernel '
                                  '(writes 8 bytes per edge, 50M edges per launch) for WRITE_SIZE',
                       'expected_read_bytes': 8 << 30, 'expected_write_bytes': 50_000_000 * 8,
                       'measured_read_corrected': cal_f, 'measured_write': cal_w},
       'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), %s' % tag,
       'traffic_kind': 'L2 <-> fabric requests (TCC_EA): Infinity Cache (MALL) hits are inside this figure',
       'kernel_source_sha256': hashlib.sha256(open(os.path.join(ROOT, 'magnet_amd', 'csrc', 'cspmm.hip'), 'rb').read()).hexdigest()}
json.dump(out, open(os.path.join(dst, 'spmm_traffic.json'), 'w'), indent=1)
open(os.path.join(dst, tag + '_bench_under_rocprof.json'), 'w').write(json.dumps(bench) + '\n')
print(json.dumps(out, indent=1))
