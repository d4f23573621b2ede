#!/usr/bin/env python3
"""Register, LDS and occupancy figures of every kernel of libmagnet_hip.so, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed):

    python3 tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'magnet_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function', '-Rpass-analysis=kernel-resource-usage']


def demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
    clean = []
    for n in out:
        n = n.replace('(anonymous namespace)::', '')
        n = re.sub(r'\(.*$', '', n).replace('void ', '')
        clean.append(n)
    return clean


def main():
    print('kernel resource usage, gfx950, hipcc %s' % ' '.join(FLAGS[:4]))
    print('occupancy = waves per SIMD the register and LDS budget allows (8 is the hardware maximum; a CU has 4 SIMDs, 160 KiB of LDS)')
    print('%-14s %-58s %6s %6s %6s %8s %9s %10s' % ('file', 'kernel', 'VGPRs', 'AGPRs', 'SGPRs', 'scratch', 'LDS bytes', 'waves/SIMD'))
    for f in ['cspmm.hip', 'radix_sort.hip', 'lap_build.hip', 'halo.hip', 'aux.hip']:
        r = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', os.path.join(SRC, f), '-o', '/dev/null'], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-2000:])
        recs, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r'remark:\s+(.*?) \[-Rpass-analysis', line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith('Function Name:'):
                cur = {'name': t.split(':', 1)[1].strip()}
                recs.append(cur)
            elif cur is not None and ':' in t:
                k, v = t.rsplit(':', 1)
                cur[k.strip()] = v.strip()
        names = demangle([r_['name'] for r_ in recs])
        for n, r_ in zip(names, recs):
            print('%-14s %-58s %6s %6s %6s %8s %9s %10s' % (f, n[:58], r_.get('VGPRs'), r_.get('AGPRs'), r_.get('TotalSGPRs'),
                                                           r_.get('ScratchSize [bytes/lane]'), r_.get('LDS Size [bytes/block]'),
                                                           r_.get('Occupancy [waves/SIMD]')))


if __name__ == '__main__':
    main()
