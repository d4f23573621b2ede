#!/usr/bin/env python3
"""Four builds of the config-5 graph (10M vertices, 200M edges) and nothing else, for
`rocprofv3 --kernel-trace --output-format csv -- python3 tools/builder_trace.py`: the kernel trace of the last build is
the builder's per-kernel time table."""
import os, sys
import torch
sys.path.insert(0, os.environ.get('MAGNET_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
n, e = 10_000_000, 200_000_000
src, dst = powerlaw_edges(n, e, 43, device='cuda')
for _ in range(4):
    L = build_magnetic_laplacian(src, dst, n, q=0.25)
torch.cuda.synchronize()
print('nnz', L.nnz)
