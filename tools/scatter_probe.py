#!/usr/bin/env python3
"""What the data-dependent part of the radix scatter costs: the same number of records sorted when all edges are
the same pair (every tile is one run: contiguous LDS placement and one contiguous 64 KiB write per tile) against
the power-law graph (256 runs of about 256 bytes per tile, bank conflicts in the digit-sorted placement).  Run under
`rocprofv3 --kernel-trace --stats` and compare the scatter rows of the two builds (tools/builder_table.py cuts the last)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
n, e = 10_000_000, 200_000_000
kind = sys.argv[1] if len(sys.argv) > 1 else 'same'
if kind == 'same':
    src = torch.zeros(e, dtype=torch.int32, device='cuda'); dst = torch.ones(e, dtype=torch.int32, device='cuda')
elif kind == 'sorted':      # distinct keys already in order: digits change slowly, runs are long
    src = (torch.arange(e, device='cuda') // 20).to(torch.int32); dst = (torch.arange(e, device='cuda') % n).to(torch.int32)
else:
    src, dst = powerlaw_edges(n, e, 43, device='cuda')
for _ in range(3):
    L = build_magnetic_laplacian(src, dst, n, q=0.25)
torch.cuda.synchronize()
print(kind, 'nnz', L.nnz)
