#!/usr/bin/env python3
"""Four builds of one graph and nothing else, for
`rocprofv3 --kernel-trace --output-format csv -- python3 tools/builder_trace.py [--weighted] [--nodes N] [--edges E]`:
the kernel trace of the last build is the builder's per-kernel time table (tools/builder_table.py prints it).
Default: the config-5 graph (10M vertices, 200M edges), unweighted."""
import argparse, os, sys
import torch
sys.path.insert(0, os.environ.get('MAGNET_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=10_000_000)
ap.add_argument('--edges', type=int, default=200_000_000)
ap.add_argument('--weighted', action='store_true')
a = ap.parse_args()
src, dst = powerlaw_edges(a.nodes, a.edges, 43, device='cuda')
w = ((src.long() * 7 + dst.long() * 3) % 5 + 1).float() * 0.5 if a.weighted else None
for _ in range(4):
    L = build_magnetic_laplacian(src, dst, a.nodes, q=0.25, edge_weight=w)
torch.cuda.synchronize()
print('nnz', L.nnz)
