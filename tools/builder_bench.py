#!/usr/bin/env python3
"""Laplacian builder throughput (edges/s) at a few sizes; MH_SORT=rocprim selects the library sort for A/B."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
for n, e, weighted in [(2_000_000, 40_000_000, False), (10_000_000, 200_000_000, False), (2_000_000, 40_000_000, True), (100_000, 2_000_000, False)]:
    src, dst = powerlaw_edges(n, e, 42, device='cuda')
    w = (torch.rand(e, device='cuda') * 4).floor() + 1 if weighted else None
    for _ in range(2):
        L = build_magnetic_laplacian(src, dst, n, q=0.25, edge_weight=w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        L = build_magnetic_laplacian(src, dst, n, q=0.25, edge_weight=w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print('sort=%s n=%d e=%d weighted=%s: %.2f ms  %.2f G edges/s  nnz=%d' % (os.environ.get('MH_SORT', 'own'), n, e, weighted, dt * 1e3, e / dt / 1e9, L.nnz), flush=True)
    del src, dst, L, w
    torch.cuda.empty_cache()
