#!/usr/bin/env python3
"""Laplacian builder throughput (edges/s) at a few sizes: every build timed by HIP events, minimum and median of 10.
MAGNET_ROOT selects another copy of the package (A/B runs of two library builds on one GPU)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get('MAGNET_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
cases = [(2_000_000, 40_000_000, False), (10_000_000, 200_000_000, False), (2_000_000, 40_000_000, True), (100_000, 2_000_000, False)]
if os.environ.get('ONLY_C5'):
    cases = cases[1:2]
for n, e, weighted in cases:
    src, dst = powerlaw_edges(n, e, 42, device='cuda')
    w = (torch.rand(e, device='cuda') * 4).floor() + 1 if weighted else None
    for _ in range(2):
        L = build_magnetic_laplacian(src, dst, n, q=0.25, edge_weight=w)
    times = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L = build_magnetic_laplacian(src, dst, n, q=0.25, edge_weight=w)
        b.record(); b.synchronize()
        times.append(a.elapsed_time(b))
    times.sort()
    dt = times[len(times) // 2] * 1e-3
    print('sort=own n=%d e=%d weighted=%s: median %.2f ms (min %.2f, max %.2f)  %.2f G edges/s  nnz=%d' % (
        n, e, weighted, dt * 1e3, times[0], times[-1], e / dt / 1e9, L.nnz), flush=True)
    del src, dst, L, w
    torch.cuda.empty_cache()
