#!/usr/bin/env python3
"""SpMM on a graph WITH id locality (every edge joins nodes at most `band` ids apart) next to the
locality-free power-law graph of the same size: algorithmic GB/s of one order."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges

dev = torch.device('cuda', 0)


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(name, src, dst, n, f):
    lap = build_magnetic_laplacian(src, dst, n, q=0.25)
    x = torch.randn(n, 2, f, device=dev); y = torch.empty_like(x)
    ms = timeit(lambda: lap.spmm(x, out=y))
    alg = lap.nnz * (12 + 8 * f) + n * 8 * f
    print('%-34s N=%d nnz=%d F=%d: %.2f ms  %.0f GB/s algorithmic' % (name, n, lap.nnz, f, ms, alg / ms / 1e6), flush=True)


n, e = 2_000_000, 40_000_000
g = torch.Generator(device=dev).manual_seed(5)
for f in (64, 128, 256):
    src, dst = powerlaw_edges(n, e, 42, device=dev)
    run('power-law (no locality)', src, dst, n, f)
    for band in (1000, 100_000):
        s = torch.randint(0, n, (e,), device=dev, generator=g)
        d = (s + torch.randint(-band, band + 1, (e,), device=dev, generator=g)).clamp_(0, n - 1)
        run('banded, |i-j| <= %d' % band, s, d, n, f)
