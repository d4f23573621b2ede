#!/usr/bin/env python3
"""Per-rank SpMM time of the column-sharded strategy at config 5, measured on ONE GPU: with N ranks
each rank runs the full CSR on F/N feature columns, so the single-GPU time at width F/N is the
per-rank compute time of an N-GPU run (communication not included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.spmm_sweep import sweep
sweep(10_000_000, 200_000_000, [32, 64, 128, 256], seed=43)

# the last order of a column-sharded conv is computed in N row blocks (each peer's rows leave as soon
# as they exist): cost of the block-wise launches against the single launch, same widths
import torch
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
dev = torch.device('cuda', 0)
n, e = 10_000_000, 200_000_000
src, dst = powerlaw_edges(n, e, 43, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for world in (2, 4, 8):
    f = 256 // world
    x = torch.randn(n, 2, f, device=dev); z1 = torch.randn(n, 2, f, device=dev); y = torch.empty_like(x)
    rb = [n * p // world for p in range(world)] + [n]
    ops = [lap.rows(rb[p], rb[p + 1]) for p in range(world)]

    def blocks():
        for p in range(world):
            ops[p].spmm(x, out=y[rb[p]:rb[p + 1]], alpha=2.0, z1=z1[rb[p]:rb[p + 1]], beta1=-1.0)
    t1 = timeit(lambda: lap.spmm(x, out=y, alpha=2.0, z1=z1, beta1=-1.0))
    t2 = timeit(blocks)
    print('N=%d ranks, F/N=%d: one launch %.2f ms, %d row blocks %.2f ms' % (world, f, t1, world, t2), flush=True)
    del x, z1, y, ops
