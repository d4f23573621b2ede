#!/usr/bin/env python3
"""Builder rate when the boundary hands over HOST buffers (the drop-in hermitian_decomp_sparse:
numpy in, scipy CSR out) against device-resident inputs/outputs, config-4 sized graph."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
from magnet_amd.utils.hermitian import hermitian_decomp_sparse
n, e = 2_000_000, 40_000_000
src, dst = powerlaw_edges(n, e, 42, device='cuda')
hs, hd = src.cpu().numpy().astype(np.int64), dst.cpu().numpy().astype(np.int64)
for _ in range(2):
    build_magnetic_laplacian(src, dst, n, q=0.25)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    L = build_magnetic_laplacian(src, dst, n, q=0.25)
torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / 5
hermitian_decomp_sparse(hs, hd, n, 0.25)
t0 = time.perf_counter()
for _ in range(3):
    S = hermitian_decomp_sparse(hs, hd, n, 0.25)
t_host = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    Ld = build_magnetic_laplacian(hs, hd, n, q=0.25); torch.cuda.synchronize()
t_in = (time.perf_counter() - t0) / 3
print('device in/out: %.1f ms (%.2f G edges/s); host int64 arrays in, device out: %.1f ms (%.2f G edges/s); '
      'host in, scipy CSR out: %.1f ms (%.3f G edges/s); nnz=%d' % (t_dev * 1e3, e / t_dev / 1e9, t_in * 1e3, e / t_in / 1e9,
                                                                t_host * 1e3, e / t_host / 1e9, S.nnz))
