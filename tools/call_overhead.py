#!/usr/bin/env python3
"""Host-side cost of one SpMM call on a tiny matrix (the GPU work is negligible): the torch.ops.magnet
operator of the C++ extension next to a direct ctypes call of the same C-ABI entry point."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd import _lib                                                     # noqa: E402
from magnet_amd.laplacian import build_magnetic_laplacian                       # noqa: E402

dev = torch.device('cuda', 0)
n, f = 256, 16
g = torch.Generator().manual_seed(0)
src = torch.randint(0, n, (2000,), generator=g).to(dev); dst = torch.randint(0, n, (2000,), generator=g).to(dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
x = torch.randn(n, 2, f, device=dev); y = torch.empty_like(x)
iters = 5000


def timed(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / iters * 1e6


def via_ctypes():
    p, s = _lib.ptr, _lib.stream_ptr
    _lib.check(_lib.lib.mh_cspmm_planned(p(lap.row_ptr), p(lap.col_idx), p(lap.val), n, p(x), 2 * f, f, p(y), 2 * f, f, 1.0,
                                         None, 0, 0, 0.0, None, 0, 0, 0.0, None, 0, None, f, None, 0, 0, 512, None, 0, s(dev)))


raw = torch.ops.magnet.cspmm.default
print('python -> MagneticLaplacian.spmm (extension)   %.2f us per call' % timed(lambda: lap.spmm(x, out=y)))
print('python -> torch.ops.magnet.cspmm directly      %.2f us per call' % timed(
    lambda: raw(lap.row_ptr, lap.col_idx, lap.val, n, x, y, 1.0, None, 0.0, None, 0.0, None, False, None, None, 0, 0, 512, None)))
print('python -> ctypes mh_cspmm_planned              %.2f us per call' % timed(via_ctypes))
print('torch.mm of the same size for scale            %.2f us per call' % timed(lambda: torch.mm(x.view(2 * n, f), x.view(2 * n, f).t())))
