#!/usr/bin/env python3
"""SpMM roofline sweep over feature widths / graph sizes, and epoch times of the small configs.
Run on the GPU box:  python tools/spmm_sweep.py [--quick]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges   # noqa: E402
from magnet_amd.layer.sparse_magnet import ChebNet, ChebNet_Edge            # noqa: E402
from magnet_amd.utils.hermitian import ChebBasis                            # noqa: E402
from magnet_amd.train import node_epoch, link_epoch                         # noqa: E402

dev = 'cuda'


def alg_bytes(nnz, n, f, addends):
    return nnz * 12 + 4 * (n + 1) + nnz * 8 * f + n * 8 * f * (1 + addends)


def time_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def sweep(n, e, widths, seed=42):
    src, dst = powerlaw_edges(n, e, seed, device=dev)
    L = build_magnetic_laplacian(src, dst, n, q=0.25)
    rows = []
    for f in widths:
        x = torch.randn(n, 2, f, device=dev); z = torch.randn(n, 2, f, device=dev); y = torch.empty_like(x)
        ms = time_ms(lambda: L.spmm(x, out=y, alpha=2.0, z1=z, beta1=-1.0))
        gbs = alg_bytes(L.nnz, n, f, 1) / ms / 1e6
        rows.append(dict(n=n, e=e, nnz=L.nnz, F=f, ms=round(ms, 3), GBs=round(gbs), frac=round(gbs / 8000, 3)))
        print(json.dumps(rows[-1]), flush=True)
        del x, z, y
    return rows


def epochs():
    # config 2: cyclic-DSBM sized graph (N=2500, E~187k), X [N,1], K=2, F=16, 2 layers, 5 classes
    for name, n, e, cin, K, Fh, ncls in [('C2-like', 2500, 187000, 1, 2, 16, 5), ('C1-like', 2995, 8416, 2879, 1, 16, 7),
                                         ('C4', 2_000_000, 40_000_000, 128, 2, 128, 5)]:
        src, dst = powerlaw_edges(n, e, 10, device=dev)
        L = build_magnetic_laplacian(src, dst, n, q=0.1)
        net = ChebNet(cin, ChebBasis(L, K), None, num_filter=Fh, K=K, label_dim=ncls, activation=True, layer=2).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
        X = torch.randn(n, cin, device=dev); y = torch.randint(0, ncls, (1, n), device=dev)
        tr = torch.rand(n, device=dev) < 0.6
        ms = time_ms(lambda: node_epoch(net, opt, X, X, y, tr, ~tr), iters=10 if n > 100000 else 50)
        print(json.dumps(dict(config=name, n=n, e=e, nnz=L.nnz, epoch_ms=round(ms, 3))), flush=True)
    # config 3: CiteSeer sized link prediction
    n, e, K, Fh = 3312, 4004, 2, 16
    src, dst = powerlaw_edges(n, e, 11, device=dev)
    L = build_magnetic_laplacian(src, dst, n, q=0.1)
    net = ChebNet_Edge(2, ChebBasis(L, K), None, num_filter=Fh, K=K, label_dim=3, activation=True, layer=2).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
    X = torch.randn(n, 2, device=dev)
    pt = torch.randint(0, n, (7345, 2), device=dev); yt = torch.randint(0, 3, (7345,), device=dev)
    pv = torch.randint(0, n, (382, 2), device=dev); yv = torch.randint(0, 3, (382,), device=dev)
    ms = time_ms(lambda: link_epoch(net, opt, X, X, pt, yt, pv, yv), iters=50)
    print(json.dumps(dict(config='C3-like', n=n, e=e, nnz=L.nnz, epoch_ms=round(ms, 3))), flush=True)


if __name__ == '__main__':
    quick = '--quick' in sys.argv
    if '--epochs' in sys.argv or not quick:
        epochs()
    sweep(2_000_000, 40_000_000, [4, 8, 16, 32, 64, 128, 256, 512] if not quick else [16, 128])
    if not quick:
        sweep(200_000, 4_000_000, [16, 64, 128, 256])
        sweep(2500, 187000, [1, 16, 64])
