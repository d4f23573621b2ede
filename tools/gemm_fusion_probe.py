#!/usr/bin/env python3
"""Probe: per-order weight GEMMs on separate order matrices vs one GEMM over the orders laid out side
by side ([N, 2, K+1, F]: each order a strided view the SpMM reads and writes in place)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges

dev = torch.device('cuda', 0)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(n, e, f, K, seed):
    src, dst = powerlaw_edges(n, e, seed, device=dev)
    lap = build_magnetic_laplacian(src, dst, n, q=0.25)
    del src, dst
    k1 = K + 1
    w = torch.randn(k1, f, f, device=dev) / f ** 0.5
    b = torch.zeros(1, f, device=dev)
    x = torch.randn(n, 2, f, device=dev)
    # separate matrices
    zs = [x] + [torch.empty_like(x) for _ in range(K)]
    out = torch.empty(2 * n, f, device=dev)

    def spmm_sep():
        lap.spmm(zs[0], out=zs[1])
        for k in range(2, k1):
            lap.spmm(zs[k - 1], out=zs[k], alpha=2.0, z1=zs[k - 2], beta1=-1.0)

    def gemm_sep():
        torch.addmm(b, zs[0].view(2 * n, f), w[0], out=out)
        for k in range(1, k1):
            out.addmm_(zs[k].view(2 * n, f), w[k])

    t_s, t_g = timeit(spmm_sep), timeit(gemm_sep)
    print('N=%d F=%d K=%d  separate: spmm %.2f ms  gemm %.2f ms' % (n, f, K, t_s, t_g), flush=True)
    del zs
    cat = torch.empty(n, 2, k1, f, device=dev)
    cat[:, :, 0, :].copy_(x)
    zc = [cat[:, :, k, :] for k in range(k1)]
    nxt = torch.empty(n, 2, k1, f, device=dev)
    out_strided = nxt.view(2 * n, k1 * f)[:, :f]

    def spmm_cat():
        lap.spmm(zc[0], out=zc[1])
        for k in range(2, k1):
            lap.spmm(zc[k - 1], out=zc[k], alpha=2.0, z1=zc[k - 2], beta1=-1.0)

    wf = w.reshape(k1 * f, f)
    t_s = timeit(spmm_cat)
    t_g = timeit(lambda: torch.addmm(b, cat.view(2 * n, k1 * f), wf, out=out))
    t_g2 = timeit(lambda: torch.addmm(b, cat.view(2 * n, k1 * f), wf, out=out_strided))
    ref = out.clone()
    gemm_sep_check = torch.addmm(b, zc[0].reshape(2 * n, f), w[0])
    for k in range(1, k1):
        gemm_sep_check.addmm_(zc[k].reshape(2 * n, f), w[k])
    err = float((ref - gemm_sep_check).abs().max() / gemm_sep_check.abs().max())
    err2 = float((out_strided - gemm_sep_check).abs().max() / gemm_sep_check.abs().max())
    print('N=%d F=%d K=%d  side by side: spmm %.2f ms  one gemm %.2f ms (strided out %.2f ms)  rel diff %.1e %.1e'
          % (n, f, K, t_s, t_g, t_g2, err, err2), flush=True)
    # interleaved as a conv runs them: K SpMMs, then the GEMM; SpMM and GEMM parts timed separately
    evs = []
    for _ in range(5):
        a, b_, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); spmm_cat(); b_.record(); torch.addmm(b, cat.view(2 * n, k1 * f), wf, out=out_strided); c.record()
        evs.append((a, b_, c))
    torch.cuda.synchronize()
    print('   interleaved: spmm %s ms, gemm %s ms' % (['%.1f' % x.elapsed_time(y) for x, y, _ in evs],
                                                    ['%.1f' % y.elapsed_time(z) for _, y, z in evs]), flush=True)
    # weight gradient: per order vs one gram
    g = torch.randn(2 * n, f, device=dev)
    t_w1 = timeit(lambda: [torch.mm(zc[k].reshape(2 * n, f).t(), g) for k in range(k1)])
    t_w2 = timeit(lambda: torch.mm(cat.view(2 * n, k1 * f).t(), g))
    print('   dW: per order %.2f ms (strided, copies)  one gram %.2f ms' % (t_w1, t_w2), flush=True)


if __name__ == '__main__':
    run(2_000_000, 40_000_000, 128, 2, 42)
    torch.cuda.empty_cache()
    run(10_000_000, 200_000_000, 256, 3, 43)
