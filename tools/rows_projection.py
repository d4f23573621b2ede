#!/usr/bin/env python3
"""Per-rank compute of the ROW partition, measured on ONE GPU: for W = 2, 4, 8 the first and the last rank's share of a
graph is set up exactly as that rank would hold it (its row block, the halo plan, columns renumbered to the compact
[own | halo] buffer, the block split into own-column and halo-column parts) and the K propagation orders are timed
with the transfers switched off.  Against the single-GPU time / W this shows what the split into two SpMM passes and
the block's share of long rows cost; the exchange itself needs several GPUs and is not part of the figure.

Graphs: (a) planted communities at config-4 size, relabelled with magnet_amd.partition (the case the row layout is
for); (b) the bench's power-law graph at config-4 size (no locality: the halo is nearly every other vertex)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import dist as D, partition as P
from magnet_amd.laplacian import MagneticLaplacian, build_magnetic_laplacian, powerlaw_edges

dev = torch.device('cuda', 0)
make = lambda rp, ci, v, nc: MagneticLaplacian(rp, ci, v, nc)
no_lists = lambda need, rank, world, group: [torch.empty(0, dtype=torch.int32, device=dev)] * world


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def planted(n, c, e_in, e_out, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    comm = torch.randint(0, c, (n,), device=dev, generator=g)
    order = torch.argsort(comm, stable=True)
    counts = torch.bincount(comm, minlength=c); starts = torch.cumsum(counts, 0) - counts
    s = torch.randint(0, n, (e_in,), device=dev, generator=g)
    r = torch.rand(e_in, device=dev, generator=g)
    d = order[starts[comm[s]] + (r * counts[comm[s]]).long().clamp(max=n - 1)]
    return (torch.cat([s, torch.randint(0, n, (e_out,), device=dev, generator=g)]),
            torch.cat([d, torch.randint(0, n, (e_out,), device=dev, generator=g)]))


def project(name, lap, n, f, K, bounds_for):
    x = torch.randn(n, 2, f, device=dev)
    zs = [torch.empty_like(x) for _ in range(K)]

    def full():
        lap.spmm(x, out=zs[0])
        for k in range(1, K):
            lap.spmm(zs[k - 1], out=zs[k], alpha=2.0, z1=x if k == 1 else zs[k - 2], beta1=-1.0)
    t_full = timeit(full)
    print('%s: N=%d nnz=%d F=%d K=%d: single GPU %.2f ms for the K orders' % (name, n, lap.nnz, f, K, t_full), flush=True)
    del zs
    for world in (2, 4, 8):
        bounds = bounds_for(world)
        for rank in (0, world - 1):
            rp, ci, v = D.slice_rows(lap.row_ptr, lap.col_idx, lap.val, bounds[rank], bounds[rank + 1])
            part = D.RowPartition(rp, ci, v, n, bounds, rank, world, make, exchange_lists=no_lists)
            part.comm_enabled = False
            xb = part.new_buffer(f, dev); xb.normal_()
            bufs = [part.new_buffer(f, dev) for _ in range(K)]
            for b in bufs:
                b.normal_()                    # halo slices hold finite numbers (nothing arrives with the transfers off)
            t = timeit(lambda: part.cheb_terms(xb, K + 1, bufs))
            print('  W=%d rank %d: rows %d (%.1f%%) nnz %d (%.1f%%) halo rows %d, own-column share of nnz %.2f: %.2f ms = %.2f x (single/W)' % (
                world, rank, part.n_local, 100.0 * part.n_local / n, part.local_nnz, 100.0 * part.local_nnz / lap.nnz, part.n_halo,
                part.L_int.nnz / max(part.local_nnz, 1), t, t / (t_full / world)), flush=True)
            del part, xb, bufs
    del x


n, e, f, K = 2_000_000, 40_000_000, 128, 2
src, dst = planted(n, 8, 36_000_000, 4_000_000, 1)
orders = {}
for world in (2, 4, 8):
    parts = P.balanced_label_propagation(src, dst, n, world)
    orders[world] = P.order_from_parts(parts, world)
# one numbering for all W: the 8-part order (its parts nest in the coarser cuts only approximately, so the bounds of
# W = 2, 4 are taken by work on that numbering)
new_of_old, _, b8 = orders[8]
lap = build_magnetic_laplacian(new_of_old[src], new_of_old[dst], n, q=0.25)
project('planted communities, relabelled', lap, n, f, K,
        lambda w: b8 if w == 8 else D.partition_bounds(lap.row_ptr, w))
del lap, src, dst
src, dst = powerlaw_edges(n, e, 42, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst
project('power-law (bench generator)', lap, n, f, K, lambda w: D.partition_bounds(lap.row_ptr, w))
