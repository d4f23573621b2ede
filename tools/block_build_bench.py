"""One rank's row block of the power-law graph built without and with every node's record count (mh_lap_block_count_tagged:
the columns' degrees travel in the sort keys), whole graph as one block and one block of eight, config-4 and config-5 size."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from magnet_amd import dist as D
from magnet_amd.laplacian import powerlaw_edges, LaplacianBlockBuild, build_magnetic_laplacian
dev = torch.device('cuda', 0)
for n, e in ((2_000_000, 40_000_000), (10_000_000, 200_000_000)):
    src, dst = powerlaw_edges(n, e, 43, device=dev)
    deg = (torch.bincount(src.long(), minlength=n) + torch.bincount(dst.long(), minlength=n)).int()
    for world in (1, 8):
        bounds = D.bounds_from_degrees(deg, world)
        r = world // 2
        r0, r1 = bounds[r], bounds[r + 1]
        fm, bm = (src >= r0) & (src < r1), (dst >= r0) & (dst < r1)
        fwd, bwd = (src[fm], dst[fm]), (src[bm], dst[bm])
        dinv = build_magnetic_laplacian(src, dst, n, q=0.25).dinv if world == 1 else dinv
        for tag in (False, True):
            ts = []
            for it in range(6):
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                b = LaplacianBlockBuild(fwd, bwd, n, r0, r1, node_records=deg if tag else None)
                blk = b.finish(dinv, 0.25)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            print('N=%d E=%d world=%d rank %d block (%d records): %s %.2f ms (median of the last 5)' % (
                n, e, world, r, fwd[0].numel() + bwd[0].numel(), 'tagged  ' if tag else 'untagged', sorted(ts[1:])[2]), flush=True)
    del src, dst, deg
