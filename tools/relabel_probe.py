"""Set-up cost and effect of magnet_amd.partition on the GPU: a planted-community digraph with random vertex ids at
config-4 size (2M vertices, 40M edges), 8 parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import partition as P

dev = torch.device('cuda', 0)
n, c, e_in, e_out = 2_000_000, 8, 36_000_000, 4_000_000
g = torch.Generator(device=dev).manual_seed(1)
comm = torch.randint(0, c, (n,), device=dev, generator=g)
order = torch.argsort(comm, stable=True)
counts = torch.bincount(comm, minlength=c); starts = torch.cumsum(counts, 0) - counts
s = torch.randint(0, n, (e_in,), device=dev, generator=g)
r = torch.rand(e_in, device=dev, generator=g)
d = order[starts[comm[s]] + (r * counts[comm[s]]).long().clamp(max=n - 1)]
src = torch.cat([s, torch.randint(0, n, (e_out,), device=dev, generator=g)])
dst = torch.cat([d, torch.randint(0, n, (e_out,), device=dev, generator=g)])
torch.cuda.synchronize()
t0 = time.time()
parts = P.balanced_label_propagation(src, dst, n, c)
torch.cuda.synchronize()
dt = time.time() - t0
w = torch.bincount(torch.cat([src, dst]), minlength=n).double() + 2
load = torch.zeros(c, dtype=torch.float64, device=dev).index_add_(0, parts, w)
blocks = torch.arange(n, device=dev) * c // n
print('N=%d E=%d parts=%d: %.2f s; cut fraction: id blocks %.3f, found %.3f, planted %.3f; load max/mean %.3f; peak %.1f GB' % (
    n, src.numel(), c, dt, P.cut_fraction(src, dst, blocks), P.cut_fraction(src, dst, parts), P.cut_fraction(src, dst, comm),
    float(load.max() / load.mean()), torch.cuda.max_memory_allocated() / 1e9))
