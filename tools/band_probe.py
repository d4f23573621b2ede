#!/usr/bin/env python3
"""Time of the three config-5 SpMM orders in this process, with the clocks / power rocm-smi reports while
the kernels run (the run-to-run spread of the headline number is between processes, not inside one)."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
from magnet_amd.layer.sparse_magnet import _fill_orders

dev = torch.device('cuda', 0)
n, e, f, K = 10_000_000, 200_000_000, 256, 3
src, dst = powerlaw_edges(n, e, 43, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst
torch.cuda.empty_cache()
blk = torch.empty(n, 2, K + 1, f, device=dev)
blk[:, :, 0, :].normal_()
_fill_orders(lap, blk); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8):
    _fill_orders(lap, blk)
e1.record()
smi = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=30).stdout
torch.cuda.synchronize()
keep = [l.split(':', 1)[1].strip() for l in smi.splitlines() if ('clk' in l or 'ower' in l or 'junction' in l) and ':' in l and 'GPU[' in l]
print('%.1f ms per 3 orders | %s' % (e0.elapsed_time(e1) / 8, ' | '.join(keep)), flush=True)
