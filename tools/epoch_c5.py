#!/usr/bin/env python3
"""Does a ChebNet (LAYERS=2 by default) train on the config-5 graph (10M nodes / 200M edges, 256 features, K=3) on
ONE GPU?  MH_RECOMPUTE_ORDERS_BYTES=<bytes> trades K SpMMs per conv for the memory of the saved orders; SHARED=1 runs the
epochs through train.SharedForwardEpochs (one pass through the convolutions per epoch), NODE_TARGETS=1 uses
train.NodeTargets for the losses (no gather of the predictions)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
from magnet_amd.layer.sparse_magnet import ChebNet, ChebConv
from magnet_amd.utils.hermitian import ChebBasis
from magnet_amd.train import node_epoch, NodeTargets, SharedForwardEpochs
dev = 'cuda'
n, e, f, K = 10_000_000, 200_000_000, int(os.environ.get('F', 256)), int(os.environ.get('K', 3))
ChebConv.cache_input_propagation = os.environ.get('CACHE', '1') == '1'
src, dst = powerlaw_edges(n, e, 43, device=dev)
lap = build_magnetic_laplacian(src, dst, n, q=0.25)
del src, dst
torch.cuda.empty_cache()
net = ChebNet(f, ChebBasis(lap, K), None, num_filter=f, K=K, label_dim=5, activation=True, layer=int(os.environ.get('LAYERS', 2))).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
X = torch.randn(n, f, device=dev); y = torch.randint(0, 5, (1, n), device=dev)
tr = (torch.rand(n, device=dev) < 0.6).nonzero().squeeze(1); va = torch.arange(0, n, 7, device=dev)
if os.environ.get('NODE_TARGETS', '0') == '1':
    tr, va = NodeTargets(y, tr), NodeTargets(y, va)
shared = SharedForwardEpochs(net, opt) if os.environ.get('SHARED', '0') == '1' else None
for it in range(int(os.environ.get('EPOCHS', 3))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = shared.node_epoch(X, X, y, tr, va) if shared else node_epoch(net, opt, X, X, y, tr, va)
    torch.cuda.synchronize()
    print('epoch %d: %.1f ms, loss %.4f, peak %.1f GB' % (it, (time.perf_counter() - t0) * 1e3, out[0].item(), torch.cuda.max_memory_allocated() / 1e9), flush=True)
