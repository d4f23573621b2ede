#!/usr/bin/env python3
"""Epoch time of the small reference configurations on one MI355X: this package (eager and as a
HIP graph) next to the reference formulation (uncoalesced torch.sparse COO, 4 spmm + 4 matmul per
order with precomputed sparse T_k) run on the same GPU through the oracle's restatement."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd import data as D                                               # noqa: E402
from magnet_amd.layer.sparse_magnet import ChebNet                              # noqa: E402
from magnet_amd.utils.hermitian import hermitian_decomp_sparse_device, cheb_poly_sparse   # noqa: E402
from magnet_amd.train import node_epoch, GraphedEpoch                           # noqa: E402
from oracle import magnet_oracle as O                                           # noqa: E402

dev = 'cuda'


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def run(name, g, q, K, Fh, layer, iters=200):
    n = g.x.size(0)
    ncls = int(g.y.max()) + 1
    L = hermitian_decomp_sparse_device(g.edge_index[0], g.edge_index[1], n, q)
    X = g.x.to(dev); y = g.y[None].to(dev); tr = g.train_mask[:, 0].nonzero().squeeze(1).to(dev); va = g.val_mask[:, 0].nonzero().squeeze(1).to(dev)   # index form: capturable
    torch.manual_seed(0)
    net = ChebNet(X.size(1), cheb_poly_sparse(L, K), None, num_filter=Fh, K=K, label_dim=ncls, activation=True, layer=layer).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True)
    eager = timeit(lambda: node_epoch(net, opt, X, X, y, tr, va), iters)
    graphed = GraphedEpoch(lambda: node_epoch(net, opt, X, X, y, tr, va))
    gms = timeit(graphed, iters)
    # reference formulation on the same GPU
    Lo = L.to_scipy()
    Lr, Li = O.laplacian_lists(O.cheb_poly_sparse(Lo, K))
    Lr = [t.to(dev) for t in Lr]; Li = [t.to(dev) for t in Li]
    state = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt2 = torch.optim.Adam(list(state.values()), lr=5e-3, weight_decay=5e-4)

    def ref_epoch():
        preds = O.chebnet_forward(state, Lr, Li, X, X, layer, True)
        loss = torch.nn.NLLLoss()(preds[:, :, tr], y[:, tr])
        opt2.zero_grad(); loss.backward(); opt2.step()
        with torch.no_grad():
            O.chebnet_forward(state, Lr, Li, X, X, layer, True)
    ref = timeit(ref_epoch, max(iters // 4, 10))
    print(json.dumps(dict(config=name, nodes=n, edges=int(g.edge_index.size(1)), nnz=L.nnz, K=K, filters=Fh,
                          epoch_ms_eager=round(eager, 3), epoch_ms_hipgraph=round(gms, 3),
                          epoch_ms_reference_formulation_same_gpu=round(ref, 3),
                          nnz_T=[int(t._nnz()) for t in Lr])), flush=True)


if __name__ == '__main__':
    run('C2 cyclic DSBM q=0.1 K=2', D.dsbm_cyclic(seed=10), 0.1, 2, 16, 2)
    g = D.dsbm_cyclic(seed=10)
    g.x = torch.randn(g.x.size(0), 64)
    run('cyclic DSBM, 64 input features, K=1', g, 0.25, 1, 16, 2)
