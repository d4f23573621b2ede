#!/usr/bin/env python3
"""Epoch time of the small reference configurations on one MI355X: this package (eager and as a
HIP graph) next to the reference formulation (uncoalesced torch.sparse COO, 4 spmm + 4 matmul per
order with precomputed sparse T_k) run on the same GPU, restated below with torch's own sparse kernels."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnet_amd import data as D                                               # noqa: E402
from magnet_amd.layer.sparse_magnet import ChebNet                              # noqa: E402
from magnet_amd.utils.hermitian import hermitian_decomp_sparse_device, cheb_poly_sparse   # noqa: E402
from magnet_amd.train import node_epoch, GraphedEpoch, SharedForwardEpochs, GraphedSharedEpochs      # noqa: E402
from magnet_amd.sparse_Magnet import sparse_mx_to_torch_sparse_tensor          # noqa: E402

dev = 'cuda'


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def reference_formulation(state, Lr, Li, X, layer):
    """ChebNet forward as the reference computes it (src/layer/sparse_magnet.py:8-18,60-72,79-81,
    119-130): per order four torch.spmm on the precomputed T_k and two matmuls, bias on both parts,
    complex ReLU, 1x1 Conv1d head."""
    real = imag = X
    for j in range(layer):
        w, b = state['Chebs.%d.weight' % (2 * j)], state['Chebs.%d.bias' % (2 * j)]
        r = i = 0.0
        for k in range(len(Lr)):
            r = r + torch.matmul(torch.spmm(Lr[k], real) - torch.spmm(Li[k], imag), w[k])
            i = i + torch.matmul(torch.spmm(Li[k], real) + torch.spmm(Lr[k], imag), w[k])
        mask = 1.0 * (r + b >= 0)
        real, imag = mask * (r + b), mask * (i + b)
    x = torch.cat((real, imag), dim=-1).unsqueeze(0).permute(0, 2, 1)
    x = torch.nn.functional.conv1d(x, state['Conv.weight'], state['Conv.bias'])
    return torch.log_softmax(x, dim=1)


def run(name, g, q, K, Fh, layer, iters=200):
    n = g.x.size(0)
    ncls = int(g.y.max()) + 1
    L = hermitian_decomp_sparse_device(g.edge_index[0], g.edge_index[1], n, q)
    X = g.x.to(dev); y = g.y[None].to(dev); tr = g.train_mask[:, 0].nonzero().squeeze(1).to(dev); va = g.val_mask[:, 0].nonzero().squeeze(1).to(dev)   # index form: capturable
    torch.manual_seed(0)
    net = ChebNet(X.size(1), cheb_poly_sparse(L, K), None, num_filter=Fh, K=K, label_dim=ncls, activation=True, layer=layer).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True)
    eager = timeit(lambda: node_epoch(net, opt, X, X, y, tr, va), iters)
    sh = SharedForwardEpochs(net, opt)
    shared = timeit(lambda: sh.node_epoch(X, X, y, tr, va), iters)
    sh.reset()
    graphed = GraphedEpoch(lambda: node_epoch(net, opt, X, X, y, tr, va))
    gms = timeit(graphed, iters)
    gsh = timeit(GraphedSharedEpochs.for_nodes(net, opt, X, X, y, tr, va), iters)
    # reference formulation on the same GPU
    Ts = list(cheb_poly_sparse(L, K))                     # the sparse T_k the reference precomputes (host SpGEMM)
    Lr = [sparse_mx_to_torch_sparse_tensor(t.real).to(dev) for t in Ts]
    Li = [sparse_mx_to_torch_sparse_tensor(t.imag).to(dev) for t in Ts]
    state = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt2 = torch.optim.Adam(list(state.values()), lr=5e-3, weight_decay=5e-4)

    def ref_epoch():
        preds = reference_formulation(state, Lr, Li, X, layer)
        loss = torch.nn.NLLLoss()(preds[:, :, tr], y[:, tr])
        opt2.zero_grad(); loss.backward(); opt2.step()
        with torch.no_grad():
            reference_formulation(state, Lr, Li, X, layer)
    ref = timeit(ref_epoch, max(iters // 4, 10))
    print(json.dumps(dict(config=name, nodes=n, edges=int(g.edge_index.size(1)), nnz=L.nnz, K=K, filters=Fh,
                          epoch_ms_eager=round(eager, 3), epoch_ms_eager_shared_forward=round(shared, 3),
                          epoch_ms_hipgraph=round(gms, 3), epoch_ms_hipgraph_shared_forward=round(gsh, 3),
                          epoch_ms_reference_formulation_same_gpu=round(ref, 3),
                          nnz_T=[int(t._nnz()) for t in Lr])), flush=True)


def run_link(name, case, iters=200):
    """Configuration 3 on split 0 of a link-split cache the reference ships (arrays in tests/golden/link_split_cases.npz):
    ChebNet_Edge epoch = training step on the split's training pairs + evaluation forward on its validation pairs."""
    import numpy as np
    from magnet_amd.layer.sparse_magnet import ChebNet_Edge
    from magnet_amd.train import link_epoch
    from magnet_amd.utils.edge_data import in_out_degree
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'link_split_cases.npz'), allow_pickle=False)
    pre = case + '/'
    size, K, q = int(g[pre + 'size']), int(g[pre + 'params'][0]), float(g[pre + 'params'][1])
    edges = torch.from_numpy(g[pre + 'graph'].astype(np.int64)).to(dev)
    L = hermitian_decomp_sparse_device(edges[0], edges[1], size, q)
    X = in_out_degree(edges, size).to(dev)
    t = lambda k: torch.from_numpy(g[pre + k].astype(np.int64)).to(dev)
    tp, tl, vp, vl = t('train/pairs'), t('train/label'), t('validate/pairs'), t('validate/label')
    torch.manual_seed(0)
    net = ChebNet_Edge(2, cheb_poly_sparse(L, K), None, num_filter=16, K=K, label_dim=2, activation=True, layer=2, dropout=0.0).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True)
    eager = timeit(lambda: link_epoch(net, opt, X, X, tp, tl, vp, vl), iters)
    sh = SharedForwardEpochs(net, opt)
    shared = timeit(lambda: sh.link_epoch(X, X, tp, tl, vp, vl), iters)
    sh.reset()
    gms = timeit(GraphedEpoch(lambda: link_epoch(net, opt, X, X, tp, tl, vp, vl)), iters)
    gsh = timeit(GraphedSharedEpochs.for_links(net, opt, X, X, tp, tl, vp, vl), iters)
    Ts = list(cheb_poly_sparse(L, K))
    Lr = [sparse_mx_to_torch_sparse_tensor(m.real).to(dev) for m in Ts]
    Li = [sparse_mx_to_torch_sparse_tensor(m.imag).to(dev) for m in Ts]
    state = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt2 = torch.optim.Adam(list(state.values()), lr=5e-3, weight_decay=5e-4)

    def ref_forward(index):
        # src/layer/sparse_magnet.py:8-18,60-72,79-81,157-165 with torch's own sparse kernels
        real = imag = X
        for j in range(2):
            w, b = state['Chebs.%d.weight' % (2 * j)], state['Chebs.%d.bias' % (2 * j)]
            r = i = 0.0
            for k in range(len(Lr)):
                r = r + torch.matmul(torch.spmm(Lr[k], real) - torch.spmm(Li[k], imag), w[k])
                i = i + torch.matmul(torch.spmm(Li[k], real) + torch.spmm(Lr[k], imag), w[k])
            mask = 1.0 * (r + b >= 0)
            real, imag = mask * (r + b), mask * (i + b)
        x = torch.cat((real[index[:, 0]], real[index[:, 1]], imag[index[:, 0]], imag[index[:, 1]]), dim=-1)
        return torch.log_softmax(torch.nn.functional.linear(x, state['linear.weight'], state['linear.bias']), dim=1)

    def ref_epoch():
        loss = torch.nn.functional.nll_loss(ref_forward(tp), tl)
        opt2.zero_grad(); loss.backward(); opt2.step()
        with torch.no_grad():
            ref_forward(vp)
    ref = timeit(ref_epoch, max(iters // 4, 10))
    print(json.dumps(dict(config=name, nodes=size, edges=int(edges.size(1)), nnz=L.nnz, K=K, filters=16, train_pairs=int(tp.size(0)),
                          epoch_ms_eager=round(eager, 3), epoch_ms_eager_shared_forward=round(shared, 3),
                          epoch_ms_hipgraph=round(gms, 3), epoch_ms_hipgraph_shared_forward=round(gsh, 3),
                          epoch_ms_reference_formulation_same_gpu=round(ref, 3), nnz_T=[int(m._nnz()) for m in Lr])), flush=True)


if __name__ == '__main__':
    run('C2 cyclic DSBM q=0.1 K=2', D.dsbm_cyclic(seed=10), 0.1, 2, 16, 2)
    g = D.dsbm_cyclic(seed=10)
    g.x = torch.randn(g.x.size(0), 64)
    run('cyclic DSBM, 64 input features, K=1', g, 0.25, 1, 16, 2)
    run_link('C3 CiteSeer existence, split 0 of the shipped cache, q=0.1 K=2', 'citeseer_task1')
    run_link('C3 Cora-ML direction, split 0 of the shipped cache, q=0.1 K=2', 'cora_ml_task2')
