"""Config-4 training epochs (2-layer ChebNet, 128 filters, K=2, on a 2M-node / 40M-edge power-law digraph) and nothing else,
for `rocprofv3 --kernel-trace --stats -- python3 tools/epoch_profile.py`: the per-kernel table then shows where an epoch's
time goes.  --cached keeps the first layer's propagated input across forwards; --shared runs the epochs through
train.SharedForwardEpochs (the evaluation's pass through the convolutions kept for the next training step)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--nodes', type=int, default=2_000_000)
    p.add_argument('--edges', type=int, default=40_000_000)
    p.add_argument('--epochs', type=int, default=5)
    p.add_argument('--cached', action='store_true')
    p.add_argument('--shared', action='store_true')
    p.add_argument('--gather-loss', action='store_true', help='the reference form of the loss: criterion(preds[:, :, idx], label[:, idx])')
    p.add_argument('--layout', choices=['single', 'cols', 'rows'], default='single',
                   help='cols / rows: the multi-GPU networks (ShardedChebNet / RowShardedChebNet) with one rank')
    a = p.parse_args()
    from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges
    from magnet_amd.layer.sparse_magnet import ChebNet, ChebConv
    from magnet_amd.utils.hermitian import ChebBasis
    from magnet_amd.train import node_epoch, NodeTargets, SharedForwardEpochs
    dev = torch.device('cuda', 0)
    f, K, ncls, n = 128, 2, 5, a.nodes
    src, dst = powerlaw_edges(n, a.edges, 42, device=dev)
    lap = build_magnetic_laplacian(src, dst, n, q=0.25)
    del src, dst
    torch.manual_seed(2)
    net = ChebNet(f, ChebBasis(lap, K), None, num_filter=f, K=K, label_dim=ncls, activation=True, layer=2).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
    gen = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn(n, f, device=dev, generator=gen)
    y = torch.randint(0, ncls, (1, n), device=dev, generator=gen)
    train = torch.rand(n, device=dev, generator=gen) < 0.6
    tr = train.nonzero().squeeze(1); va = (~train).nonzero().squeeze(1)
    n_tr, n_va = int(tr.numel()), int(va.numel())
    if not a.gather_loss:
        tr, va = NodeTargets(y, tr), NodeTargets(y, va)
    ChebConv.cache_input_propagation = a.cached
    run = lambda: node_epoch(net, opt, X, X, y, tr, va)
    if a.shared:
        shared = SharedForwardEpochs(net, opt)
        run = lambda: shared.node_epoch(X, X, y, tr, va)
    if a.layout != 'single':
        import torch.distributed as dist
        from magnet_amd import dist as D
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29541', rank=0, world_size=1)
        if a.layout == 'rows':
            part = D.RowPartition.from_global(lap.row_ptr, lap.col_idx, lap.val, n, 0, 1, lambda rp, ci, v, nc: type(lap)(rp, ci, v, nc))
            sh = D.RowShardedChebNet(net, part)
        else:
            sh = D.ShardedChebNet(net, lap, n, 0, 1)
        sh.cache_input_propagation = a.cached
        xp = torch.stack([X, X], dim=1).contiguous()
        run = lambda: sh.epoch(opt, xp, y, tr, va, n_tr, n_va)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.epochs):
        run()
    torch.cuda.synchronize()
    print('epoch %.2f ms (nnz %d, cached=%s, shared=%s, layout=%s)' % ((time.perf_counter() - t0) / a.epochs * 1e3, lap.nnz, a.cached, a.shared, a.layout))


if __name__ == '__main__':
    main()
