"""The row-partition / block-exchange logic with real processes: world_size 2 and 3 over gloo on
the CPU.  The sparse operator is injected (scipy here; the HIP kernel in production), so what is
tested is exactly the code bench.py runs with --gpus N: partition bounds, row slicing, the
local/remote column split, the overlapped propagate order and the point-to-point exchange."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import magnet_oracle as O
from magnet_amd import dist as D


class ScipyOperator:
    """CSR tensors -> object with the .spmm signature of MagneticLaplacian (CPU, test only)."""

    def __init__(self, rp, ci, val, n_cols):
        v = val.numpy().astype(np.float64)
        self.m = sp.csr_matrix((v[:, 0] + 1j * v[:, 1], ci.numpy(), rp.numpy()), shape=(rp.numel() - 1, n_cols))
        self.nnz = int(ci.numel())

    def rows(self, r0, r1):
        o = ScipyOperator.__new__(ScipyOperator)
        o.m = self.m[r0:r1]
        o.nnz = int(o.m.nnz)
        return o

    def spmm(self, X, out=None, alpha=1.0, z1=None, beta1=0.0, z2=None, beta2=0.0, bias=None, relu=False, mask=None):
        c = lambda t: t[:, 0, :].detach().double().numpy() + 1j * t[:, 1, :].detach().double().numpy()
        y = alpha * (self.m @ c(X))
        if z1 is not None:
            y = y + beta1 * c(z1)
        if z2 is not None:
            y = y + beta2 * c(z2)
        if bias is not None:
            b = bias.detach().double().numpy().reshape(-1)
            y = y + b + 1j * b
        if relu:
            m = y.real >= 0
            y = y * m
            if mask is not None:
                mask.copy_(torch.from_numpy(m.astype(np.uint8)))
        if out is None:
            out = torch.empty((self.m.shape[0], 2, X.size(2)), dtype=torch.float32)
        out[:, 0, :] = torch.from_numpy(y.real).float()
        out[:, 1, :] = torch.from_numpy(y.imag).float()
        return out


def _graph(n=400, e=6000, seed=4):
    src, dst = O.powerlaw_edges(n, e, seed)
    L = O.hermitian_decomp_sparse(src, dst, n, 0.2)
    val = torch.from_numpy(np.stack([L.data.real, L.data.imag], 1).astype(np.float32))
    return L, torch.from_numpy(L.indptr), torch.from_numpy(L.indices), val


def _banded_graph(n=600, reach=6, seed=2):
    """A graph with locality: node i points to a few nodes within `reach` of it (ring), so a row block's
    halo is a thin band of its neighbours' blocks."""
    rng = np.random.default_rng(seed)
    src = np.repeat(np.arange(n), 3)
    dst = (src + rng.integers(1, reach + 1, len(src))) % n
    L = O.hermitian_decomp_sparse(src, dst, n, 0.2)
    val = torch.from_numpy(np.stack([L.data.real, L.data.imag], 1).astype(np.float32))
    return L, torch.from_numpy(L.indptr), torch.from_numpy(L.indices), val


def _worker(rank, world, port, n_terms, out_dir, banded=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        L, rp, ci, val = _banded_graph() if banded else _graph()
        n = L.shape[0]
        part = D.RowPartition.from_global(rp, ci, val, n, rank, world, ScipyOperator)
        x = torch.randn(n, 2, 5, generator=torch.Generator().manual_seed(0))
        xb = part.new_buffer(5, 'cpu'); xb.fill_(float('nan'))
        part.block(xb).copy_(x[part.r0:part.r1])               # the rank starts from its own rows only
        zs = part.cheb_terms(xb, n_terms)
        torch.save({'bounds': part.bounds, 'blocks': [part.block(z).clone() for z in zs],
                    'halo': [z[part.n_local:].clone() for z in zs[:-1]], 'halo_ids': part.halo_ids.clone(),
                    'peer_ptr': part.peer_ptr, 'sent_rows': part.sent_rows(), 'n_local': part.n_local,
                    'nnz': (part.L_int.nnz, part.L_ext.nnz, part.local_nnz)}, os.path.join(out_dir, 'r%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_modes(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        L, rp, ci, val = _graph()
        n, f, n_terms = L.shape[0], 4, 4
        part = D.RowPartition.from_global(rp, ci, val, n, rank, world, ScipyOperator)
        x = torch.randn(n, 2, f, generator=torch.Generator().manual_seed(0))
        ys = torch.randn(n, 2, n_terms, f, generator=torch.Generator().manual_seed(1))
        out = {'bounds': part.bounds}
        for mode in D.RowPartition.MODES:
            part.set_mode(mode, int_grid_cap=64)
            block = torch.full((part.n_local, 2, n_terms, f), float('nan'))
            block[:, :, 0, :] = x[part.r0:part.r1]
            part.cheb_block(block)
            cl = part.clenshaw([ys[part.r0:part.r1, :, k, :] for k in range(n_terms)])
            out[mode] = (block.clone(), cl.clone())
        torch.save(out, os.path.join(out_dir, 'm%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_row_partition_exchange_modes_agree(world, tmp_path):
    """Every arrangement of the halo exchange (overlapped point-to-point, the same with a capped launch width,
    exchange-then-compute, one all-gather of the padded own blocks) gives the recursion and the Clenshaw sum of
    the single-process run."""
    mp.spawn(_worker_modes, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    L, rp, ci, val = _graph()
    n, f, n_terms = L.shape[0], 4, 4
    x = torch.randn(n, 2, f, generator=torch.Generator().manual_seed(0))
    ys = torch.randn(n, 2, n_terms, f, generator=torch.Generator().manual_seed(1))
    c = lambda t: t[:, 0, :].double().numpy() + 1j * t[:, 1, :].double().numpy()
    Ld = L.astype(np.complex128)
    ref = [c(x), Ld @ c(x)]
    for k in range(2, n_terms):
        ref.append(2 * (Ld @ ref[k - 1]) - ref[k - 2])
    # sum_k T_k y_k with T_k applied by the same recursion
    want_cl = np.zeros((n, f), dtype=np.complex128)
    for k in range(n_terms):
        t = [c(ys[:, :, k, :]), Ld @ c(ys[:, :, k, :])]
        for j in range(2, k + 1):
            t.append(2 * (Ld @ t[j - 1]) - t[j - 2])
        want_cl += t[k]
    outs = [torch.load(os.path.join(str(tmp_path), 'm%d.pt' % r)) for r in range(world)]
    for mode in D.RowPartition.MODES:
        for k in range(n_terms):
            got = np.concatenate([c(o[mode][0][:, :, k, :]) for o in outs])
            assert np.abs(got - ref[k]).max() <= 1e-5 * np.abs(ref[k]).max(), (mode, k)
        got = np.concatenate([c(o[mode][1]) for o in outs])
        assert np.abs(got - want_cl).max() <= 1e-5 * np.abs(want_cl).max(), mode


def _worker_cols(rank, world, port, n_terms, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        L, rp, ci, val = _graph()
        n, f, cout = L.shape[0], 10, 6
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n, 2, f, generator=g)
        W = torch.randn(n_terms, f, cout, generator=g); b = torch.randn(1, cout, generator=g)
        sh = D.ColumnSharded(ScipyOperator(rp, ci, val, n), n, f, rank, world)
        out = sh.conv(x[:, :, sh.c0:sh.c1].contiguous(), W, b)
        torch.save({'rb': sh.rb, 'cb': sh.cb, 'out': out}, os.path.join(out_dir, 'c%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_terms', [(2, 4), (3, 4), (4, 3), (2, 2), (3, 1)])
def test_column_sharded_conv_matches_single_process(world, n_terms, tmp_path):
    mp.spawn(_worker_cols, args=(world, _free_port(), n_terms, str(tmp_path)), nprocs=world, join=True)
    L, rp, ci, val = _graph()
    n, f, cout = L.shape[0], 10, 6
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, 2, f, generator=g)
    W = torch.randn(n_terms, f, cout, generator=g); b = torch.randn(1, cout, generator=g)
    rr, ri = O.cheb_conv_recursion(L, W, b, x[:, 0, :], x[:, 1, :])
    outs = [torch.load(os.path.join(str(tmp_path), 'c%d.pt' % r)) for r in range(world)]
    assert outs[0]['cb'][0] == 0 and outs[0]['cb'][-1] == f and outs[0]['rb'][-1] == n
    got = torch.cat([o['out'] for o in outs], dim=0)
    scale = float(rr.abs().max())
    assert float((got[:, 0, :] - rr).abs().max()) <= 1e-5 * scale
    assert float((got[:, 1, :] - ri).abs().max()) <= 1e-5 * scale


def test_column_bounds():
    assert D.column_bounds(256, 8) == [32 * p for p in range(9)]
    b = D.column_bounds(100, 8)
    assert b[0] == 0 and b[-1] == 100 and all(x <= y for x, y in zip(b, b[1:])) and all(x % 4 == 0 for x in b[:-1])
    b = D.column_bounds(5, 4)
    assert b[0] == 0 and b[-1] == 5 and all(x <= y for x, y in zip(b, b[1:]))


def _worker_train(rank, world, port, out_dir, layer, cached=False, rows=False, shared=False, epochs=2):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import magnet_amd.layer.sparse_magnet as M
        from magnet_amd.utils.hermitian import ChebBasis
        # CPU rehearsal: the ReLU helpers normally call the HIP library; route them to torch here
        def bias_relu_(p, bias, relu):
            if bias is not None:
                p += bias.detach().reshape(1, 1, -1)
            if not relu:
                return None
            m = (p[:, 0, :] >= 0)
            p *= m.unsqueeze(1).float()
            return m.to(torch.uint8)
        M._bias_relu_ = bias_relu_
        M._relu_bwd_ = lambda g, mask: g.mul_(mask.unsqueeze(1).float())
        def relu_bwd(g, mask, out=None):
            r = g * mask.unsqueeze(1).float()
            return r if out is None else out.copy_(r)
        M._relu_bwd = relu_bwd
        L, rp, ci, val = _graph()
        n, cin, Fh, K, ncls = L.shape[0], 6, 8, 2, 4
        op = ScipyOperator(rp, ci, val, n)
        torch.manual_seed(0)
        net = M.ChebNet(cin, ChebBasis(op, K), None, num_filter=Fh, K=K, label_dim=ncls, activation=True, layer=layer)
        g = torch.Generator().manual_seed(1)
        X = torch.randn(n, cin, generator=g); y = torch.randint(0, ncls, (1, n), generator=g)
        train = torch.rand(n, generator=g) < 0.5
        if rows:            # the row partition: every rank holds its row block only
            sh = D.RowShardedChebNet(net, D.RowPartition.from_global(rp, ci, val, n, rank, world, ScipyOperator))
        else:
            sh = D.ShardedChebNet(net, op, n, rank, world)
        sh.cache_input_propagation = cached
        sh.shared_forward = shared
        r0, r1 = sh.r0, sh.r1
        xp = torch.stack([X[r0:r1], X[r0:r1]], dim=1)
        opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
        tr_rows = train[r0:r1].nonzero().squeeze(1); va_rows = (~train[r0:r1]).nonzero().squeeze(1)
        losses = []
        for _ in range(epochs):
            tl, vl = sh.epoch(opt, xp, y[:, r0:r1], tr_rows, va_rows, int(train.sum()), int((~train).sum()))
            losses.append((tl.item(), vl.item()))
        torch.save({'losses': losses, 'state': {k: v.clone() for k, v in net.state_dict().items()}},
                   os.path.join(out_dir, 't%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,layer,cached,rows', [(2, 2, False, False), (3, 3, False, False), (2, 2, True, False), (3, 1, True, False),
                                                     (2, 2, False, True), (3, 3, False, True), (2, 2, True, True)])
def test_sharded_chebnet_training_matches_single_process(world, layer, cached, rows, tmp_path):
    mp.spawn(_worker_train, args=(world, _free_port(), str(tmp_path), layer, cached, rows), nprocs=world, join=True)
    # single-process reference: the oracle's forward under autograd with the same initial weights
    import magnet_amd.layer.sparse_magnet as M
    from magnet_amd.utils.hermitian import ChebBasis
    L, rp, ci, val = _graph()
    n, cin, Fh, K, ncls = L.shape[0], 6, 8, 2, 4
    torch.manual_seed(0)
    net = M.ChebNet(cin, ChebBasis(ScipyOperator(rp, ci, val, n), K), None, num_filter=Fh, K=K, label_dim=ncls,
                    activation=True, layer=layer)
    state = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    X = torch.randn(n, cin, generator=g); y = torch.randint(0, ncls, (1, n), generator=g)
    train = torch.rand(n, generator=g) < 0.5
    Lr, Li = O.laplacian_lists(O.cheb_poly_sparse(L, K))
    opt = torch.optim.Adam([state[k] for k, _ in net.named_parameters()], lr=5e-3, weight_decay=5e-4)
    ref_losses = []
    for _ in range(2):
        preds = O.chebnet_forward(state, Lr, Li, X, X, layer, True)
        loss = torch.nn.NLLLoss()(preds[:, :, train], y[:, train])
        opt.zero_grad(); loss.backward(); opt.step()
        with torch.no_grad():
            v = torch.nn.NLLLoss()(O.chebnet_forward(state, Lr, Li, X, X, layer, True)[:, :, ~train], y[:, ~train])
        ref_losses.append((loss.item(), v.item()))
    outs = [torch.load(os.path.join(str(tmp_path), 't%d.pt' % r)) for r in range(world)]
    for o in outs:
        for (a, b), (c, d) in zip(o['losses'], ref_losses):
            assert abs(a - c) <= 5e-5 * abs(c) and abs(b - d) <= 5e-5 * abs(d)
        for k in state:
            assert float((o['state'][k] - state[k].detach()).abs().max()) <= 5e-5 * float(state[k].detach().abs().max()) + 1e-6, k


@pytest.mark.parametrize('world,layer,cached,rows', [(2, 2, False, False), (2, 2, True, True), (3, 2, False, True)])
def test_sharded_epochs_with_shared_forward_match_plain_epochs(world, layer, cached, rows, tmp_path):
    """shared_forward (the evaluation's pass through the convolutions kept for the next training step) gives the
    losses and parameters of the plain sharded epochs, over three epochs, in both layouts."""
    plain, shared = os.path.join(str(tmp_path), 'plain'), os.path.join(str(tmp_path), 'shared')
    os.makedirs(plain); os.makedirs(shared)
    mp.spawn(_worker_train, args=(world, _free_port(), plain, layer, cached, rows, False, 3), nprocs=world, join=True)
    mp.spawn(_worker_train, args=(world, _free_port(), shared, layer, cached, rows, True, 3), nprocs=world, join=True)
    for r in range(world):
        a = torch.load(os.path.join(plain, 't%d.pt' % r)); b = torch.load(os.path.join(shared, 't%d.pt' % r))
        for (x0, y0), (x1, y1) in zip(a['losses'], b['losses']):
            assert abs(x0 - x1) <= 1e-6 * abs(x0) and abs(y0 - y1) <= 1e-6 * abs(y0)
        for k in a['state']:
            assert float((a['state'][k] - b['state'][k]).abs().max()) <= 1e-6 * float(a['state'][k].abs().max()) + 1e-7, k


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('world,banded', [(2, False), (3, False), (3, True), (4, True)])
def test_row_partitioned_recursion_matches_single_process(world, banded, tmp_path):
    n_terms = 4
    mp.spawn(_worker, args=(world, _free_port(), n_terms, str(tmp_path), banded), nprocs=world, join=True)
    L, rp, ci, val = _banded_graph() if banded else _graph()
    n = L.shape[0]
    x = torch.randn(n, 2, 5, generator=torch.Generator().manual_seed(0))
    xc = x[:, 0, :].double().numpy() + 1j * x[:, 1, :].double().numpy()
    Ld = L.astype(np.complex128)
    ref = [xc, Ld @ xc]
    for k in range(2, n_terms):
        ref.append(2 * (Ld @ ref[k - 1]) - ref[k - 2])
    outs = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    bounds = outs[0]['bounds']
    assert bounds[0] == 0 and bounds[-1] == n and all(o['bounds'] == bounds for o in outs)
    assert sum(o['nnz'][2] for o in outs) == L.nnz
    for r, o in enumerate(outs):
        assert o['nnz'][0] + o['nnz'][1] == o['nnz'][2]
        for k in range(n_terms):
            got = o['blocks'][k][:, 0, :].double().numpy() + 1j * o['blocks'][k][:, 1, :].double().numpy()
            want = ref[k][bounds[r]:bounds[r + 1]]
            assert np.abs(got - want).max() <= 1e-5 * np.abs(ref[k]).max()
        # after an order's exchange the halo holds the peers' rows of that order, in halo_ids order
        ids = o['halo_ids'].long().numpy()
        assert np.all(np.diff(ids) > 0) and not np.any((ids >= bounds[r]) & (ids < bounds[r + 1]))
        assert o['peer_ptr'][0] == 0 and o['peer_ptr'][-1] == len(ids)
        for k in range(n_terms - 1):
            got = o['halo'][k][:, 0, :].double().numpy() + 1j * o['halo'][k][:, 1, :].double().numpy()
            assert np.abs(got - ref[k][ids]).max() <= 1e-5 * np.abs(ref[k]).max()
        if banded:
            assert len(ids) <= 4 * 6 and o['sent_rows'] <= 4 * 6                  # only the band around the block travels
        else:
            assert len(ids) > 0.5 * (n - o['n_local'])                           # power-law graph: most remote rows
    assert sum(o['sent_rows'] for o in outs) == sum(len(o['halo_ids']) for o in outs)


def test_halo_plan_restatement_and_index_exchange_single_rank():
    ci = torch.tensor([0, 5, 9, 2, 3, 9, 7, 1], dtype=torch.int32)
    bounds = torch.tensor([0, 2, 6, 10], dtype=torch.int32)
    ids, pp, cl = D.halo_plan_torch(ci, 10, 2, 6, bounds)                        # own columns 2..5
    assert ids.tolist() == [0, 1, 7, 9] and pp.tolist() == [0, 2, 2, 4]
    assert cl.tolist() == [4, 3, 7, 0, 1, 7, 6, 5]
    assert D.exchange_index_lists([None], 0, 1) == [None]
    assert D.row_partition_floats(10, 4, 3) == 3 * 10 * 8 and D.column_sharding_floats(100, 8, 3, 4) == 4 * 100 * 16 * 3 // 16


def test_partition_bounds_balance_and_edge_cases():
    _, rp, ci, val = _graph()
    n = rp.numel() - 1
    for world in (1, 2, 4, 8):
        b = D.partition_bounds(rp, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(world))
        cost = [int(rp[b[i + 1]] - rp[b[i]]) + 2 * (b[i + 1] - b[i]) for i in range(world)]
        assert max(cost) <= sum(cost) / world + int((rp[1:] - rp[:-1]).max()) + 2
    empty = torch.zeros(6, dtype=torch.int32)
    assert D.partition_bounds(empty, 4)[-1] == 5
    assert D.partition_bounds(torch.zeros(1, dtype=torch.int32), 2) == [0, 0, 0]


def test_split_columns_and_slice_rows():
    L, rp, ci, val = _graph()
    n = L.shape[0]
    r0, r1 = 100, 250
    rps, cis, vs = D.slice_rows(rp, ci, val, r0, r1)
    sub = L[r0:r1]
    assert np.array_equal(rps.numpy(), sub.indptr) and np.array_equal(cis.numpy(), sub.indices)
    (rpi, cii, vi), (rpe, cie, ve) = D.split_columns(rps, cis, vs, r0, r1)
    assert bool(((cii >= r0) & (cii < r1)).all()) and bool(((cie < r0) | (cie >= r1)).all())
    a = ScipyOperator(rpi, cii, vi, n).m + ScipyOperator(rpe, cie, ve, n).m
    assert abs(a - sub.astype(np.complex128)).max() < 1e-7


def test_choose_strategy_follows_the_exchange_volume():
    """Row partition when it needs no or little exchange (K <= 1, graphs whose neighbours have nearby ids),
    column sharding on a graph without locality."""
    L, rp, ci, val = _graph(n=4000, e=60000)
    assert D.choose_strategy(rp, ci, 4000, 64, 3, 4) == 'cols'
    assert D.choose_strategy(rp, ci, 4000, 64, 0, 4) == 'rows'          # nothing to propagate, nothing to exchange
    assert D.choose_strategy(rp, ci, 4000, 6, 3, 4) == 'rows'           # columns do not split into float4 slices
    assert D.choose_strategy(rp, ci, 4000, 64, 3, 1) == 'rows'
    Lb, rpb, cib, valb = _banded_graph(n=4000, reach=6)
    r, c = D.exchange_volumes(rpb, cib, 4000, 64, 3, 4)
    assert r < c / 10 and D.choose_strategy(rpb, cib, 4000, 64, 3, 4) == 'rows'
    r, c = D.exchange_volumes(rp, ci, 4000, 64, 3, 4)
    assert r > c
