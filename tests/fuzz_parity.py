#!/usr/bin/env python3
"""Randomised parity sweep (not collected by pytest; run by hand on a GPU box):

    python tests/fuzz_parity.py --cases 300 --seed 0

Every case draws a random digraph (sizes, duplicates, self-loops, reciprocal pairs, integer or float
weights, q, max_eigen, gcn_appr), builds the magnetic Laplacian with the HIP builder and with the oracle
(structure must be identical, values within 1e-5), then runs one ChebConv forward + backward with random
widths / order / ReLU against the oracle's fp64 recursion under autograd."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import magnet_oracle as O                      # noqa: E402
from magnet_amd.utils import hermitian as H                # noqa: E402
from magnet_amd.layer import sparse_magnet as M            # noqa: E402


def close(a, b, rel, what, floor=1e-7):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
    err = float(np.abs(a - b).max()) if b.size else 0.0
    if err > rel * scale + floor:
        raise AssertionError('%s: max err %.3e vs scale %.3e' % (what, err, scale))


def one_case(rng, idx):
    n = int(rng.choice([1, 2, 3, 7, 50, 300, 2000, 20000]))
    e = int(rng.integers(0, 30 * n + 1)) if n > 1 else int(rng.integers(0, 3))
    row = rng.integers(0, n, e); col = rng.integers(0, n, e)
    if e > 4:                                              # reciprocal pairs and duplicates on purpose
        k = e // 4
        row[:k], col[:k] = col[k:2 * k].copy(), row[k:2 * k].copy()
        row[2 * k:2 * k + k // 2] = row[:k // 2]; col[2 * k:2 * k + k // 2] = col[:k // 2]
    if n >= 2000 and rng.integers(0, 3) == 0:              # hub rows: long enough for the chunked hub-row path
        hubs = rng.integers(0, n, 2)
        extra = int(rng.integers(1500, 6000))
        hr = np.concatenate([np.full(extra, hubs[0]), rng.integers(0, n, extra)])
        hc = np.concatenate([rng.integers(0, n, extra), np.full(extra, hubs[1])])
        row = np.concatenate([row, hr]); col = np.concatenate([col, hc]); e = len(row)
    kind = rng.integers(0, 3)
    w = None if kind == 0 else (rng.integers(1, 9, e).astype(np.float32) if kind == 1 else rng.random(e).astype(np.float32))
    q = float(rng.choice([0.0, 0.05, 0.1, 0.25, 0.5, rng.random() * 0.5]))
    max_eigen = float(rng.choice([2.0, 2.0, 3.0, 1.5]))
    gcn = bool(rng.integers(0, 4) == 0)
    desc = 'n=%d e=%d w=%d q=%r lam=%.1f gcn=%d' % (n, e, kind, q, max_eigen, gcn)
    Lo = O.hermitian_decomp_sparse(row, col, n, q, max_eigen=max_eigen, gcn_appr=gcn, edge_weight=w)
    L = H.hermitian_decomp_sparse_device(row, col, n, q, max_eigen=max_eigen, gcn_appr=gcn, edge_weight=w)
    S = L.to_scipy()
    Lo.sort_indices()
    if kind != 2:                                          # float weights: duplicate sums are order dependent in scipy
        if not (np.array_equal(S.indptr, Lo.indptr) and np.array_equal(S.indices, Lo.indices)):
            D = (abs(S) > 0).astype(np.int8) - (abs(Lo) > 0).astype(np.int8)
            D.eliminate_zeros(); D = D.tocoo()
            extra = [(int(i), int(j), int(v), complex(S[i, j]), complex(Lo[i, j])) for i, j, v in zip(D.row[:5], D.col[:5], D.data[:5])]
            raise AssertionError('structure %s: nnz %d vs %d, stored-zero counts %d / %d, first differences %s' % (
                desc, S.nnz, Lo.nnz, int((S.data == 0).sum()), int((Lo.data == 0).sum()), extra))
        for part, a_, b_ in (('real', S.data.real, Lo.data.real), ('imag', S.data.imag, Lo.data.imag)):
            try:
                close(a_, b_, 1e-5, 'L ' + part)
            except AssertionError as ex:
                k = int(np.abs(a_ - b_).argmax())
                r_ = int(np.searchsorted(S.indptr, k, side='right') - 1)
                raise AssertionError('%s %s: entry (%d,%d) hip %r oracle %r' % (ex, desc, r_, S.indices[k], S.data[k], Lo.data[k]))
    else:
        D = (S - Lo).tocoo()
        close(np.abs(D.data) if D.nnz else np.zeros(1), np.zeros(max(D.nnz, 1)), 0, 'L (float weights)') if D.nnz and np.abs(D.data).max() > 1e-5 * max(np.abs(Lo.data).max(), 1e-30) + 1e-7 else None
    # one conv, forward + backward
    K = int(rng.integers(0, 4)); cin = int(rng.choice([1, 2, 3, 4, 8, 17, 32])); cout = int(rng.choice([1, 2, 4, 5, 16, 40]))
    relu = bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    conv = M.ChebConv(cin, cout, K, H.cheb_poly_sparse(L, K), None).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-0.3, 0.3)
    Xr = torch.randn(n, cin, generator=g); Xi = torch.randn(n, cin, generator=g)
    xr = Xr.cuda().requires_grad_(True); xi = Xi.cuda().requires_grad_(True)
    p = conv.forward_packed(M._as_packed(xr, xi), relu=relu)
    Wt = torch.randn(n, 2, cout, generator=g)
    (p * Wt.cuda()).sum().backward()
    # oracle in fp64 under autograd (S is the HIP-built matrix: the builder was checked above)
    Ld = torch.sparse_coo_tensor(np.vstack(S.tocoo().coords), torch.from_numpy(S.tocoo().data.astype(np.complex128)), (n, n)).coalesce()
    W = conv.weight.detach().cpu().double().requires_grad_(True); b = conv.bias.detach().cpu().double().requires_grad_(True)
    X = torch.complex(Xr.double(), Xi.double()).requires_grad_(True)
    zs = [X]
    if K >= 1:
        zs.append(torch.sparse.mm(Ld, X))
    for k in range(2, K + 1):
        zs.append(2.0 * torch.sparse.mm(Ld, zs[k - 1]) - zs[k - 2])
    out = sum(z @ W[k].to(torch.complex128) for k, z in enumerate(zs)) + torch.complex(b, b)
    o_re, o_im = out.real, out.imag
    if relu:
        m = (o_re >= 0).double()
        near = (o_re.abs() < 1e-5 * max(float(o_re.abs().max()), 1e-30))          # sign undecided within rounding
        if bool(near.any()):
            return 'skipped (a real part within rounding of zero decides the gate)'
        o_re, o_im = o_re * m, o_im * m
    (o_re * Wt[:, 0].double() + o_im * Wt[:, 1].double()).sum().backward()
    pc = p.detach().cpu()
    # an output that cancellation has made small is accurate relative to the magnitude of the terms it is summed from
    # (float32 backward error), not relative to itself
    terms = sum(z.detach().abs() @ W[k].detach().abs() for k, z in enumerate(zs)) + b.detach().abs()
    out_floor = max(1e-7, 8 * 6e-8 * float(terms.max()))
    close(pc[:, 0], o_re.detach(), 1e-5, 'out real', out_floor); close(pc[:, 1], o_im.detach(), 1e-5, 'out imag', out_floor)
    # dW[k] is a sum over all nodes of products of both signs, like db below: float32 accumulates it to within its
    # backward error — a few ulps of the sum of the products' magnitudes — not of a result that cancellation made small
    gmag = Wt[:, 0].double().abs() + Wt[:, 1].double().abs()
    dw_floor = 8 * 6e-8 * max(float(((z.detach().real.abs() + z.detach().imag.abs()).t() @ gmag).max()) for z in zs)
    close(conv.weight.grad.cpu(), W.grad, 1e-5, 'dW', max(1e-7, dw_floor))
    # db is a plain sum over all nodes of terms of both signs: a float32 sum is accurate relative to the sum of the
    # terms' magnitudes (backward error), not to a result that cancellation has made small
    db_err = float((conv.bias.grad.cpu().double() - b.grad).abs().max())
    db_floor = 8 * 6e-8 * float(Wt.double().abs().sum(dim=(0, 1)).max())
    if db_err > 1e-5 * float(b.grad.abs().max()) + db_floor:
        raise AssertionError('db: max err %.3e vs scale %.3e (summation floor %.1e)' % (db_err, float(b.grad.abs().max()), db_floor))
    # d/dX of a real loss wrt complex X: torch returns conj-gradient = dRe + i dIm
    close(xr.grad.cpu(), X.grad.real, 1e-5, 'dX real'); close(xi.grad.cpu(), X.grad.imag, 1e-5, 'dX imag')
    msg = 'ok n=%d e=%d nnz=%d K=%d %d->%d relu=%d w=%d q=%.3f lam=%.1f gcn=%d' % (n, e, L.nnz, K, cin, cout, relu, kind, q, max_eigen, gcn)
    if n <= 2000 and n >= 3 and max_eigen == 2.0 and not gcn:
        msg += ' | ' + net_case(rng, row, col, n, q, w, L, S)
    return msg


def spmm_case(rng):
    """One fused SpMM call with random shape, feature width, operand strides, axpy operands (also aliasing
    the output), bias, ReLU + mask, against a complex128 computation on the host."""
    import scipy.sparse as sp
    from magnet_amd.laplacian import MagneticLaplacian
    n_rows = int(rng.choice([1, 5, 64, 700, 5000])); n_cols = int(rng.choice([1, 9, 300, 5000]))
    dens = float(rng.choice([0.0, 0.01, 0.2, 1.0]))
    nnz_target = min(int(dens * n_rows * n_cols) + int(rng.integers(0, 3)), 400_000)
    r = rng.integers(0, n_rows, nnz_target); c = rng.integers(0, n_cols, nnz_target)
    if n_rows >= 700 and rng.integers(0, 2):                # a hub row for the chunked path
        extra = int(rng.integers(1100, 5000)); r = np.concatenate([r, np.full(extra, rng.integers(0, n_rows))]); c = np.concatenate([c, rng.integers(0, n_cols, extra)])
    A = sp.coo_matrix(((rng.standard_normal(len(r)) + 1j * rng.standard_normal(len(r))).astype(np.complex64), (r, c)), shape=(n_rows, n_cols)).tocsr()
    A.sum_duplicates(); A.sort_indices()
    L = MagneticLaplacian.from_scipy(A)
    F = int(rng.choice([1, 2, 3, 4, 8, 12, 20, 64, 100, 256, 260, 1028, 1100]))
    if n_rows * F > 4_000_000 or n_cols * F > 4_000_000:
        F = 4
    pad = int(rng.choice([0, 0, 4, 5])); gap = int(rng.choice([0, 0, 8, 3]))

    def alloc(rows):                                         # row-planar with a padded row stride and a gap before the imaginary part
        im = F + gap; ld = im + F + pad
        buf = torch.randn(rows * ld + 8, device='cuda')
        return torch.as_strided(buf, (rows, 2, F), (ld, im, 1), int(rng.choice([0, 4])))
    X = alloc(n_cols); Y = alloc(n_rows)
    use1, use2 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    alias = use1 and bool(rng.integers(0, 3) == 0)
    Z1 = (Y if alias else alloc(n_rows)) if use1 else None
    Z2 = alloc(n_rows) if use2 else None
    alpha, b1, b2 = float(rng.choice([1.0, 2.0, -0.5])), float(rng.choice([1.0, -1.0, 0.25])), float(rng.choice([-1.0, 0.5]))
    bias = torch.randn(F, device='cuda') if rng.integers(0, 2) else None
    relu = bool(rng.integers(0, 2))
    mask = torch.empty((n_rows, F), dtype=torch.uint8, device='cuda') if relu and rng.integers(0, 2) else None
    cx = lambda t: t[:, 0, :].double().cpu().numpy() + 1j * t[:, 1, :].double().cpu().numpy()
    want = alpha * (A.astype(np.complex128) @ cx(X))
    if use1:
        want = want + b1 * cx(Z1)
    if use2:
        want = want + b2 * cx(Z2)
    if bias is not None:
        bb = bias.double().cpu().numpy(); want = want + bb + 1j * bb
    L.spmm(X, out=Y, alpha=alpha, z1=Z1, beta1=b1, z2=Z2, beta2=b2, bias=bias, relu=relu, mask=mask)
    got = cx(Y)
    scale = max(float(np.abs(want).max()) if want.size else 0.0, 1e-30)
    if relu:
        undecided = np.abs(want.real) < 1e-5 * scale
        m = want.real >= 0
        want = want * m
        got = np.where(undecided, want, got)
        if mask is not None:
            mk = mask.cpu().numpy().astype(bool)
            assert np.array_equal(mk[~undecided], m[~undecided]), 'mask'
    err = float(np.abs(got - want).max()) if want.size else 0.0
    if err > 1e-5 * scale + 1e-7:
        raise AssertionError('spmm: err %.3e scale %.3e (rows=%d cols=%d nnz=%d F=%d pad=%d gap=%d z1=%d alias=%d z2=%d bias=%d relu=%d)' % (
            err, scale, n_rows, n_cols, A.nnz, F, pad, gap, use1, alias, use2, bias is not None, relu))
    return 'spmm ok rows=%d cols=%d nnz=%d F=%d' % (n_rows, n_cols, A.nnz, F)


def aux_case(rng):
    """Layout / activation / row-staging kernels with random shapes and strides against torch."""
    from magnet_amd import laplacian as lapmod
    n = int(rng.choice([0, 1, 7, 300, 5000])); F = int(rng.choice([1, 2, 3, 4, 8, 20, 64, 130, 1028]))
    pad, gap = int(rng.choice([0, 4, 3])), int(rng.choice([0, 8, 5]))

    def alloc(rows):
        im = F + gap; ld = im + F + pad
        buf = torch.randn(rows * ld + 8, device='cuda')
        return torch.as_strided(buf, (rows, 2, F), (ld, im, 1), int(rng.choice([0, 4])))
    re = torch.randn(n, F + 3, device='cuda')[:, :F]; im_ = torch.randn(n, F, device='cuda')
    p = lapmod.pack_complex(re, im_, out=alloc(n))
    assert torch.equal(p[:, 0, :], re) and torch.equal(p[:, 1, :], im_), 'pack'
    r2, i2 = lapmod.unpack_complex(p)
    assert torch.equal(r2, re) and torch.equal(i2, im_), 'unpack'
    if n:
        bias = torch.randn(F, device='cuda')
        want = p.clone() + bias.view(1, 1, F)
        m = (want[:, 0, :] >= 0)
        want = want * m.unsqueeze(1)
        q_ = p.clone() if rng.integers(0, 2) else alloc(n).copy_(p)
        mask = M._bias_relu_(q_, bias, True)
        assert torch.equal(q_, want) and torch.equal(mask.bool(), m), 'bias_relu'
        g = alloc(n)
        g0 = g.clone()
        M._relu_bwd_(g, mask)
        assert torch.equal(g, g0 * m.unsqueeze(1)), 'relu_bwd'
        dst = alloc(n)
        assert M._relu_bwd(g0, mask, out=dst) is dst and torch.equal(dst, g0 * m.unsqueeze(1)), 'relu_bwd out of place'
        assert torch.equal(M._relu_bwd(g0.transpose(0, 1).contiguous().transpose(0, 1), mask), g0 * m.unsqueeze(1)), 'relu_bwd of a permuted gradient'
        k = int(rng.integers(0, n + 1))
        idx = torch.from_numpy(rng.permutation(n)[:k].astype(np.int32)).cuda()
        buf = lapmod.gather_rows(p, idx)
        assert torch.equal(buf, p[idx.long()]), 'gather_rows'
        tgt = alloc(n); t0 = tgt.clone()
        lapmod.scatter_rows(buf, idx, tgt)
        t0[idx.long()] = buf
        assert torch.equal(tgt, t0), 'scatter_rows'
    return 'aux ok n=%d F=%d' % (n, F)


def net_case(rng, row, col, n, q, w, L, S):
    """A whole ChebNet or ChebNet_Edge (random depth, width, order, activation) forward + backward against
    the oracle's restatement of the reference formulation (precomputed sparse T_k, four spmm per order) run
    under autograd on the CPU."""
    K = int(rng.integers(1, 4)); layer = int(rng.integers(1, 4)); F = int(rng.choice([1, 2, 4, 8, 16, 20]))
    cin = int(rng.choice([1, 2, 5, 16])); act = bool(rng.integers(0, 2)); edge = bool(rng.integers(0, 2))
    ncls = int(rng.integers(2, 5))
    Ts = O.cheb_poly_sparse(S, K)
    if sum(t.nnz for t in Ts) > 3_000_000:
        return 'net skipped (T_k too dense for the CPU oracle)'
    # the oracle runs in float64 on the float32 Laplacian values: the exact result a float32 implementation is held to
    Lr, Li = O.laplacian_lists(O.cheb_poly_sparse(S.astype(np.complex128), K), np.float64)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    torch.manual_seed(int(rng.integers(0, 2 ** 31)))
    cls = M.ChebNet_Edge if edge else M.ChebNet
    net = cls(cin, H.cheb_poly_sparse(L, K), None, num_filter=F, K=K, label_dim=ncls, activation=act, layer=layer, dropout=0.0).cuda()
    with torch.no_grad():
        for m in net.Chebs:
            if hasattr(m, 'bias'):
                m.bias.uniform_(-0.3, 0.3)
    state = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    Xr = torch.randn(n, cin, generator=g); Xi = torch.randn(n, cin, generator=g)
    if act:
        # A complex ReLU whose input real part is within float32 rounding of zero may legitimately open in one
        # computation and close in the other.  That is decided from the pre-activations themselves (float64), never
        # from the size of the final error.
        real, imag, j = Xr.double(), Xi.double(), 0
        with torch.no_grad():
            for li in range(layer):
                real, imag = O.cheb_conv(Lr, Li, state['Chebs.%d.weight' % j], state['Chebs.%d.bias' % j], real, imag)
                j += 1
                if not (edge and layer == 1 and li == 0):
                    if bool((real.abs() < 2e-5 * max(float(real.abs().max()), 1e-30)).any()):
                        return 'net skipped (a pre-activation real part within rounding of zero decides a gate)'
                    real, imag = O.complex_relu(real, imag)
                    j += 1
    if edge:
        m_pairs = int(rng.integers(1, 400))
        pairs = torch.from_numpy(rng.integers(0, n, (m_pairs, 2)))
        y = torch.from_numpy(rng.integers(0, ncls, m_pairs))
        out = net(Xr.cuda(), Xi.cuda(), pairs.cuda())
        ref = O.chebnet_edge_forward(state, Lr, Li, Xr.double(), Xi.double(), pairs, layer, act)
        loss, rloss = torch.nn.functional.nll_loss(out, y.cuda()), torch.nn.functional.nll_loss(ref, y)
    else:
        y = torch.from_numpy(rng.integers(0, ncls, (1, n)))
        out = net(Xr.cuda(), Xi.cuda())
        ref = O.chebnet_forward(state, Lr, Li, Xr.double(), Xi.double(), layer, act)
        loss, rloss = torch.nn.NLLLoss()(out, y.cuda()), torch.nn.NLLLoss()(ref, y)
    close(out.detach().cpu().double(), ref.detach(), 1e-5, 'net out')
    loss.backward(); rloss.backward()
    for k, p in net.named_parameters():
        gr = state[k].grad
        scale = max(float(gr.abs().max()), 1e-6)
        if float((p.grad.cpu().double() - gr).abs().max()) > 1e-5 * scale + 1e-7:
            raise AssertionError('net grad %s: err %.3e scale %.3e (K=%d layer=%d F=%d cin=%d act=%d edge=%d)' % (
                k, float((p.grad.cpu().double() - gr).abs().max()), scale, K, layer, F, cin, act, edge))
    return 'net ok K=%d layer=%d F=%d cin=%d act=%d edge=%d' % (K, layer, F, cin, act, edge)


def dist_case(rng):
    """The multi-GPU building blocks on a random graph, all ranks emulated in this process: per-rank row blocks
    (mh_lap_block_*) stacked against the whole-graph build (bit for bit), the halo plan kernels against their tensor
    restatement, and the recursion on compact [own | halo] buffers with a packed exchange against the single-GPU
    recursion."""
    from magnet_amd import dist as D
    from magnet_amd import laplacian as lapmod
    dev = torch.device('cuda', 0)
    n = int(rng.choice([1, 2, 5, 40, 200, 300, 3000, 9000, 30000]))      # 40, 200, 9000, 30000: sizes whose keys can carry tags
    e = int(rng.integers(0, 25 * n + 1))
    world = int(rng.integers(1, 6))
    row = torch.from_numpy(rng.integers(0, n, e)).to(dev); col = torch.from_numpy(rng.integers(0, n, e)).to(dev)
    if e > 4 and rng.integers(0, 2):
        k = e // 3
        row[:k], col[:k] = col[k:2 * k].clone(), row[k:2 * k].clone()            # reciprocal pairs
    if rng.integers(0, 3) == 0 and e > 200:
        row[-150:] = row[0]; col[-150:] = col[0]                                   # one directed pair 150 times: a long run
    kind = int(rng.integers(0, 3))
    w = None if kind == 0 else (torch.from_numpy(rng.integers(1, 9, e).astype(np.float32)).to(dev) if kind == 1
                                else torch.from_numpy(rng.random(e).astype(np.float32)).to(dev))
    if rng.integers(0, 2):
        row, col = row.int(), col.int()
    q = float(rng.choice([0.0, 0.1, 0.25, rng.random() * 0.5])); lam = float(rng.choice([2.0, 2.0, 3.0]))
    L = lapmod.build_magnetic_laplacian(row, col, n, q=q, edge_weight=w, max_eigen=lam)
    deg = torch.bincount(row.long(), minlength=n) + torch.bincount(col.long(), minlength=n)
    bounds = D.bounds_from_degrees(deg, world)
    use_counts = bool(rng.integers(0, 2))
    builds = []
    for r in range(world):
        r0, r1 = bounds[r], bounds[r + 1]
        fm, bm = (row >= r0) & (row < r1), (col >= r0) & (col < r1)
        fwd = (row[fm], col[fm]) + ((w[fm],) if w is not None else ())
        bwd = (row[bm], col[bm]) + ((w[bm],) if w is not None else ())
        # unweighted builds: half of the cases hand the ranks every node's record count (tagged keys where the size allows)
        builds.append(lapmod.LaplacianBlockBuild(fwd, bwd, n, r0, r1, max_eigen=lam,
                                                 node_records=deg.int() if (w is None and use_counts) else None))
    dinv = torch.cat([b.dinv_block for b in builds])
    assert torch.equal(dinv, L.dinv), 'dinv'
    blocks = [b.finish(dinv, q) for b in builds]
    make = lambda rp, ci, v, nc: lapmod.MagneticLaplacian(rp, ci, v, nc)
    parts = []
    for r, blk in enumerate(blocks):
        rp, ci, v = D.slice_rows(L.row_ptr, L.col_idx, L.val, bounds[r], bounds[r + 1])
        assert torch.equal(blk.row_ptr, rp) and torch.equal(blk.col_idx, ci) and torch.equal(blk.val, v), 'block %d of %d' % (r, world)
        bt = torch.tensor(bounds, dtype=torch.int32, device=dev)
        got, want = lapmod.halo_plan(ci, n, bounds[r], bounds[r + 1], bt), D.halo_plan_torch(ci, n, bounds[r], bounds[r + 1], bt)
        assert all(torch.equal(a, b) for a, b in zip(got, want)), 'halo plan of rank %d' % r
        parts.append(D.RowPartition(blk.row_ptr, blk.col_idx, blk.val, n, bounds, r, world, make,
                                    exchange_lists=lambda need, rank, world, group: [None] * world))
    for r, p in enumerate(parts):
        p.give = [None if o == r else parts[o].need[r] for o in range(world)]

    def exchange(bufs):
        for r, p in enumerate(parts):
            for o, other in enumerate(parts):
                if o != r and p.peer_ptr[o + 1] > p.peer_ptr[o]:
                    bufs[r][p.n_local + p.peer_ptr[o]:p.n_local + p.peer_ptr[o + 1]] = D.HipRowMover.gather(bufs[o], other.give[r])
    f = int(rng.choice([1, 3, 4, 16, 64])); terms = int(rng.integers(1, 5))
    x = torch.randn(n, 2, f, device=dev, generator=torch.Generator(device=dev).manual_seed(int(rng.integers(0, 2 ** 31))))
    ref = [x] + ([L.spmm(x)] if terms > 1 else [])
    for k in range(2, terms):
        ref.append(L.spmm(ref[k - 1], alpha=2.0, z1=ref[k - 2], beta1=-1.0))
    zs = []
    for p in parts:
        b = p.new_buffer(f, dev); b.fill_(float('nan')); p.block(b).copy_(x[p.r0:p.r1]); zs.append([b])
    exchange([z[0] for z in zs])
    for k in range(1, terms):
        new = [torch.full((p.n_buf, 2, f), float('nan'), device=dev) for p in parts]
        for r, p in enumerate(parts):
            p.propagate(zs[r][k - 1], new[r], 1.0 if k == 1 else 2.0, None if k == 1 else zs[r][k - 2], 0.0 if k == 1 else -1.0)
        exchange(new)
        for r in range(world):
            zs[r].append(new[r])
    scale = max(float(ref[-1].abs().max()), 1e-30)
    for r, p in enumerate(parts):
        if p.n_local:
            err = float((p.block(zs[r][-1]) - ref[-1][p.r0:p.r1]).abs().max())
            assert err <= 1e-5 * scale, 'recursion on rank %d: %.3e vs %.3e' % (r, err, scale)
    return 'dist ok n=%d e=%d world=%d w=%d F=%d terms=%d nnz=%d halo=%s' % (n, e, world, kind, f, terms, L.nnz, [p.n_halo for p in parts])


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=100)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--kind', choices=['graph', 'spmm', 'aux', 'dist'], default='graph')
    a = ap.parse_args()
    bad = 0
    for i in range(a.first, a.first + a.cases):
        rng = np.random.default_rng([a.seed, i])           # every case reproducible on its own (--first i --cases 1)
        try:
            msg = (spmm_case(rng) if a.kind == 'spmm' else aux_case(rng) if a.kind == 'aux' else dist_case(rng) if a.kind == 'dist'
                   else one_case(rng, i))
        except Exception as ex:                            # keep going: report every failing case
            import traceback
            bad += 1
            msg = 'FAILED %r\n%s' % (ex, traceback.format_exc(limit=3))
        if msg.startswith('FAILED') or i % 20 == 0:
            print(i, msg, flush=True)
    print('%d cases, %d failures' % (a.cases, bad), flush=True)
    sys.exit(1 if bad else 0)
