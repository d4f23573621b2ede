"""Drop-in for the reference's layer/sparse_magnet.py on MI355X.

Same names, constructor arguments, forward signatures and state_dict keys as
reference src/layer/sparse_magnet.py:8-165:

    process(mul_L_real, mul_L_imag, weight, X_real, X_imag) -> [2, N, C_out]
    ChebConv(in_c, out_c, K, L_norm_real, L_norm_imag, bias=True).forward((X_real, X_imag))
    complex_relu_layer().forward(real, img=None)
    ChebNet(in_c, L_norm_real, L_norm_imag, num_filter, K, label_dim, activation, layer, dropout)
        .forward(real, imag)                      -> [1, label_dim, N]
    ChebNet_Edge(... same ...).forward(real, imag, index) -> [M, label_dim]
    MagNetConv(in_c, out_c, K, q).forward(real, imag, edge_index, edge_weight=None)

`L_norm_real` / `L_norm_imag` may be
  * a ChebBasis (utils.hermitian.cheb_poly_sparse) or a MagneticLaplacian: the layer runs the
    feature-side recursion  Z_k = 2 L Z_{k-1} - Z_{k-2}  with the fused complex SpMM and never
    forms T_k (L_norm_imag is then ignored and may be None);
  * the reference's two lists of K+1 torch sparse tensors (src/sparse_Magnet.py:108-112): every
    T_k is converted once to a complex CSR and applied with the same kernel
    (sum_k (T_k X) W_k, exactly the reference's formula for arbitrary lists).

Between layers features stay in the row-planar layout [N, 2, F] (one tensor holding the real
and the imaginary block of every node); the (real, imag) tensors handed back are views of it.
The K+1 propagated matrices of a conv sit side by side in one block [N, 2, K+1, F] (each order a strided
view the SpMM works on in place), so the weights are applied by ONE GEMM per conv (torch.addmm, hipBLASLt)
instead of the reference's K+1 matmul pairs; inside a network a conv writes its output directly into
order 0 of the next conv's block.  All sparse work, bias and the complex ReLU run in libmagnet_hip.so.
GPU only.
"""
import math
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import ops, MagnetHipError
from ..laplacian import MagneticLaplacian, build_magnetic_laplacian, pack_complex, _packed_strides
from ..utils.hermitian import ChebBasis


# ----------------------------------------------------------------------------
# packed (row-planar) features
# ----------------------------------------------------------------------------
class _PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real, imag):
        return pack_complex(real, imag)

    @staticmethod
    def backward(ctx, g):
        return g[:, 0, :], g[:, 1, :]


class _PackIntoBlockFn(torch.autograd.Function):
    """(real, imag) packed straight into order 0 of a new side-by-side block [N, 2, k1, F]; returns that order-0
    view.  The conv that follows fills the other orders in place (no packed copy of the input in between)."""

    @staticmethod
    def forward(ctx, real, imag, k1):
        block = _new_block(real.size(0), k1, real.size(1), real.device)
        return pack_complex(real, imag, out=_order(block, 0))

    @staticmethod
    def backward(ctx, g):
        return g[:, 0, :], g[:, 1, :], None


def _as_packed(real, imag):
    """Row-planar tensor of a (real, imag) pair; free when the pair are views handed out by a layer here."""
    p = getattr(real, '_mh_packed', None)
    if p is not None and getattr(imag, '_mh_packed', None) is p:
        return p
    if not real.is_cuda:
        raise MagnetHipError('magnet_amd layers run on the GPU only (got %s tensors)' % real.device)
    return _PackFn.apply(real, imag)


def _views(p):
    """(real, imag) views of a row-planar tensor.  Each view remembers the tensor it was cut from, so a
    layer that is handed the pair back (the reference passes (real, imag) tuples between layers) finds the
    packed tensor again instead of packing a copy (_as_packed)."""
    real, imag = p[:, 0, :], p[:, 1, :]
    real._mh_packed = p
    imag._mh_packed = p
    return real, imag


def _rows2(p):
    """[N, 2, F] -> [2N, F] matrix for the weight GEMM (real and imaginary rows share the real weight).
    A view also for one order of a side-by-side block (row stride = twice the part stride)."""
    n, _, f = p.shape
    if p.is_contiguous():
        return p.view(n * 2, f)
    if p.stride(2) == 1 and p.stride(0) == 2 * p.stride(1):
        return p.as_strided((2 * n, f), (p.stride(1), 1), p.storage_offset())
    return p.reshape(n * 2, f)


# Orders side by side: the K+1 matrices T_k(L) X of one conv live in ONE block [N, 2, K+1, F]; order k is
# the strided view block[:, :, k, :] (the SpMM reads and writes such views in place), and the block seen as
# [2N, (K+1) F] is the left operand of a single GEMM with the weights stacked to [(K+1) F, C_out].  One
# GEMM per conv instead of K+1 reads the output once instead of K+1 times; measured on MI355X
# (N=2M, F=128, K=2): 3.6 ms against 6.0 ms, and at N=10M, F=256, K=3: 75 ms against 87 ms with the
# SpMM orders themselves 6 % faster on the interleaved block.
def _new_block(n, k1, f, device):
    return torch.empty((n, 2, k1, f), dtype=torch.float32, device=device)


def _order(block, k):
    return block[:, :, k, :]


def _block_rows(block):
    n, _, k1, f = block.shape
    return block.view(2 * n, k1 * f)


def _block_of(p, k1):
    """The side-by-side block whose order 0 is `p` — for a `p` that a conv here allocated that way
    (it asked for `next_terms = k1`); the other orders are free for the consumer to fill."""
    n, _, f = p.shape
    if k1 == 1:
        return p.detach().unsqueeze(2) if p.is_contiguous() else None
    if p.stride() != (2 * k1 * f, k1 * f, 1):
        return None
    return torch.as_strided(p.detach(), (n, 2, k1, f), (2 * k1 * f, k1 * f, f, 1), p.storage_offset())


def _alloc_out(n, f, next_terms, device):
    """Output [N, 2, F] of a conv: plain, or order 0 of the next conv's side-by-side block."""
    if next_terms and next_terms > 1:
        return _order(_new_block(n, next_terms, f, device), 0)
    return torch.empty((n, 2, f), dtype=torch.float32, device=device)


def _slabs(r, min_rows, most):
    """Largest power of two <= most that divides r into slabs of at least min_rows rows."""
    slabs = 1
    while slabs < most and r % (slabs * 2) == 0 and r // (slabs * 2) >= min_rows:
        slabs *= 2
    return slabs


def _gram(a, b):
    """a^T b for two tall matrices [R, m], [R, n] (R = 2N rows, m, n = channel counts): the weight
    gradient.  A single GEMM has only m*n/tile^2 output tiles to spread over 256 CUs, so the rows are
    cut into slabs that are multiplied as one batched GEMM and summed (split-K).  Graph-scale row counts take slabs
    of >= 4096 rows; a few thousand rows (the reference's own datasets, where an epoch is a string of
    microsecond kernels and a one-workgroup GEMM over all rows is the longest of them) slabs of >= 256."""
    r = a.size(0)
    if a.size(1) * b.size(1) <= (1 << 18):
        slabs = _slabs(r, 4096, 128) if r >= (1 << 16) else _slabs(r, 256, 32) if r >= 2048 else 1
        if slabs > 1:
            return torch.bmm(a.unflatten(0, (slabs, r // slabs)).transpose(1, 2),
                             b.unflatten(0, (slabs, r // slabs))).sum(0)
    return torch.mm(a.t(), b)


def _col_sums(g):
    """Sum of a packed gradient [N, 2, F] over nodes and parts -> [F]: the bias gradient (the same real bias sits on
    both parts, src/layer/sparse_magnet.py:72)."""
    return g.sum(dim=(0, 1))


def _bias_relu_(p, bias, relu):
    """In place: + bias on both parts, complex ReLU; returns the uint8 mask (or None)."""
    n, _, f = p.shape
    mask = torch.empty((n, f), dtype=torch.uint8, device=p.device) if relu else None
    if bias is None and not relu:
        return None
    b = None if bias is None else bias.detach().reshape(-1).float().contiguous()
    if n:
        ops.bias_relu_(p, b, bool(relu), mask)
    return mask


def _relu_bwd_(g, mask):
    if g.size(0):
        ops.relu_bwd_(g, mask)
    return g


def _relu_bwd(g, mask, out=None):
    """out = g * mask on both parts (backward of the complex ReLU), out of place: `g` is read where it lies
    (any row-planar strides), `out` is a new contiguous tensor or the given [N, 2, F] view — e.g. order 0 of a
    side-by-side block — so the incoming gradient is neither made contiguous nor cloned first."""
    n, _, f = g.shape
    if (f > 1 and g.stride(2) != 1) or (n > 1 and g.stride(0) < f) or g.stride(1) < f:
        g = g.contiguous()
    if out is None:
        out = torch.empty((n, 2, f), dtype=torch.float32, device=g.device)
    if n:
        ops.relu_bwd_out(g, mask, out)
    return out


# ----------------------------------------------------------------------------
# Chebyshev propagation with the fused SpMM
# ----------------------------------------------------------------------------
def _fill_orders(lap, block):
    """block[:, :, 0] holds x; fills block[:, :, k] = T_k(L) x for k >= 1 by the three-term recursion
    (K SpMMs)."""
    k1 = block.size(2)
    if k1 > 1:
        lap.spmm(_order(block, 0), out=_order(block, 1))
    for k in range(2, k1):
        lap.spmm(_order(block, k - 1), out=_order(block, k), alpha=2.0, z1=_order(block, k - 2), beta1=-1.0)
    return block


def _cheb_terms(lap, x0, n_terms):
    """[N, 2, n_terms, F] block of T_0 x0 ... T_{n-1} x0 (one copy of x0, K SpMMs)."""
    n, _, f = x0.shape
    block = _new_block(n, n_terms, f, x0.device)
    _order(block, 0).copy_(x0)
    return _fill_orders(lap, block)


def _clenshaw(lap, ys, bias=None, relu=False, n_terms=None, out=None):
    """sum_k T_k(L) y_k by Clenshaw's recurrence (K SpMMs); bias / complex ReLU fused into the
    last SpMM.  `ys` is a list, a block [N, 2, K+1, F], or a callable k -> y_k with `n_terms` given
    (each y_k is then made only when the recurrence reaches it, and dropped right after).  `out`:
    where to write the result (may be a strided [N, 2, F] view).  Returns (result, mask)."""
    if torch.is_tensor(ys):
        block = ys
        ys = [_order(block, k) for k in range(block.size(2))]
    get = ys if callable(ys) else (lambda k: ys[k])
    if n_terms is None:
        n_terms = len(ys)
    y_last = get(n_terms - 1)
    n, _, f = y_last.shape
    mask = torch.empty((n, f), dtype=torch.uint8, device=y_last.device) if relu else None
    if n_terms == 1:
        if out is None:
            out = torch.empty((n, 2, f), dtype=torch.float32, device=y_last.device)
        out.copy_(y_last)
        m = _bias_relu_(out, bias, relu)
        return out, m
    b2 = None                     # b_{k+2}
    b1 = y_last                   # b_{k+1}, starts as b_K = y_K
    y_last = None
    for k in range(n_terms - 2, 0, -1):
        nb = lap.spmm(b1, alpha=2.0, z1=get(k), beta1=1.0, z2=b2, beta2=-1.0)
        b2, b1 = b1, nb
    out = lap.spmm(b1, out=out, alpha=1.0, z1=get(0), beta1=1.0, z2=b2, beta2=-1.0, bias=bias, relu=relu, mask=mask)
    return out, mask


# In the backward of the SpMM-first ordering the K+1 matrices G W_k^T come from one GEMM when the block
# that holds them is at most this large; above it they are formed one at a time inside Clenshaw's
# recurrence (a 10M-node, 256-feature, K=3 block would be 80 GB).
FUSED_BACKWARD_BYTES = 8 << 30

# Memory for time at graph scale: a conv whose block of saved orders would exceed this many bytes keeps only
# its input for the backward pass and recomputes T_k(L) X there (K more SpMMs per conv and training step).
# 0 = never (default); e.g. 32 GiB lets a 3-layer, 256-filter, K=3 net train on the 10M-node graph on one GPU.
RECOMPUTE_ORDERS_BYTES = int(os.environ.get('MH_RECOMPUTE_ORDERS_BYTES', '0'))


class _ChebConvFn(torch.autograd.Function):
    """out = sum_k T_k(L) X W_k + bias (+ complex ReLU), L Hermitian.

    SpMM-first (C_in <= C_out): the orders Z_k = T_k X side by side in one block, then ONE GEMM with
    the stacked weights.  GEMM-first (C_out < C_in): Y = X [W_0 ... W_K] in one GEMM, then Clenshaw on
    the narrower orders of Y.  Backward uses T_k^H = T_k: dX = sum_k T_k (G W_k^T),
    dW_k = (T_k X)^T G = X^T (T_k G).

    `in_block`: a block whose order 0 is xp and whose other orders this conv may fill (or that are
    filled already: `orders_ready`).  `next_terms`: allocate the output as order 0 of a block with that
    many orders, for the conv that follows."""

    @staticmethod
    def forward(ctx, xp, weight, bias, lap, relu, in_block=None, orders_ready=False, next_terms=0):
        n, _, cin = xp.shape
        k1, _, cout = weight.shape
        w = weight.detach()
        b = None if bias is None else bias.detach()
        if in_block is not None and (tuple(in_block.shape) != (n, 2, k1, cin) or not in_block.is_contiguous()
                                     or in_block.data_ptr() != xp.data_ptr()):
            raise MagnetHipError('in_block must be the contiguous [N, 2, K+1, C_in] block whose order 0 is the input')
        gemm_first = cout < cin and in_block is None
        out = _alloc_out(n, cout, next_terms, xp.device)
        if gemm_first:
            # one GEMM makes all K+1 matrices X W_k side by side; the SpMM gathers whole rows of an order
            ys = torch.mm(_rows2(xp), w.permute(1, 0, 2).reshape(cin, k1 * cout)).view(n, 2, k1, cout)
            out, mask = _clenshaw(lap, ys, b, relu, out=out)
            block = None
        else:
            if in_block is None:
                block = _cheb_terms(lap, xp, k1)
            else:
                block = in_block if orders_ready else _fill_orders(lap, in_block)
            o2 = _rows2(out)
            if b is None or relu:
                # with an activation to apply the bias joins its in-place pass: addmm would first broadcast the bias into
                # the output (one more write of it) and have the GEMM read that back
                torch.mm(_block_rows(block), w.reshape(k1 * cin, cout), out=o2)
                mask = _bias_relu_(out, b, relu)
            else:                                          # no activation pass: the bias rides on the GEMM
                torch.addmm(b.reshape(1, cout), _block_rows(block), w.reshape(k1 * cin, cout), out=o2)
                mask = None
        ctx.x0 = None
        if (block is not None and not orders_ready and RECOMPUTE_ORDERS_BYTES > 0
                and 8 * n * k1 * cin > RECOMPUTE_ORDERS_BYTES and ctx.needs_input_grad[1]):
            # keep the input only (a compact copy when it is order 0 of a handed-over block, which would
            # otherwise pin the whole block) and let the block go
            ctx.x0 = xp if in_block is None else xp.clone(memory_format=torch.contiguous_format)
            block = None
        ctx.lap, ctx.relu, ctx.gemm_first, ctx.has_bias = lap, relu, gemm_first, bias is not None
        ctx.mask = mask
        ctx.block = block
        ctx.shape = (n, cin)
        ctx.save_for_backward(xp if gemm_first else None, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        xp, weight = ctx.saved_tensors
        lap = ctx.lap
        n, cin = ctx.shape
        k1, _, cout = weight.shape
        w = weight.detach()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = dw = db = None
        if ctx.gemm_first:
            # T_k G side by side; G (times the ReLU mask) goes straight into order 0
            ds = _new_block(n, k1, cout, g.device)
            g0 = _order(ds, 0)
            if ctx.relu:
                _relu_bwd(g, ctx.mask, out=g0)
            else:
                g0.copy_(g)
            if need_b:
                db = _col_sums(g0).view(1, cout)          # the same real bias sits on both parts
            if need_w or need_x:
                _fill_orders(lap, ds)
            if need_w:                                     # X^T [T_0 G ... T_K G] -> [C_in, K+1, C_out]
                dw = _gram(_rows2(xp), _block_rows(ds)).view(cin, k1, cout).permute(1, 0, 2).contiguous()
            if need_x:
                dx = torch.mm(_block_rows(ds), w.transpose(1, 2).reshape(k1 * cout, cin)).view(n, 2, cin)
        else:
            g = _relu_bwd(g, ctx.mask) if ctx.relu else g.contiguous()
            if need_b:
                db = _col_sums(g).view(1, cout)
            g2 = _rows2(g)
            if need_w:
                block = ctx.block if ctx.block is not None else _cheb_terms(lap, ctx.x0, k1)   # recomputed orders
                dw = _gram(_block_rows(block), g2).view(k1, cin, cout)
                block = None
            ctx.block = ctx.x0 = None                      # the saved orders are not needed past this point
            if need_x:
                if 8 * n * k1 * cin <= FUSED_BACKWARD_BYTES:
                    ys = torch.mm(g2, w.permute(2, 0, 1).reshape(cout, k1 * cin)).view(n, 2, k1, cin)
                    dx, _ = _clenshaw(lap, ys)
                else:
                    # G W_k^T is formed when the recurrence reaches order k, not for all orders up front
                    dx, _ = _clenshaw(lap, lambda k: torch.mm(g2, w[k].t()).view(n, 2, cin), n_terms=k1)
        return dx, dw, db, None, None, None, None, None


class _ChebTermsFn(torch.autograd.Function):
    """x [N, 2, F] -> the orders T_k(L) x side by side, [N, 2, K+1, F] (K SpMMs).  Backward:
    dx = sum_k T_k dZ_k by Clenshaw (T_k Hermitian).  Building block of the multi-GPU layers, where the
    weight GEMM is applied after a regrouping of the orders."""

    @staticmethod
    def forward(ctx, x, lap, n_terms):
        ctx.lap = lap
        return _cheb_terms(lap, x, n_terms)

    @staticmethod
    def backward(ctx, dz):
        dx, _ = _clenshaw(ctx.lap, dz.contiguous())
        return dx, None, None


class _ClenshawFn(torch.autograd.Function):
    """ycat [N, 2, K+1, F] -> sum_k T_k(L) ycat[:, :, k] + bias (+ complex ReLU) (K SpMMs, epilogue
    fused).  Backward: dY_k = T_k G by the forward recursion."""

    @staticmethod
    def forward(ctx, ycat, bias, lap, relu):
        k1 = ycat.size(2)
        out, mask = _clenshaw(lap, ycat.contiguous(), None if bias is None else bias.detach(), relu)
        ctx.lap, ctx.relu, ctx.mask, ctx.k1, ctx.has_bias = lap, relu, mask, k1, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        g = _relu_bwd(g, ctx.mask) if ctx.relu else g.contiguous()
        db = _col_sums(g) if ctx.has_bias and ctx.needs_input_grad[1] else None
        dy = _cheb_terms(ctx.lap, g, ctx.k1) if ctx.needs_input_grad[0] else None
        return dy, db, None, None


class _ListConvFn(torch.autograd.Function):
    """out = sum_k (T_k X) W_k + bias (+ complex ReLU) for an arbitrary list of complex CSR matrices
    (the reference's calling convention).  Backward applies the conjugate transposes."""

    @staticmethod
    def forward(ctx, xp, weight, bias, laps, relu):
        n, _, cin = xp.shape
        k1, _, cout = weight.shape
        w = weight.detach()[:len(laps)]
        rows = laps[0].n_rows
        block = _new_block(rows, len(laps), cin, xp.device)         # T_k X side by side, one GEMM
        for k, lap in enumerate(laps):
            lap.spmm(xp, out=_order(block, k))
        out = torch.mm(_block_rows(block), w.reshape(len(laps) * cin, cout)).view(rows, 2, cout)
        ctx.mask = _bias_relu_(out, None if bias is None else bias.detach(), relu)
        ctx.laps, ctx.relu, ctx.block, ctx.has_bias = laps, relu, block, bias is not None
        ctx.shape = (n, cin)
        ctx.save_for_backward(weight)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, = ctx.saved_tensors
        n, cin = ctx.shape
        k1, _, cout = weight.shape
        nl = len(ctx.laps)
        w = weight.detach()
        g = _relu_bwd(g, ctx.mask) if ctx.relu else g.contiguous()
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _col_sums(g).view(1, cout)
        g2 = _rows2(g)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            dw[:nl] = _gram(_block_rows(ctx.block), g2).view(nl, cin, cout)
        ctx.block = None
        if ctx.needs_input_grad[0]:
            for k, lap in enumerate(ctx.laps):
                h = torch.mm(g2, w[k].t()).view(g.size(0), 2, cin)
                dx = lap.adjoint().spmm(h, z1=dx, beta1=1.0 if dx is not None else 0.0)
        return dx, dw, db, None, None


class _HeadFn(torch.autograd.Function):
    """Node-classification head: x [N, D] -> logits [1, C, N] = (x W^T + b)^T for a few classes (W [C, D]) — the
    reference's Conv1d(2F, label_dim, kernel_size=1) on the [1, 2F, N] features (src/layer/sparse_magnet.py:113-117,
    126-128), written class-major as log_softmax(dim=1) and the loss read it (no transposed copy).  Backward: the
    weight gradient g x is a product of two tall matrices with a tiny result, taken as the split-K gram the convs use
    (the library GEMM for a 5 x 256 result over 2M rows ran at a sixth of the rate the read of x allows); the bias
    gradient is a row sum of the class-major gradient (a column sum of an [N, C] matrix keeps 8 workgroups busy)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        out = torch.mm(weight, x.t()) if bias is None else torch.addmm(bias.view(-1, 1), weight, x.t())
        return out.unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.squeeze(0).contiguous()                                  # [C, N]
        dx = torch.mm(g.t(), weight) if ctx.needs_input_grad[0] else None
        dw = _gram(g.t(), x) if ctx.needs_input_grad[1] else None
        db = g.sum(1) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class _ComplexReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        out = p.clone()
        ctx.mask = _bias_relu_(out, None, True)
        return out

    @staticmethod
    def backward(ctx, g):
        return _relu_bwd(g, ctx.mask)


# ----------------------------------------------------------------------------
# pair gather of the link-prediction head, written straight into the concatenated layout
# ----------------------------------------------------------------------------
class _PairPlan:
    """Selection matrices of a pair list `index` [M, 2] as unit-valued CSRs for the SpMM kernel:
    pick[s] (M x N, one entry per row) gathers node index[:, s]; scatter[s] (N x M, its transpose,
    pairs listed in ascending order per node) adds pair gradients back per node without atomics."""

    def __init__(self, index, n):
        dev = index.device
        m = index.size(0)
        ones = torch.zeros((m, 2), dtype=torch.float32, device=dev)
        ones[:, 0] = 1.0
        arange = torch.arange(m + 1, dtype=torch.int32, device=dev)
        self.pick, self.scatter = [], []
        for s in (0, 1):
            nodes = index[:, s].long()
            if m and (int(nodes.min()) < 0 or int(nodes.max()) >= n):
                raise MagnetHipError('pair index out of range [0, %d)' % n)
            self.pick.append(MagneticLaplacian(arange, nodes.to(torch.int32), ones, n))
            order = torch.argsort(nodes, stable=True)
            rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            rp[1:] = torch.cumsum(torch.bincount(nodes, minlength=n), 0)
            self.scatter.append(MagneticLaplacian(rp.to(torch.int32), order.to(torch.int32), ones, m))
        self.m, self.n = m, n


# Selection matrices of the pair lists seen so far.  An entry lives exactly as long as its index tensor: the
# tensor is held weakly and its finalizer drops the entry, so nothing is pinned once the caller lets go.
_PAIR_CACHE = {}


def _pair_plan(index, n):
    key = (id(index), n)
    hit = _PAIR_CACHE.get(key)
    if hit is not None and hit[0]() is index and hit[1] == index._version:
        return hit[2]
    plan = _PairPlan(index, n)
    if hit is None:
        weakref.finalize(index, _PAIR_CACHE.pop, key, None)
    _PAIR_CACHE[key] = (weakref.ref(index), index._version, plan)
    return plan


def _pair_views(x, f):
    """[M, 4F] = [re_i | re_j | im_i | im_j]  ->  the i-slot and the j-slot as row-planar [M, 2, F] views."""
    m = x.size(0)
    return (torch.as_strided(x, (m, 2, f), (4 * f, 2 * f, 1), x.storage_offset()),
            torch.as_strided(x, (m, 2, f), (4 * f, 2 * f, 1), x.storage_offset() + f))


class _PairGatherFn(torch.autograd.Function):
    """cat(real[i], real[j], imag[i], imag[j]) of reference src/layer/sparse_magnet.py:159 from the
    row-planar trunk output, in one pass per slot; backward is a per-node sum in fixed order."""

    @staticmethod
    def forward(ctx, p, index):
        p = p.contiguous()
        n, _, f = p.shape
        plan = _pair_plan(index, n)
        out = torch.empty((plan.m, 4 * f), dtype=torch.float32, device=p.device)
        yi, yj = _pair_views(out, f)
        if plan.m:
            plan.pick[0].spmm(p, out=yi)
            plan.pick[1].spmm(p, out=yj)
        ctx.plan, ctx.f = plan, f
        return out

    @staticmethod
    def backward(ctx, g):
        plan, f = ctx.plan, ctx.f
        g = g.contiguous()
        dp = torch.zeros((plan.n, 2, f), dtype=torch.float32, device=g.device)
        if plan.m:
            gi, gj = _pair_views(g, f)
            plan.scatter[0].spmm(gi, out=dp)
            plan.scatter[1].spmm(gj, out=dp, z1=dp, beta1=1.0)
        return dp, None


# ----------------------------------------------------------------------------
# Laplacian argument handling
# ----------------------------------------------------------------------------
# Complex CSRs made from the reference's (real, imag) COO pairs.  The pair is held weakly: an entry (and the CSR
# copy it owns) goes away with the first of the two tensors, so the cache never keeps a Laplacian alive.
_COO_CACHE = {}


def _lap_from_pair(l_real, l_imag):
    """Complex CSR of one (real, imag) pair of torch sparse tensors, cached per tensor pair."""
    key = (id(l_real), id(l_imag))
    hit = _COO_CACHE.get(key)
    if hit is not None and hit[0]() is l_real and hit[1]() is l_imag:
        return hit[2]
    lap = MagneticLaplacian.from_coo_pair(l_real, l_imag)
    if hit is None:
        weakref.finalize(l_real, _COO_CACHE.pop, key, None)
        weakref.finalize(l_imag, _COO_CACHE.pop, key, None)
    _COO_CACHE[key] = (weakref.ref(l_real), weakref.ref(l_imag), lap)
    return lap


def _resolve_laplacian(L_norm_real, L_norm_imag):
    """-> ('cheb', MagneticLaplacian, n_terms or None) or ('list', [MagneticLaplacian...], n_terms)"""
    if isinstance(L_norm_real, ChebBasis):
        return 'cheb', L_norm_real.laplacian, len(L_norm_real)
    if isinstance(L_norm_real, MagneticLaplacian):
        return 'cheb', L_norm_real, None
    if isinstance(L_norm_real, (list, tuple)) and all(isinstance(l, MagneticLaplacian) for l in L_norm_real):
        return 'list', list(L_norm_real), len(L_norm_real)
    if isinstance(L_norm_real, (list, tuple)) and isinstance(L_norm_imag, (list, tuple)):
        if len(L_norm_real) != len(L_norm_imag):
            raise MagnetHipError('L_norm_real and L_norm_imag differ in length')
        return 'list', None, len(L_norm_real)          # converted lazily (tensors may still move to the GPU)
    raise MagnetHipError('unsupported Laplacian argument of type %s' % type(L_norm_real))


# ----------------------------------------------------------------------------
# Reference API
# ----------------------------------------------------------------------------
def process(mul_L_real, mul_L_imag, weight, X_real, X_imag):
    """One Chebyshev term (reference src/layer/sparse_magnet.py:8-18): stack([real, imag]) with
    real = (L_re X_re - L_im X_im) W, imag = (L_im X_re + L_re X_im) W."""
    lap = mul_L_real if isinstance(mul_L_real, MagneticLaplacian) else _lap_from_pair(mul_L_real, mul_L_imag)
    xp = _as_packed(X_real, X_imag)
    out = _ListConvFn.apply(xp, weight.unsqueeze(0), None, [lap], False)
    return out.permute(1, 0, 2)


class ChebConv(nn.Module):
    """The MagNet convolution (reference src/layer/sparse_magnet.py:21-72).

    weight [K+1, in_c, out_c] ~ U(-1/sqrt(out_c), 1/sqrt(out_c)), bias [1, out_c] zeros, the same
    real bias added to both parts."""

    def __init__(self, in_c, out_c, K, L_norm_real, L_norm_imag, bias=True):
        super(ChebConv, self).__init__()
        self.mul_L_real = L_norm_real
        self.mul_L_imag = L_norm_imag
        self._mode, self._lap, n_terms = _resolve_laplacian(L_norm_real, L_norm_imag)
        self._n_terms = K + 1 if n_terms is None else n_terms
        self.weight = nn.Parameter(torch.Tensor(K + 1, in_c, out_c))
        stdv = 1. / math.sqrt(self.weight.size(-1))
        self.weight.data.uniform_(-stdv, stdv)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(1, out_c))
            nn.init.zeros_(self.bias)
        else:
            self.register_parameter("bias", None)

    def _laps(self):
        if self._lap is None:                          # reference-style lists, converted on first use
            self._lap = [_lap_from_pair(r, i) for r, i in zip(self.mul_L_real, self.mul_L_imag)]
        return self._lap

    # The propagated input features T_k(L) X of a network's FIRST layer do not depend on any weight:
    # in full-batch training they are the same in every forward of every epoch.  With this switch on,
    # ChebNet / ChebNet_Edge keep them (K+1 matrices of the input width) for as long as the same input
    # tensors are passed, and skip K SpMMs per forward.  OFF by default, as the reference recomputes on every
    # forward: the cache is keyed on tensor identity and version counter, which a write through `.data` or
    # `set_()` does not advance, and it pins the input and its K+1 orders for the module's lifetime.  Callers
    # that own their input tensors and never write them in place (the drivers, bench.py) switch it on:
    # `ChebConv.cache_input_propagation = True` for all layers, or on one first layer.  A block larger than
    # RECOMPUTE_ORDERS_BYTES (when that budget is set) is never kept.
    cache_input_propagation = False

    def _spmm_first(self):
        """The ordering a conv in 'cheb' mode takes for a plain input: orders first when C_in <= C_out."""
        return self._mode == 'cheb' and self.weight.size(2) >= self.weight.size(1)

    def forward_first(self, real, imag, relu=False, next_terms=0):
        """First layer of a network: like forward_packed on (real, imag), reusing the packed input
        and its propagated orders when the inputs are the tensors (same objects, same version) seen last
        time and need no gradient."""
        usable = (self.cache_input_propagation and self._mode == 'cheb' and not real.requires_grad
                  and not imag.requires_grad and getattr(real, '_mh_packed', None) is None
                  and not (0 < RECOMPUTE_ORDERS_BYTES < 8 * real.size(0) * self._n_terms * real.size(1)))
        if not usable:
            if (self._mode == 'cheb' and self._spmm_first() and self._n_terms > 1 and real.is_cuda and real.dim() == 2
                    and real.shape == imag.shape and getattr(real, '_mh_packed', None) is None
                    and not (0 < RECOMPUTE_ORDERS_BYTES < 8 * real.size(0) * self._n_terms * real.size(1))):
                # the input goes straight into order 0 of the block this conv propagates in
                xp = _PackIntoBlockFn.apply(real, imag, self._n_terms)
                return self.forward_packed(xp, relu, next_terms, in_block=_block_of(xp, self._n_terms))
            return self.forward_packed(_as_packed(real, imag), relu, next_terms)
        c = getattr(self, '_input_cache', None)
        w = self.weight[:self._n_terms]
        if not (c is not None and c[0] is real and c[1] is imag and c[2] == real._version and c[3] == imag._version
                and c[6] == self._n_terms):
            if self._spmm_first():
                block = _new_block(real.size(0), self._n_terms, real.size(1), real.device)
                xp = pack_complex(real, imag, out=_order(block, 0))
                _fill_orders(self._laps(), block)
            else:
                block, xp = None, pack_complex(real, imag)
            c = (real, imag, real._version, imag._version, xp, block, self._n_terms)
            self._input_cache = c
        return _ChebConvFn.apply(c[4], w, self.bias, self._laps(), relu, c[5], c[5] is not None, next_terms)

    def forward_packed(self, xp, relu=False, next_terms=0, in_block=None):
        """Row-planar in, row-planar out; `relu` fuses the following complex ReLU.  `next_terms` > 1 makes
        the output order 0 of a side-by-side block for the conv that follows; `in_block` is such a block
        around `xp`, handed over by the conv before."""
        if self._n_terms > self.weight.size(0):
            raise MagnetHipError('%d Laplacian terms but weight has %d slices' % (self._n_terms, self.weight.size(0)))
        w = self.weight[:self._n_terms]
        if self._mode == 'cheb':
            return _ChebConvFn.apply(xp, w, self.bias, self._laps(), relu, in_block, False, next_terms)
        return _ListConvFn.apply(xp.contiguous(), w, self.bias, self._laps(), relu)

    def forward(self, data):
        return _views(self.forward_packed(_as_packed(data[0], data[1])))


class complex_relu_layer(nn.Module):
    """mask = (real >= 0) applied to both parts (reference src/layer/sparse_magnet.py:75-91)."""

    def __init__(self, ):
        super(complex_relu_layer, self).__init__()

    def complex_relu(self, real, img):
        return _views(_ComplexReluFn.apply(_as_packed(real, img)))

    def forward(self, real, img=None):
        if img is None:
            img = real[1]
            real = real[0]
        return self.complex_relu(real, img)


def _run_trunk(chebs, real, imag):
    """nn.Sequential of ChebConv / complex_relu_layer with each ReLU fused into the conv before it, and
    each conv writing its output where the next conv wants its order 0."""
    mods = list(chebs)
    p = None
    block = None                   # side-by-side block around p, made for the conv that consumes p
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, ChebConv):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], complex_relu_layer)
            step = 2 if fuse else 1
            nxt = mods[i + step] if i + step < len(mods) else None
            next_terms = nxt._n_terms if isinstance(nxt, ChebConv) and nxt._spmm_first() else 0
            if next_terms and 0 < RECOMPUTE_ORDERS_BYTES < 8 * real.size(0) * next_terms * nxt.weight.size(1):
                next_terms = 0     # over the memory budget: blocks live one at a time, so none is allocated ahead
            if p is None:
                p = m.forward_first(real, imag, relu=fuse, next_terms=next_terms)
            else:
                p = m.forward_packed(p, relu=fuse, next_terms=next_terms, in_block=block)
            block = _block_of(p, next_terms) if next_terms > 1 else None
            i += step
        elif isinstance(m, complex_relu_layer):
            p = _ComplexReluFn.apply(_as_packed(real, imag) if p is None else p)
            block = None
            i += 1
        else:
            real2, imag2 = m((real, imag) if p is None else _views(p))
            p = _as_packed(real2, imag2)
            block = None
            i += 1
    return _as_packed(real, imag) if p is None else p


class ChebNet(nn.Module):
    """Node-classification network (reference src/layer/sparse_magnet.py:94-130)."""

    def __init__(self, in_c, L_norm_real, L_norm_imag, num_filter=2, K=2, label_dim=2, activation=False, layer=2,
                 dropout=False):
        super(ChebNet, self).__init__()
        chebs = [ChebConv(in_c=in_c, out_c=num_filter, K=K, L_norm_real=L_norm_real, L_norm_imag=L_norm_imag)]
        if activation:
            chebs.append(complex_relu_layer())
        for i in range(1, layer):
            chebs.append(ChebConv(in_c=num_filter, out_c=num_filter, K=K, L_norm_real=L_norm_real,
                                  L_norm_imag=L_norm_imag))
            if activation:
                chebs.append(complex_relu_layer())
        self.Chebs = torch.nn.Sequential(*chebs)
        last_dim = 2
        self.Conv = nn.Conv1d(num_filter * last_dim, label_dim, kernel_size=1)
        self.dropout = dropout

    def trunk(self, real, imag):
        """The convolutions and activations: row-planar [N, 2, F] output of the last one.  Nothing in it depends
        on train / eval mode (dropout acts after it), so one trunk pass can serve an evaluation forward and the
        training forward that follows it under the same weights (magnet_amd.train.SharedForwardEpochs)."""
        return _run_trunk(self.Chebs, real, imag)

    def head(self, p):
        """unwind -> dropout -> Conv1d(2F, label_dim, 1) -> log_softmax of a trunk output."""
        x = p.reshape(p.size(0), -1)                   # == cat((real, imag), -1): same memory
        if self.dropout > 0:
            x = F.dropout(x, self.dropout, training=self.training)
        # Conv1d with kernel 1 over [1, 2F, N] is a linear map per node
        x = _HeadFn.apply(x, self.Conv.weight.squeeze(-1), self.Conv.bias)          # [1, label_dim, N]
        return F.log_softmax(x, dim=1)

    def forward(self, real, imag):
        return self.head(self.trunk(real, imag))


class ChebNet_Edge(nn.Module):
    """Link-prediction network (reference src/layer/sparse_magnet.py:133-165); `index` [M, 2] are
    the node pairs to score."""

    def __init__(self, in_c, L_norm_real, L_norm_imag, num_filter=2, K=2, label_dim=2, activation=False, layer=2,
                 dropout=False):
        super(ChebNet_Edge, self).__init__()
        chebs = [ChebConv(in_c=in_c, out_c=num_filter, K=K, L_norm_real=L_norm_real, L_norm_imag=L_norm_imag)]
        if activation and (layer != 1):
            chebs.append(complex_relu_layer())
        for i in range(1, layer):
            chebs.append(ChebConv(in_c=num_filter, out_c=num_filter, K=K, L_norm_real=L_norm_real,
                                  L_norm_imag=L_norm_imag))
            if activation:
                chebs.append(complex_relu_layer())
        self.Chebs = torch.nn.Sequential(*chebs)
        last_dim = 2
        self.linear = nn.Linear(num_filter * last_dim * 2, label_dim)
        self.dropout = dropout

    def trunk(self, real, imag):
        """The convolutions and activations (independent of train / eval mode and of the query pairs)."""
        return _run_trunk(self.Chebs, real, imag)

    def head(self, p, index):
        """pair gather -> dropout -> Linear(4F, label_dim) -> log_softmax of a trunk output."""
        x = _PairGatherFn.apply(p, index)              # == cat(real[i], real[j], imag[i], imag[j])
        if self.dropout > 0:
            x = F.dropout(x, self.dropout, training=self.training)
        x = self.linear(x)
        return F.log_softmax(x, dim=1)

    def forward(self, real, imag, index):
        return self.head(self.trunk(real, imag), index)


class MagNetConv(nn.Module):
    """ChebConv that takes the graph at call time: forward(real, imag, edge_index, edge_weight=None).
    The magnetic Laplacian of (edge_index, edge_weight, q) is built on the GPU and cached until a
    different graph tensor is passed."""

    def __init__(self, in_c, out_c, K, q=0.25, bias=True):
        super(MagNetConv, self).__init__()
        self.K, self.q = K, q
        self.weight = nn.Parameter(torch.Tensor(K + 1, in_c, out_c))
        stdv = 1. / math.sqrt(out_c)
        self.weight.data.uniform_(-stdv, stdv)
        if bias:
            self.bias = nn.Parameter(torch.zeros(1, out_c))
        else:
            self.register_parameter("bias", None)
        self._cache = None

    def laplacian(self, edge_index, edge_weight, n):
        c = self._cache
        ver = (edge_index._version, None if edge_weight is None else edge_weight._version)
        if c is not None and c[0] is edge_index and c[1] is edge_weight and c[2] == n and c[4] == ver:
            return c[3]
        lap = build_magnetic_laplacian(edge_index[0], edge_index[1], n, q=self.q, edge_weight=edge_weight,
                                       device=edge_index.device if edge_index.is_cuda else None)
        self._cache = (edge_index, edge_weight, n, lap, ver)      # same tensors, unmodified -> same graph
        return lap

    def forward(self, real, imag, edge_index, edge_weight=None):
        lap = self.laplacian(edge_index, edge_weight, real.size(0))
        return _views(_ChebConvFn.apply(_as_packed(real, imag), self.weight, self.bias, lap, False, None, False, 0))
