"""Device-resident magnetic Laplacian (complex CSR) and the calls into libmagnet_hip.so
that build it and apply it.

`build_magnetic_laplacian` is the GPU counterpart of hermitian_decomp_sparse
(reference src/utils/hermitian.py:136-168); `MagneticLaplacian.spmm` is one order of the
propagation (the four torch.spmm of reference src/layer/sparse_magnet.py:9-17 in one pass).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, require_cuda, MagnetHipError


# Optional observer of SpMM launches: set to a callable(start_event, end_event, info) to have every
# spmm() bracketed by HIP events on the stream it launches on (used by bench.py for per-kernel times).
SPMM_OBSERVER = None


def _packed_strides(t, what):
    """(ld, im) of a row-planar complex tensor [N, 2, F] (last dim contiguous)."""
    if t.dim() != 3 or t.size(1) != 2 or t.dtype != torch.float32:
        raise MagnetHipError('%s must be float32 [N, 2, F], got %s %s' % (what, t.dtype, tuple(t.shape)))
    if t.size(2) > 1 and t.stride(2) != 1:
        raise MagnetHipError('%s: feature dimension must be contiguous' % what)
    return t.stride(0), t.stride(1)


def pack_complex(real, imag, out=None):
    """(real [N,F], imag [N,F]) -> row-planar float32 [N, 2, F] on the GPU (mh_pack)."""
    require_cuda(real, imag)
    if real.shape != imag.shape or real.dim() != 2:
        raise MagnetHipError('real/imag must be [N, F] of equal shape, got %s and %s' % (tuple(real.shape), tuple(imag.shape)))
    real = real.float(); imag = imag.float()
    if real.size(1) > 1 and real.stride(1) != 1:
        real = real.contiguous()
    if imag.size(1) > 1 and imag.stride(1) != 1:
        imag = imag.contiguous()
    n, f = real.shape
    if out is None:
        out = torch.empty((n, 2, f), dtype=torch.float32, device=real.device)
    ld, im = _packed_strides(out, 'out')
    if n:
        check(_lib.lib.mh_pack(ptr(real), real.stride(0), ptr(imag), imag.stride(0), ptr(out), ld, im, n, f,
                               stream_ptr(real.device)))
    return out


def unpack_complex(p):
    """Row-planar [N, 2, F] -> two contiguous planar matrices (mh_unpack); p[:, 0], p[:, 1] are free views
    when strided results are acceptable."""
    require_cuda(p)
    ld, im = _packed_strides(p, 'p')
    n, _, f = p.shape
    re = torch.empty((n, f), dtype=torch.float32, device=p.device); imag = torch.empty_like(re)
    if n:
        check(_lib.lib.mh_unpack(ptr(p), ld, im, ptr(re), f, ptr(imag), f, n, f, stream_ptr(p.device)))
    return re, imag


def gather_rows(x, idx, out=None):
    """Rows idx of row-planar x into a dense [len(idx), 2, F] buffer (mh_gather_rows): halo send staging."""
    require_cuda(x, idx)
    ld, im = _packed_strides(x, 'x')
    idx = idx.to(torch.int32).contiguous()
    f = x.size(2)
    if out is None:
        out = torch.empty((idx.numel(), 2, f), dtype=torch.float32, device=x.device)
    check(_lib.lib.mh_gather_rows(ptr(x), ld, im, ptr(idx), idx.numel(), ptr(out), f, stream_ptr(x.device)))
    return out


def scatter_rows(buf, idx, x):
    """Row i of dense buf [len(idx), 2, F] into row idx[i] of row-planar x (mh_scatter_rows): halo receive."""
    require_cuda(x, idx, buf)
    ld, im = _packed_strides(x, 'x')
    idx = idx.to(torch.int32).contiguous()
    if not buf.is_contiguous() or buf.size(0) != idx.numel() or buf.size(2) != x.size(2):
        raise MagnetHipError('buf must be contiguous [len(idx), 2, F]')
    check(_lib.lib.mh_scatter_rows(ptr(buf), ptr(idx), idx.numel(), ptr(x), ld, im, x.size(2), stream_ptr(x.device)))
    return x


class MagneticLaplacian:
    """Complex CSR on the device: row_ptr int32 [n_rows+1], col_idx int32 [nnz],
    val float32 [nnz, 2] (re, im).  `n_cols` rows of the operand are addressed by col_idx."""

    def __init__(self, row_ptr, col_idx, val, n_cols, q=None, dinv=None):
        require_cuda(row_ptr, col_idx, val)
        assert row_ptr.dtype == torch.int32 and col_idx.dtype == torch.int32 and val.dtype == torch.float32
        self.row_ptr, self.col_idx, self.val = row_ptr.contiguous(), col_idx.contiguous(), val.contiguous()
        self.n_rows = row_ptr.numel() - 1
        self.n_cols = int(n_cols)
        self.nnz = col_idx.numel()
        self.q = q
        self.dinv = dinv
        self._adjoint = None
        self._plan = None

    # rows longer than 2*PLAN_CHUNK nonzeros are cut into chunks that get a wavefront each
    PLAN_CHUNK = 512

    def plan(self):
        """(plan tensor or None, n_long, n_chunks, chunk), built on first use (mh_cspmm_plan_*)."""
        if self._plan is None:
            lib = _lib.lib
            chunk = self.PLAN_CHUNK
            n_long, n_chunks = C.c_int32(0), C.c_int32(0)
            if self.n_rows > 0 and self.nnz > 2 * chunk:
                with torch.cuda.device(self.device):
                    tmp = torch.empty(2, dtype=torch.int32, device=self.device)
                    s = stream_ptr(self.device)
                    check(lib.mh_cspmm_plan_count(ptr(self.row_ptr), self.n_rows, chunk, ptr(tmp), C.byref(n_long),
                                                  C.byref(n_chunks), s))
                    buf = None
                    if n_long.value > 0:
                        buf = torch.empty(lib.mh_cspmm_plan_ints(n_long.value, n_chunks.value), dtype=torch.int32,
                                          device=self.device)
                        check(lib.mh_cspmm_plan_fill(ptr(self.row_ptr), self.n_rows, chunk, n_long.value,
                                                     n_chunks.value, ptr(buf), s))
                    self._plan = (buf, n_long.value, n_chunks.value, chunk)
            else:
                self._plan = (None, 0, 0, chunk)
        return self._plan

    @property
    def device(self):
        return self.row_ptr.device

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    # -- one propagation order ------------------------------------------------
    def spmm(self, X, out=None, alpha=1.0, z1=None, beta1=0.0, z2=None, beta2=0.0, bias=None, relu=False,
             mask=None):
        """out = alpha * (L X) + beta1 * z1 + beta2 * z2 [+ bias][complex ReLU]; all row-planar [*, 2, F]."""
        require_cuda(X, out, z1, z2, bias, mask)
        ldx, imx = _packed_strides(X, 'X')
        if X.size(0) != self.n_cols:
            raise MagnetHipError('X has %d rows, the Laplacian has %d columns' % (X.size(0), self.n_cols))
        f = X.size(2)
        if out is None:
            out = torch.empty((self.n_rows, 2, f), dtype=torch.float32, device=X.device)
        ldy, imy = _packed_strides(out, 'out')
        if out.size(0) != self.n_rows or out.size(2) != f:
            raise MagnetHipError('out has shape %s, expected (%d, 2, %d)' % (tuple(out.shape), self.n_rows, f))
        zargs = []
        for z, b, name in ((z1, beta1, 'z1'), (z2, beta2, 'z2')):
            if z is None:
                zargs += [None, 0, 0, 0.0]
            else:
                ldz, imz = _packed_strides(z, name)
                if z.size(0) != self.n_rows or z.size(2) != f:
                    raise MagnetHipError('%s has shape %s, expected (%d, 2, %d)' % (name, tuple(z.shape), self.n_rows, f))
                zargs += [ptr(z), ldz, imz, float(b)]
        if bias is not None:
            bias = bias.reshape(-1).float().contiguous()
            if bias.numel() != f:
                raise MagnetHipError('bias has %d entries, expected %d' % (bias.numel(), f))
        if mask is not None and (mask.dtype != torch.uint8 or mask.numel() != self.n_rows * f or not mask.is_contiguous()):
            raise MagnetHipError('mask must be contiguous uint8 [n_rows, F]')
        obs = SPMM_OBSERVER
        if obs is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(X.device))
        plan, n_long, n_chunks, chunk = self.plan()
        scratch = None
        if n_long > 0:
            scratch = torch.empty(n_chunks * 2 * f, dtype=torch.float32, device=X.device)
        check(_lib.lib.mh_cspmm_planned(ptr(self.row_ptr), ptr(self.col_idx), ptr(self.val), self.n_rows,
                                        ptr(X), ldx, imx, ptr(out), ldy, imy, float(alpha), *zargs,
                                        ptr(bias), int(bool(relu)), ptr(mask), f,
                                        ptr(plan), n_long, n_chunks, chunk, ptr(scratch),
                                        0 if scratch is None else scratch.numel() * 4, stream_ptr(X.device)))
        if obs is not None:
            ev1.record(torch.cuda.current_stream(X.device))
            obs(ev0, ev1, dict(nnz=self.nnz, n_rows=self.n_rows, F=f, n_addends=int(z1 is not None) + int(z2 is not None)))
        return out

    def _plan_args(self, f, device):
        plan, n_long, n_chunks, chunk = self.plan()
        scratch = torch.empty(n_chunks * 2 * f, dtype=torch.float32, device=device) if n_long > 0 else None
        return [ptr(plan), n_long, n_chunks, chunk, ptr(scratch), 0 if scratch is None else scratch.numel() * 4]

    def cheb_terms(self, x0, n_terms):
        """[T_0 x0, ..., T_{n_terms-1} x0] as one tensor [n_terms, N, 2, F] (mh_cheb_terms: the whole
        recursion in one library call)."""
        require_cuda(x0)
        _packed_strides(x0, 'x0')
        if self.n_rows != self.n_cols or x0.size(0) != self.n_cols:
            raise MagnetHipError('cheb_terms needs a square Laplacian and %d feature rows' % self.n_cols)
        n, _, f = x0.shape
        z = torch.empty((n_terms, n, 2, f), dtype=torch.float32, device=x0.device)
        z[0].copy_(x0)
        check(_lib.lib.mh_cheb_terms(ptr(self.row_ptr), ptr(self.col_idx), ptr(self.val), n, ptr(z), n * 2 * f, n_terms, f,
                                     *self._plan_args(f, x0.device), stream_ptr(x0.device)))
        return z

    def clenshaw(self, y, bias=None, relu=False):
        """sum_k T_k y[k] (+ bias, complex ReLU) for y [n_terms, N, 2, F] contiguous (mh_clenshaw).
        Returns (out [N, 2, F], mask or None)."""
        require_cuda(y, bias)
        if y.dim() != 4 or y.size(2) != 2 or not y.is_contiguous() or y.dtype != torch.float32:
            raise MagnetHipError('y must be contiguous float32 [n_terms, N, 2, F]')
        k1, n, _, f = y.shape
        if self.n_rows != self.n_cols or n != self.n_cols:
            raise MagnetHipError('clenshaw needs a square Laplacian and %d feature rows' % self.n_cols)
        out = torch.empty((n, 2, f), dtype=torch.float32, device=y.device)
        work = torch.empty((2, n, 2, f), dtype=torch.float32, device=y.device) if k1 > 2 else None
        mask = torch.empty((n, f), dtype=torch.uint8, device=y.device) if relu else None
        if bias is not None:
            bias = bias.reshape(-1).float().contiguous()
        check(_lib.lib.mh_clenshaw(ptr(self.row_ptr), ptr(self.col_idx), ptr(self.val), n, ptr(y), n * 2 * f, k1, ptr(work),
                                   ptr(out), ptr(bias), int(bool(relu)), ptr(mask), f, *self._plan_args(f, y.device),
                                   stream_ptr(y.device)))
        return out, mask

    def rows(self, r0, r1):
        """The operator restricted to output rows [r0, r1) (column ids unchanged; col_idx / val are views)."""
        a, b = int(self.row_ptr[r0]), int(self.row_ptr[r1])
        rp = (self.row_ptr[r0:r1 + 1] - self.row_ptr[r0]).to(torch.int32)
        return MagneticLaplacian(rp, self.col_idx[a:b], self.val[a:b], self.n_cols, q=self.q)

    # -- conversions ----------------------------------------------------------
    def to_scipy(self):
        import scipy.sparse as sp
        # float32 (re, im) pairs ARE complex64 in memory: no host-side assembly
        data = self.val.cpu().contiguous().numpy().view(np.complex64).reshape(-1) if self.nnz else np.zeros(0, np.complex64)
        m = sp.csr_matrix((data, self.col_idx.cpu().numpy(), self.row_ptr.cpu().numpy()),
                          shape=(self.n_rows, self.n_cols))
        m.has_sorted_indices = True               # rows are emitted in column order
        return m

    @staticmethod
    def from_scipy(mx, device='cuda'):
        mx = mx.tocsr()
        mx.sort_indices()
        data = np.asarray(mx.data).astype(np.complex64)
        val = torch.from_numpy(np.stack([data.real, data.imag], axis=1).astype(np.float32))
        return MagneticLaplacian(torch.from_numpy(mx.indptr.astype(np.int32)).to(device),
                                 torch.from_numpy(mx.indices.astype(np.int32)).to(device),
                                 val.to(device), mx.shape[1])

    @staticmethod
    def from_coo_pair(L_real, L_imag):
        """Complex CSR from the reference's two real COO tensors (src/sparse_Magnet.py:108-112)."""
        L_real = L_real.coalesce() if L_real.is_sparse else L_real.to_sparse().coalesce()
        L_imag = L_imag.coalesce() if L_imag.is_sparse else L_imag.to_sparse().coalesce()
        require_cuda(L_real.values(), L_imag.values())
        n_rows, n_cols = L_real.shape
        ir, ii = L_real.indices(), L_imag.indices()
        key = torch.cat([ir[0] * n_cols + ir[1], ii[0] * n_cols + ii[1]])
        uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
        val = torch.zeros((uniq.numel(), 2), dtype=torch.float32, device=key.device)
        nr = ir.size(1)
        val[:, 0].index_add_(0, inv[:nr], L_real.values().float())
        val[:, 1].index_add_(0, inv[nr:], L_imag.values().float())
        rows = torch.div(uniq, n_cols, rounding_mode='floor')
        cols = (uniq - rows * n_cols).to(torch.int32)
        counts = torch.bincount(rows, minlength=n_rows)
        row_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=key.device)
        row_ptr[1:] = torch.cumsum(counts, 0)
        return MagneticLaplacian(row_ptr.to(torch.int32), cols, val, n_cols)

    def adjoint(self):
        """Conjugate transpose as its own CSR (used by the backward pass of matrices that are not Hermitian)."""
        if self._adjoint is None:
            counts = (self.row_ptr[1:] - self.row_ptr[:-1]).long()
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), counts)
            key = self.col_idx.long() * self.n_rows + rows
            order = torch.argsort(key, stable=True)
            new_rows = self.col_idx.long()[order]
            val = self.val[order].clone()
            val[:, 1].neg_()
            rp = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=self.device)
            rp[1:] = torch.cumsum(torch.bincount(new_rows, minlength=self.n_cols), 0)
            self._adjoint = MagneticLaplacian(rp.to(torch.int32), rows[order].to(torch.int32), val, self.n_rows)
            self._adjoint._adjoint = self
        return self._adjoint


def build_magnetic_laplacian(row, col, size, q=0.25, max_eigen=2.0, gcn_appr=False, edge_weight=None, device=None):
    """GPU build of the normalised, rescaled magnetic Laplacian (hermitian_decomp_sparse with
    norm=True, laplacian=True; reference src/utils/hermitian.py:136-168).  `row`/`col` may be
    torch tensors (any device), numpy arrays or lists."""
    if device is None:
        device = row.device if isinstance(row, torch.Tensor) and row.is_cuda else torch.device('cuda')
    device = torch.device(device)
    if device.type != 'cuda':
        raise MagnetHipError('the Laplacian builder runs on the GPU only')

    def as_index(a):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
        t = t.reshape(-1)
        if t.dtype not in (torch.int32, torch.int64):
            t = t.long()
        return t.to(device).contiguous()

    row_t, col_t = as_index(row), as_index(col)
    if col_t.dtype != row_t.dtype:
        row_t, col_t = row_t.long(), col_t.long()
    if row_t.numel() != col_t.numel():
        raise MagnetHipError('row and col differ in length')
    w_t = None
    if edge_weight is not None:
        w_t = edge_weight if isinstance(edge_weight, torch.Tensor) else torch.as_tensor(np.asarray(edge_weight))
        w_t = w_t.reshape(-1).to(torch.float32).to(device).contiguous()
        if w_t.numel() != row_t.numel():
            raise MagnetHipError('edge_weight and row differ in length')
        if w_t.numel() == 0:
            w_t = None                      # no edges: an empty tensor has no data pointer to mark the build weighted
    n_edges, n = row_t.numel(), int(size)
    lib = _lib.lib
    with torch.cuda.device(device):
        ws_bytes = lib.mh_lap_workspace_bytes(n_edges, n, int(w_t is not None), int(bool(gcn_appr)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        row_ptr = torch.empty(n + 1, dtype=torch.int32, device=device)
        nnz = C.c_int64(0)
        s = stream_ptr(device)
        check(lib.mh_lap_build_count(ptr(row_t), ptr(col_t), int(row_t.dtype == torch.int64), ptr(w_t), n_edges, n,
                                     float(max_eigen), int(bool(gcn_appr)), ptr(ws), ws_bytes, ptr(row_ptr),
                                     C.byref(nnz), s))
        col_idx = torch.empty(nnz.value, dtype=torch.int32, device=device)
        val = torch.empty((nnz.value, 2), dtype=torch.float32, device=device)
        dinv = torch.empty(n, dtype=torch.float32, device=device)
        check(lib.mh_lap_build_fill(n_edges, n, float(q), float(max_eigen), int(w_t is not None), int(bool(gcn_appr)),
                                    ptr(ws), ws_bytes, nnz.value, ptr(col_idx), ptr(val), ptr(dinv), s))
        ws.record_stream(torch.cuda.current_stream(device))
    return MagneticLaplacian(row_ptr, col_idx, val, n, q=q, dinv=dinv)


def powerlaw_edges(n_nodes, n_edges, seed, first_edge=0, device='cuda'):
    """Synthetic power-law digraph on the device (mh_powerlaw_edges): int32 src, dst."""
    src = torch.empty(n_edges, dtype=torch.int32, device=device)
    dst = torch.empty(n_edges, dtype=torch.int32, device=device)
    check(_lib.lib.mh_powerlaw_edges(int(n_nodes), int(n_edges), int(seed), int(first_edge), ptr(src), ptr(dst),
                                     stream_ptr(torch.device(device))))
    return src, dst
