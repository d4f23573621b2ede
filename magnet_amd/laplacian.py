"""Device-resident magnetic Laplacian (complex CSR) and the calls that build it and apply it:
torch.ops.magnet.* of the thin C++ extension, which forwards to the C ABI of libmagnet_hip.so.

`build_magnetic_laplacian` is the GPU counterpart of hermitian_decomp_sparse
(reference src/utils/hermitian.py:136-168); `MagneticLaplacian.spmm` is one order of the
propagation (the four torch.spmm of reference src/layer/sparse_magnet.py:9-17 in one pass).
"""
import numpy as np
import torch

from ._lib import ops, require_cuda, MagnetHipError


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
    if n:
        ops.pack(real, imag, out)
    return out


def unpack_complex(p):
    """Row-planar [N, 2, F] -> two contiguous planar matrices (mh_unpack); p[:, 0], p[:, 1] are free views
    when strided results are acceptable."""
    require_cuda(p)
    _packed_strides(p, 'p')
    n, _, f = p.shape
    re = torch.empty((n, f), dtype=torch.float32, device=p.device); imag = torch.empty_like(re)
    if n:
        ops.unpack(p, re, imag)
    return re, imag


def gather_rows(x, idx, out=None):
    """Rows idx of row-planar x into a dense [len(idx), 2, F] buffer (mh_gather_rows): halo send staging."""
    require_cuda(x, idx)
    idx = idx.to(torch.int32).contiguous()
    f = x.size(2)
    if out is None:
        out = torch.empty((idx.numel(), 2, f), dtype=torch.float32, device=x.device)
    ops.gather_rows(x, idx, out)
    return out


def scatter_rows(buf, idx, x):
    """Row i of dense buf [len(idx), 2, F] into row idx[i] of row-planar x (mh_scatter_rows): halo receive."""
    require_cuda(x, idx, buf)
    idx = idx.to(torch.int32).contiguous()
    if not buf.is_contiguous() or buf.size(0) != idx.numel() or buf.size(2) != x.size(2):
        raise MagnetHipError('buf must be contiguous [len(idx), 2, F]')
    ops.scatter_rows(buf, idx, x)
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
        self._scratch = {}

    # rows longer than 2*PLAN_CHUNK nonzeros are cut into chunks that get a wavefront each
    PLAN_CHUNK = 512

    def plan(self):
        """(plan tensor or None, n_long, n_chunks, chunk), built on first use (mh_cspmm_plan_*)."""
        if self._plan is None:
            chunk = self.PLAN_CHUNK
            self._plan = (None, 0, 0, chunk)
            if self.n_rows > 0 and self.nnz > 2 * chunk:
                buf, n_long, n_chunks = ops.plan_build(self.row_ptr, chunk)
                if n_long > 0:
                    self._plan = (buf, n_long, n_chunks, chunk)
        return self._plan

    def _scratch_for(self, f):
        """Hub-row partial-sum buffer [n_chunks, 2, F] of this matrix, kept between calls.  One buffer per
        stream the matrix has been applied on: calls ordered on a stream may share it, calls on different
        streams may overlap and get their own."""
        n_chunks = self.plan()[2]
        if n_chunks == 0:
            return None
        key = torch.cuda.current_stream(self.device).cuda_stream
        buf = self._scratch.get(key)
        need = n_chunks * 2 * f
        if buf is None or buf.numel() < need:
            buf = self._scratch[key] = torch.empty(need, dtype=torch.float32, device=self.device)
        return buf

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
        _packed_strides(X, 'X')
        f = X.size(2)
        if out is None:
            out = torch.empty((self.n_rows, 2, f), dtype=torch.float32, device=X.device)
        if bias is not None:
            bias = bias.reshape(-1).float().contiguous()
        obs = SPMM_OBSERVER
        if obs is not None:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(X.device))
        plan, n_long, n_chunks, chunk = self.plan()
        # shapes, dtypes and strides are checked by the extension, which reads (ld, im) off the tensors
        ops.cspmm(self.row_ptr, self.col_idx, self.val, self.n_cols, X, out, float(alpha), z1, float(beta1), z2, float(beta2),
                  bias, bool(relu), mask, plan, n_long, n_chunks, chunk, self._scratch_for(f) if n_long > 0 else None)
        if obs is not None:
            ev1.record(torch.cuda.current_stream(X.device))
            obs(ev0, ev1, dict(nnz=self.nnz, n_rows=self.n_rows, F=f, n_addends=int(z1 is not None) + int(z2 is not None)))
        return out

    def _plan_args(self, f):
        plan, n_long, n_chunks, chunk = self.plan()
        return [plan, n_long, n_chunks, chunk, self._scratch_for(f) if n_long > 0 else None]

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
        ops.cheb_terms_(self.row_ptr, self.col_idx, self.val, z, *self._plan_args(f))
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
        ops.clenshaw(self.row_ptr, self.col_idx, self.val, y, work, out, bias, bool(relu), mask, *self._plan_args(f))
        return out, mask

    def rows(self, r0, r1):
        """The operator restricted to output rows [r0, r1) (column ids unchanged; col_idx / val are views)."""
        a, b = int(self.row_ptr[r0]), int(self.row_ptr[r1])
        rp = (self.row_ptr[r0:r1 + 1] - self.row_ptr[r0]).to(torch.int32)
        return MagneticLaplacian(rp, self.col_idx[a:b], self.val[a:b], self.n_cols, q=self.q)

    # -- conversions ----------------------------------------------------------
    def to_scipy(self):
        """The reference's result type: scipy CSR, complex64 values, int32 indices.  The three arrays come back
        through page-locked host buffers (a pageable destination costs ten times the copy in first-touch page
        faults: 230 ms against 20 ms for 80M nonzeros); the scipy matrix keeps those buffers alive."""
        import scipy.sparse as sp

        def host(t):
            if t.numel() < (1 << 16):
                return t.cpu()
            try:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            except RuntimeError:                       # page-locked memory exhausted: ordinary host memory
                return t.cpu()
            return h.copy_(t, non_blocking=True)
        val, col, rp = host(self.val), host(self.col_idx), host(self.row_ptr)
        torch.cuda.current_stream(self.device).synchronize()
        # float32 (re, im) pairs ARE complex64 in memory: no host-side assembly
        data = val.numpy().view(np.complex64).reshape(-1) if self.nnz else np.zeros(0, np.complex64)
        m = sp.csr_matrix((data, col.numpy(), rp.numpy()), shape=(self.n_rows, self.n_cols))
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
    with torch.cuda.device(device):
        row_ptr, col_idx, val, dinv = ops.lap_build(row_t, col_t, w_t, int(size), float(q), float(max_eigen), bool(gcn_appr))
    return MagneticLaplacian(row_ptr, col_idx, val, int(size), q=q, dinv=dinv)


class LaplacianBlockBuild:
    """Two-phase build of ONE rank's row block [r0, r1) of the magnetic Laplacian (mh_lap_block_*):

        b = LaplacianBlockBuild(fwd, bwd, n, r0, r1)      # fwd / bwd: (row, col[, weight]) of the edges whose
                                                          # source / target lies in the block, in input order
        b.dinv_block                                      # d^-1/2 of the block's rows -> gather over the ranks
        lap = b.finish(dinv_full, q)                      # CSR of the block, GLOBAL column ids

    The blocks of a partition, stacked, are bit for bit the CSR build_magnetic_laplacian returns.
    `node_records` (optional, unweighted builds): int32 [n], out-degree + in-degree of every node over ALL edges of the
    graph (duplicates counted, a self-loop twice).  With it a column's count travels through the sort in the keys'
    spare bits and the fill computes d^-1/2 of the columns from it instead of gathering it from dinv_full
    (mh_lap_block_count_tagged); the counts must be exact."""

    def __init__(self, fwd, bwd, size, r0, r1, max_eigen=2.0, node_records=None):
        def norm(lst):
            row, col = lst[0], lst[1]
            w = lst[2] if len(lst) > 2 else None
            require_cuda(row, col, w)
            row, col = row.reshape(-1).contiguous(), col.reshape(-1).contiguous()
            if w is not None:
                w = w.reshape(-1).to(torch.float32).contiguous()
            return row, col, w
        fr, fc, fw = norm(fwd)
        br, bc, bw = norm(bwd)
        dt = torch.int64 if torch.int64 in (fr.dtype, fc.dtype, br.dtype, bc.dtype) else torch.int32
        fr, fc, br, bc = (t.to(dt) for t in (fr, fc, br, bc))
        if (fw is None) != (bw is None):          # one list weighted, the other not: unit weights for the other
            fw = torch.ones(fr.numel(), device=fr.device) if fw is None else fw
            bw = torch.ones(br.numel(), device=br.device) if bw is None else bw
        self.n, self.r0, self.r1, self.max_eigen = int(size), int(r0), int(r1), float(max_eigen)
        self.n_fwd, self.n_bwd, self.weighted = fr.numel(), br.numel(), fw is not None
        with torch.cuda.device(fr.device):
            if node_records is not None:
                node_records = None if self.weighted else node_records.to(device=fr.device, dtype=torch.int32).contiguous()
            self._ws, self.row_ptr, self.dinv_block, self.nnz = ops.lap_block_count(
                fr, fc, fw, br, bc, bw, self.n, self.r0, self.r1, self.max_eigen, node_records)

    def finish(self, dinv_full, q=0.25):
        with torch.cuda.device(dinv_full.device):
            col_idx, val = ops.lap_block_fill(self._ws, self.n_fwd, self.n_bwd, self.n, self.r0, self.r1, float(q),
                                              self.max_eigen, self.weighted, self.nnz, dinv_full.contiguous())
        self._ws = None
        return MagneticLaplacian(self.row_ptr, col_idx, val, self.n, q=q, dinv=dinv_full)


def halo_plan(col_idx, n_cols, c0, c1, bounds):
    """(halo_ids, peer_ptr, col_local) of a row block whose CSR holds global column ids (mh_halo_plan_*):
    the remote feature rows it gathers, grouped by owning rank, and its columns renumbered to a compact
    [own rows | halo rows] buffer.  `bounds`: int32 device tensor [world + 1] of the ranks' row boundaries."""
    require_cuda(col_idx, bounds)
    with torch.cuda.device(col_idx.device):
        return ops.halo_plan(col_idx.contiguous(), int(n_cols), int(c0), int(c1), bounds.to(torch.int32).contiguous())


def powerlaw_edges(n_nodes, n_edges, seed, first_edge=0, device='cuda'):
    """Synthetic power-law digraph on the device (mh_powerlaw_edges): int32 src, dst."""
    src = torch.empty(n_edges, dtype=torch.int32, device=device)
    dst = torch.empty(n_edges, dtype=torch.int32, device=device)
    if n_edges:
        ops.powerlaw_edges(int(n_nodes), int(seed), int(first_edge), src, dst)
    elif n_nodes < 1:
        raise MagnetHipError('n_nodes must be positive')
    return src, dst
