"""Drop-in for the sparse half of the reference's utils/hermitian.py, built on the GPU.

    hermitian_decomp_sparse(row, col, size, q, norm, laplacian, max_eigen, gcn_appr, edge_weight)
        same signature and result type (scipy CSR complex64, int32 indices) as
        reference src/utils/hermitian.py:136-168; computed by libmagnet_hip.so.
    hermitian_decomp_sparse_device(...)
        same arguments, returns the device-resident MagneticLaplacian (no host copy).
    cheb_poly_sparse(A, K)
        reference src/utils/hermitian.py:117-134 returns the K+1 matrices T_k(A); here the
        result is a ChebBasis: a list-like of length K+1 that carries A on the device for the
        feature-side recursion and only materialises a sparse T_k (host SpGEMM) if indexed.
"""
import numpy as np
import scipy.sparse as sp

from ..laplacian import MagneticLaplacian, build_magnetic_laplacian
from .._lib import MagnetHipError


def _check_flags(norm, laplacian):
    if not laplacian:
        # the reference never assigns L on this branch (hermitian.py:156-167) and raises too
        raise ValueError('laplacian=False is not defined by the reference implementation')
    if not norm:
        raise ValueError('norm=False is not supported: every caller of the sparse path passes norm=True '
                         '(src/utils/preprocess.py:177-181,230)')


def hermitian_decomp_sparse_device(row, col, size, q=0.25, norm=True, laplacian=True, max_eigen=2,
                                   gcn_appr=False, edge_weight=None, device=None):
    _check_flags(norm, laplacian)
    return build_magnetic_laplacian(row, col, size, q=q, max_eigen=max_eigen, gcn_appr=gcn_appr,
                                    edge_weight=edge_weight, device=device)


def hermitian_decomp_sparse(row, col, size, q=0.25, norm=True, laplacian=True, max_eigen=2,
                            gcn_appr=False, edge_weight=None):
    return hermitian_decomp_sparse_device(row, col, size, q, norm, laplacian, max_eigen, gcn_appr,
                                          edge_weight).to_scipy()


class ChebBasis:
    """[T_0(L), ..., T_K(L)] without forming the polynomials: len() == K+1, `.laplacian` is the
    device CSR the layers recurse with.  Indexing returns the scipy matrix the reference would
    hold at that position (computed on the host on first use; T_k fill-in grows like deg^k)."""

    def __init__(self, laplacian, K):
        self.laplacian = laplacian
        self.K = int(K)
        self._host = None

    def __len__(self):
        return self.K + 1

    def _materialise(self):
        if self._host is None:
            A = self.laplacian.to_scipy()
            n = A.shape[0]
            out = [sp.coo_matrix((np.ones(n), (np.arange(n), np.arange(n))), shape=(n, n), dtype=np.float32)]
            if self.K >= 1:
                out.append(A)
            for k in range(2, self.K + 1):
                out.append(2.0 * A.dot(out[k - 1]) - out[k - 2])
            self._host = out
        return self._host

    def __getitem__(self, k):
        return self._materialise()[k]

    def __iter__(self):
        return iter(self._materialise())


def cheb_poly_sparse(A, K, device='cuda'):
    if isinstance(A, ChebBasis):
        A = A.laplacian
    if not isinstance(A, MagneticLaplacian):
        if not sp.issparse(A):
            raise MagnetHipError('cheb_poly_sparse expects a MagneticLaplacian or a scipy sparse matrix')
        A = MagneticLaplacian.from_scipy(A, device=device)
    return ChebBasis(A, K)
