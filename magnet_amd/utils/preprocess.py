"""Glue between dataset objects and the Laplacian (reference src/utils/preprocess.py:154-192,223-233).

Same names, arguments and return values as the reference; the Laplacian list that comes back is
a ChebBasis (device-resident L~ + K) instead of K+1 host matrices.  With save_pk=True the cache
file the reference's driver looks for (`data{q}_{K}_sparse.pk`, src/sparse_Magnet.py:91) is written
in the reference's format ({'L': [scipy matrices]}, pickle HIGHEST_PROTOCOL); forming the T_k for
it is host SpGEMM work exactly as in the reference, so pass save_pk=False at scale.
"""
import os
import pickle as pk

import numpy as np
import torch

from .hermitian import hermitian_decomp_sparse_device, cheb_poly_sparse, ChebBasis
from .inert_pickle import read_sparse_list
from ..laplacian import MagneticLaplacian


def _np(a, dtype):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a).astype(dtype)


def geometric_dataset_sparse(q, K, root='../dataset/data/tmp/', subset='Cornell', dataset=None,
                             load_only=False, save_pk=True, laplacian=True, gcn_appr=False):
    """dataset: callable(root=..., name=...) -> sequence whose item 0 has edge_index [2,E], x, y,
    train/val/test masks and optionally edge_weight (reference :154-192)."""
    ds = dataset(root=root) if subset == '' else dataset(root=root, name=subset)
    d = ds[0]
    size = d.y.size(-1) if isinstance(d.y, torch.Tensor) else np.asarray(d.y).shape[-1]
    f_node, e_node = d.edge_index[0], d.edge_index[1]
    label = _np(d.y, 'int')
    X = _np(d.x, 'float32')
    train_mask, val_mask, test_mask = (_np(m, 'bool_') for m in (d.train_mask, d.val_mask, d.test_mask))
    if load_only:
        return X, label, train_mask, val_mask, test_mask
    L = hermitian_decomp_sparse_device(f_node, e_node, size, q, norm=True, laplacian=laplacian, max_eigen=2.0,
                                       gcn_appr=gcn_appr, edge_weight=getattr(d, 'edge_weight', None))
    multi_order_laplacian = cheb_poly_sparse(L, K)
    save_name = root + '/data' + str(q) + '_' + str(K)
    if laplacian is False:
        save_name += '_P'
    if save_pk:
        with open(save_name + '_sparse.pk', 'wb') as f:
            pk.dump({'L': list(multi_order_laplacian)}, f, protocol=pk.HIGHEST_PROTOCOL)
    return X, label, train_mask, val_mask, test_mask, multi_order_laplacian


def to_edge_dataset_sparse(q, edge_index, K, data_split, size, root='../dataset/data/tmp/', laplacian=True,
                           norm=True, max_eigen=2.0, gcn_appr=False):
    """Per-split Laplacian of the observed graph (reference :223-233; like the reference it
    ignores `norm` / `max_eigen` and always builds the normalised Laplacian with max_eigen 2)."""
    save_name = root + '/edge_' + str(q) + '_' + str(K) + '_' + str(data_split) + '.pk'
    if os.path.isfile(save_name):
        return load_laplacian_cache(save_name, K)
    L = hermitian_decomp_sparse_device(edge_index[0], edge_index[1], size, q, norm=True, laplacian=laplacian,
                                       max_eigen=2.0, gcn_appr=gcn_appr)
    return cheb_poly_sparse(L, K)


def load_laplacian_cache(path, K=None, device='cuda'):
    """Read a `{'L': [T_0, T_1, ...]}` cache written by the reference (src/utils/preprocess.py:185-191)
    or by geometric_dataset_sparse above, WITHOUT unpickling it (opcode walk, nothing in the file is
    executed) and return the ChebBasis of its T_1."""
    import scipy.sparse as sp
    L = read_sparse_list(path)
    if len(L) == 0:
        raise ValueError('%s holds an empty Laplacian list' % path)
    order = len(L) - 1 if K is None else K
    if len(L) == 1:
        # a K = 0 cache holds T_0 = I alone: the Laplacian cannot be read back from it, and a network of that
        # order never applies it.  An all-zero operator of the right size stands in (T_0 X = X needs no operator).
        if order > 0:
            raise ValueError('%s was written for K = 0 and holds T_0 only; order %d needs T_1' % (path, order))
        n = L[0]['_shape'][0]
        m = sp.csr_matrix((n, n), dtype=np.complex64)
    else:
        t1 = L[1]
        n = t1['_shape'][0]
        m = sp.csr_matrix((t1['data'], t1['indices'], t1['indptr']), shape=(n, n))
    return ChebBasis(MagneticLaplacian.from_scipy(m, device=device), order)
