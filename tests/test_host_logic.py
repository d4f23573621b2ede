"""Host-side pieces that need no GPU: degree features, link metrics, the safe cache reader."""
import numpy as np
import torch

from conftest import load_golden
from oracle import magnet_oracle as O


def test_in_out_degree_on_cpu_matches_reference_output():
    from magnet_amd.utils.edge_data import in_out_degree
    g = load_golden('citeseer_link_case.npz')
    edges = torch.from_numpy(np.vstack([g['row'], g['col']]))
    assert np.array_equal(in_out_degree(edges, int(g['n'])).numpy(), g['X'])
    assert np.array_equal(in_out_degree(torch.from_numpy(g['iod_small_in']), 5).numpy(), g['iod_small_out'])
    assert np.array_equal(in_out_degree(edges, int(g['n'])).numpy(), O.in_out_degree(edges, int(g['n'])).numpy())


def test_link_prediction_evaluation():
    from magnet_amd.utils.edge_data import link_prediction_evaluation, acc
    logp = torch.log(torch.tensor([[0.7, 0.2, 0.1], [0.2, 0.7, 0.1], [0.6, 0.3, 0.1], [0.1, 0.2, 0.7], [0.3, 0.6, 0.1]]))
    y = torch.tensor([0, 1, 1, 2, 1])
    (v, t) = link_prediction_evaluation(logp, logp, y, y)
    assert v == t
    acc_full, acc2, auc, f1_micro, f1_macro = v
    assert abs(acc_full - 0.8) < 1e-12 and abs(acc2 - 0.75) < 1e-12 and abs(f1_micro - 0.75) < 1e-12
    # AUC with the reference's score out[:,0]/(out[:,0]+out[:,1]) and labels as they are:
    # positives (label 1) have scores .222, .667, .333 and the negative .778 -> no positive outranks it
    assert abs(auc - 0.0) < 1e-12
    assert abs(f1_macro - (2 / 3 + 0.8) / 2) < 1e-12
    assert acc(torch.tensor([1, 0, 1]), torch.tensor([1, 1, 1])) == 2 / 3


def test_cache_reader_round_trip(tmp_path):
    """A cache in the reference's format ({'L': [scipy matrices]}, HIGHEST_PROTOCOL) is read back by the
    opcode walker exactly, and a pickle that tries to call something is reduced to inert data."""
    import pickle
    import scipy.sparse as sp
    from magnet_amd.utils.inert_pickle import read_sparse_list, load_inert, Inert
    src, dst = O.powerlaw_edges(60, 500, 3)
    L = O.hermitian_decomp_sparse(src, dst, 60, 0.25)
    T = O.cheb_poly_sparse(L, 2)
    p = tmp_path / 'data0.25_2_sparse.pk'
    with open(p, 'wb') as f:
        pickle.dump({'L': T}, f, protocol=pickle.HIGHEST_PROTOCOL)
    back = read_sparse_list(str(p))
    assert len(back) == 3 and back[1]['__class__'].endswith('csr_matrix')
    assert np.array_equal(back[1]['indptr'], L.indptr) and np.array_equal(back[1]['indices'], L.indices)
    assert np.array_equal(back[1]['data'], L.data)
    t2 = T[2].tocsr()
    assert np.array_equal(back[2]['data'], t2.data)

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo should-never-run > %s' % (tmp_path / 'pwned'),))
    node = load_inert(pickle.dumps({'L': [Evil()]}))
    assert isinstance(node['L'][0], Inert) and not (tmp_path / 'pwned').exists()
