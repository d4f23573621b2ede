"""The CPU oracle against the golden vectors (outputs of the reference's own
functions and the Laplacian caches the reference ships).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_golden, assert_close
from oracle import magnet_oracle as O

LAP = load_golden('lap_cases.npz')
LAP_NAMES = [str(s) for s in LAP['names']]


def _lap_inputs(g, k):
    n, q, lam, gcn = g[k + '/params']
    w = g[k + '/w'] if (k + '/w') in g.files else None
    return g[k + '/row'], g[k + '/col'], int(n), float(q), float(lam), bool(gcn), w


@pytest.mark.parametrize('name', LAP_NAMES)
def test_builder_matches_reference_outputs(name):
    row, col, n, q, lam, gcn, w = _lap_inputs(LAP, name)
    L = O.hermitian_decomp_sparse(row, col, n, q, True, True, lam, gcn, w)
    assert L.dtype == np.complex64 and L.indices.dtype == np.int32 and L.indptr.dtype == np.int32
    assert np.array_equal(L.indptr, LAP[name + '/indptr'])
    assert np.array_equal(L.indices, LAP[name + '/indices'])
    assert_close(L.data, LAP[name + '/data'], rel=1e-6, what=name)


def _telegram():
    return _lap_inputs(LAP, 'telegram_weighted_q05')


@pytest.mark.parametrize('tag', [str(s) for s in load_golden('telegram_cache.npz')['names']])
def test_builder_matches_reference_shipped_caches(tag):
    """dataset/data/tmp/telegram/telegram/data{q}_{K}_sparse.pk: the K=1 caches and data0_2
    were written from the unweighted graph, data0.05_2 / data0.15_2 from the weighted one."""
    g = load_golden('telegram_cache.npz')
    q, K = tag.split('_')
    row, col, n, _, _, _, w = _telegram()
    weighted = tag in ('0.05_2', '0.15_2')
    L = O.hermitian_decomp_sparse(row, col, n, float(q), True, True, 2.0, False, w if weighted else None)
    assert np.array_equal(L.indptr, g[tag + '/T1/indptr'])
    assert np.array_equal(L.indices, g[tag + '/T1/indices'])
    assert_close(L.data, g[tag + '/T1/data'], rel=1e-6, what=tag)
    if K == '2':
        T = O.cheb_poly_sparse(L, 2)[2].tocsr(); T.sort_indices()
        # the cached SpGEMM output keeps its columns unsorted inside a row
        G = sp.csr_matrix((g[tag + '/T2/data'], g[tag + '/T2/indices'], g[tag + '/T2/indptr']), shape=(n, n))
        G.sort_indices()
        assert np.array_equal(T.indptr, G.indptr)
        assert np.array_equal(T.indices, G.indices)
        assert_close(T.data, G.data, rel=1e-5, what=tag + ' T2')


def test_cheb_poly_matches_reference():
    g = load_golden('cheb_poly_cases.npz')
    row, col, n, _, _, _, w = _telegram()
    for name, q, ww in [('unw_q25', 0.25, None), ('w_q05', 0.05, w)]:
        T = O.cheb_poly_sparse(O.hermitian_decomp_sparse(row, col, n, q, edge_weight=ww), 3)
        assert len(T) == 4 and T[0].dtype == np.float32 and T[0].nnz == n
        for k in (2, 3):
            t = T[k].tocsr(); t.sort_indices()
            assert np.array_equal(t.indptr, g['%s/T%d/indptr' % (name, k)])
            assert np.array_equal(t.indices, g['%s/T%d/indices' % (name, k)])
            assert_close(t.data, g['%s/T%d/data' % (name, k)], rel=1e-5, what='%s T%d' % (name, k))


def _telegram_lists(q, K, weighted):
    row, col, n, _, _, _, w = _telegram()
    L = O.hermitian_decomp_sparse(row, col, n, q, edge_weight=w if weighted else None)
    return L, O.laplacian_lists(O.cheb_poly_sparse(L, K))


CONV = load_golden('conv_cases.npz')


@pytest.mark.parametrize('name', [str(s) for s in CONV['names']])
def test_conv_forward_backward_matches_reference(name):
    K, cin, cout, q, weighted = CONV[name + '/params']
    L, (Lr, Li) = _telegram_lists(float(q), int(K), bool(weighted))
    t = lambda k: torch.from_numpy(CONV[name + '/' + k].copy())
    W = t('weight').requires_grad_(True); b = t('bias').requires_grad_(True)
    Xr = t('Xr').requires_grad_(True); Xi = t('Xi').requires_grad_(True)
    real, imag = O.cheb_conv(Lr, Li, W, b, Xr, Xi)
    ((real * t('Gr')).sum() + (imag * t('Gi')).sum()).backward()
    assert_close(real.detach(), CONV[name + '/real'], what='real')
    assert_close(imag.detach(), CONV[name + '/imag'], what='imag')
    assert_close(W.grad, CONV[name + '/dweight'], what='dW')
    assert_close(b.grad, CONV[name + '/dbias'], what='db')
    assert_close(Xr.grad, CONV[name + '/dXr'], what='dXr')
    assert_close(Xi.grad, CONV[name + '/dXi'], what='dXi')
    # the feature-side recursion is the same operator
    r2, i2 = O.cheb_conv_recursion(L, W.detach(), b.detach(), Xr.detach(), Xi.detach())
    assert_close(r2, CONV[name + '/real'], what='recursion real')
    assert_close(i2, CONV[name + '/imag'], what='recursion imag')


NET = load_golden('net_cases.npz')


@pytest.mark.parametrize('name', [str(s) for s in NET['names']])
def test_nets_match_reference(name):
    K, layer, act, F, q, cin = NET[name + '/params']
    _, (Lr, Li) = _telegram_lists(float(q), int(K), False)
    pre = name + '/state/'
    state = {k[len(pre):]: torch.from_numpy(NET[k].copy()) for k in NET.files if k.startswith(pre)}
    X = torch.from_numpy(NET[name + '/X'])
    if name.startswith('node'):
        out = O.chebnet_forward(state, Lr, Li, X, X, int(layer), bool(act))
        assert_close(out, NET[name + '/preds'], what='preds')
    else:
        out = O.chebnet_edge_forward(state, Lr, Li, X, X, torch.from_numpy(NET[name + '/pairs']), int(layer), bool(act))
        assert_close(out, NET[name + '/out'], what='out')


CFG = load_golden('config_cases.npz')


@pytest.mark.parametrize('name', [str(s) for s in CFG['names']])
def test_benchmark_configurations_match_reference(name):
    """The reference's ChebNet at the two small node-classification configurations (Cora citation
    graph shipped with the reference; 2 500-node cyclic DSBM with its precomputed sparse T_2):
    predictions of the reference against the oracle in both formulations."""
    K, layer, act, F, q, cin, n = CFG[name + '/params']
    K, layer, n = int(K), int(layer), int(n)
    L = O.hermitian_decomp_sparse(CFG[name + '/row'].astype(np.int64), CFG[name + '/col'].astype(np.int64), n, float(q))
    Ts = O.cheb_poly_sparse(L, K)
    assert [t.nnz for t in Ts] == CFG[name + '/nnz'].tolist()
    Lr, Li = O.laplacian_lists(Ts)
    pre = name + '/state/'
    state = {k[len(pre):]: torch.from_numpy(CFG[k].copy()) for k in CFG.files if k.startswith(pre)}
    X = torch.from_numpy(CFG[name + '/X'])
    out = O.chebnet_forward(state, Lr, Li, X, X, layer, bool(act))
    assert_close(out, CFG[name + '/preds'], what='preds')
    # feature-side recursion (what the HIP layers run) against the reference's precomputed-T_k numbers
    r, i = X, X
    for j in range(layer):
        w, b = state['Chebs.%d.weight' % (2 * j)], state['Chebs.%d.bias' % (2 * j)]
        r, i = O.cheb_conv_recursion(L, w, b, r, i)
        r, i = O.complex_relu(r, i)
    x = torch.cat((r, i), dim=-1)
    logits = x @ state['Conv.weight'].squeeze(-1).t() + state['Conv.bias']
    preds = torch.log_softmax(logits.t().unsqueeze(0), dim=1)
    assert_close(preds, CFG[name + '/preds'], rel=2e-5, what='preds by recursion')


def test_in_out_degree_quirk():
    g = load_golden('citeseer_link_case.npz')
    got = O.in_out_degree(torch.from_numpy(g['iod_small_in']), 5)
    assert np.array_equal(got.numpy(), g['iod_small_out'])
    edges = torch.from_numpy(np.vstack([g['row'], g['col']]))
    assert np.array_equal(O.in_out_degree(edges, int(g['n'])).numpy(), g['X'])


def test_citeseer_link_case():
    g = load_golden('citeseer_link_case.npz')
    n = int(g['n'])
    L = O.hermitian_decomp_sparse(g['row'], g['col'], n, 0.1)
    assert np.array_equal(L.indptr, g['indptr']) and np.array_equal(L.indices, g['indices'])
    assert_close(L.data, g['data'], rel=1e-6)
    Lr, Li = O.laplacian_lists(O.cheb_poly_sparse(L, 2))
    state = {k[6:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith('state/')}
    X = torch.from_numpy(g['X'])
    out = O.chebnet_edge_forward(state, Lr, Li, X, X, torch.from_numpy(g['pairs']), 2, True)
    assert_close(out, g['out'], what='out')


@pytest.mark.parametrize('name', ['citeseer_task1', 'cora_ml_task2', 'cornell_task1', 'cornell_task2', 'texas_task2'])
def test_reference_link_splits_step(name):
    """Configuration 3 on split 0 of the link-split caches the reference ships (citeseer existence, cora_ml
    direction, K = 2; the WebKB Cornell / Texas splits with the sweep's K = 1): the oracle against the reference's Laplacian, degree features, training-pair outputs, loss,
    validation outputs and every parameter gradient."""
    g = load_golden('link_split_cases.npz')
    pre = name + '/'
    size = int(g[pre + 'size'])
    K, q = int(g[pre + 'params'][0]), float(g[pre + 'params'][1])
    edges = torch.from_numpy(g[pre + 'graph'].astype(np.int64))
    L = O.hermitian_decomp_sparse(edges[0].numpy(), edges[1].numpy(), size, q)
    assert np.array_equal(L.indptr, g[pre + 'indptr']) and np.array_equal(L.indices, g[pre + 'indices'])
    assert_close(L.data, g[pre + 'data'], rel=1e-6)
    X = O.in_out_degree(edges, size)
    assert np.array_equal(X.numpy(), g[pre + 'X'])
    Lr, Li = O.laplacian_lists(O.cheb_poly_sparse(L, K))
    state = {k[len(pre) + 6:]: torch.from_numpy(g[k].copy()).requires_grad_(True) for k in g.files if k.startswith(pre + 'state/')}
    tp = torch.from_numpy(g[pre + 'train/pairs'].astype(np.int64)); tl = torch.from_numpy(g[pre + 'train/label'].astype(np.int64))
    out = O.chebnet_edge_forward(state, Lr, Li, X, X, tp, 2, True)
    loss = torch.nn.functional.nll_loss(out, tl)
    loss.backward()
    assert_close(out.detach(), g[pre + 'out'], what='out')
    assert abs(loss.item() - float(g[pre + 'loss'])) <= 1e-5 * float(g[pre + 'loss'])
    for k, v in state.items():
        if v.grad is not None:
            assert_close(v.grad, g[pre + 'grad/' + k], what='grad ' + k)
    with torch.no_grad():
        ov = O.chebnet_edge_forward(state, Lr, Li, X, X, torch.from_numpy(g[pre + 'validate/pairs'].astype(np.int64)), 2, True)
    assert_close(ov, g[pre + 'val_out'], what='validation out')
    counts = {'citeseer_task1': (7345, 3763), 'cora_ml_task2': (6321, 6589), 'cornell_task1': (210, 227),
              'cornell_task2': (209, 227), 'texas_task2': (204, 239)}
    assert (len(tp), edges.size(1)) == counts[name]


def test_structure_rules():
    """Self-loops and equal-weight reciprocal edges vanish from the support but count in degrees."""
    row = np.array([0, 1, 0, 2, 3, 3]); col = np.array([1, 0, 2, 2, 0, 0])
    L = O.hermitian_decomp_sparse(row, col, 5, 0.25).toarray()
    assert L[0, 1] == 0 and L[1, 0] == 0 and L[2, 2] == 0          # reciprocal, self-loop
    assert L[0, 2] != 0 and L[2, 0] == np.conj(L[0, 2])
    # node 0: s = 1 (0<->1) + .5 (0->2) + 1 (3=>0 twice, /2) = 2.5 ; node 2: .5 + 1 (self-loop) = 1.5
    assert np.isclose(abs(L[0, 2]), 0.5 / np.sqrt(2.5 * 1.5), rtol=1e-6)
    assert np.isclose(np.angle(-L[0, 2]), 2 * np.pi * 0.25 * 1.0, rtol=1e-6)
    assert np.isclose(np.angle(-L[3, 0]), 2 * np.pi * 0.25 * 2.0 - 2 * np.pi, rtol=1e-5) or \
        np.isclose(abs(np.angle(-L[3, 0])), np.pi, rtol=1e-5)


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    # counter = (0,0,0,0), key = (0,0)
    r = O.philox4x32_10(np.array([0], np.uint64), 0)
    assert [int(x[0]) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
