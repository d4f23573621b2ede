"""Host-side pieces that need no GPU: degree features, link metrics, the safe cache reader."""
import numpy as np
import pytest
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
    # against the reference's own function on random three-class outputs
    g = load_golden('label_cases.npz')
    got = link_prediction_evaluation(torch.from_numpy(g['metrics/logp_val']), torch.from_numpy(g['metrics/logp_test']),
                                     torch.from_numpy(g['metrics/y_val']), torch.from_numpy(g['metrics/y_test']))
    assert np.allclose(np.asarray(got, dtype=np.float64), g['metrics/result'], rtol=1e-12, atol=1e-12)


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


def test_link_split_generation(tmp_path):
    from magnet_amd.utils.edge_data import generate_dataset_3class, undirected_label2directed_label
    from magnet_amd.data import dsbm_cyclic
    import scipy.sparse as sp
    d = dsbm_cyclic(n_per_cluster=40, clusters=3, seed=1)
    n = 120
    ds = generate_dataset_3class(d.edge_index, n, str(tmp_path) + '/x', splits=2, probs=[0.15, 0.05], task=2, label_dim=3)
    A = sp.coo_matrix((np.ones(d.edge_index.size(1)), (d.edge_index[0].numpy(), d.edge_index[1].numpy())), shape=(n, n)).tocsr()
    for s in (0, 1):
        g = ds[s]['graph']
        assert g.dtype == torch.int64 and g.size(0) == 2
        assert all(A[int(a), int(b)] > 0 for a, b in g.t().tolist())                  # observed edges are real edges
        obs = set(map(tuple, g.t().tolist()))
        for part in ('train', 'validate', 'test'):
            pairs, lab = ds[s][part]['pairs'], ds[s][part]['label']
            assert len(pairs) == len(lab) and set(np.unique(lab)) <= {0, 1, 2}
            for (a, b), l in zip(pairs.tolist(), lab.tolist()):
                if l == 0:
                    assert A[a, b] > 0 and A[b, a] == 0
                elif l == 1:
                    assert A[b, a] > 0 and A[a, b] == 0
                else:
                    assert A[a, b] == 0 and A[b, a] == 0
        # held-out positive pairs are not in the observed graph
        for part in ('validate', 'test'):
            for (a, b), l in zip(ds[s][part]['pairs'].tolist(), ds[s][part]['label'].tolist()):
                if l < 2:
                    assert (a, b) not in obs and (b, a) not in obs
    again = generate_dataset_3class(d.edge_index, n, str(tmp_path) + '/x', splits=2, probs=[0.15, 0.05], task=2, label_dim=3)
    assert torch.equal(again[0]['graph'], ds[0]['graph'])                              # cached (read back inertly)
    for part in ('train', 'validate', 'test'):
        assert np.array_equal(again[1][part]['pairs'], ds[1][part]['pairs']) and np.array_equal(again[1][part]['label'], ds[1][part]['label'])
    # a split cache the reference ships is used as it is (build container only)
    import os, shutil
    ref = '/root/reference/dataset/data/tmp/citeseer/task1dim2prob15_5.pk'
    if os.path.isfile(ref):
        os.makedirs(str(tmp_path / 'cs'), exist_ok=True)
        shutil.copy(ref, str(tmp_path / 'cs' / 'task1dim2prob15_5.pk'))
        c = generate_dataset_3class(None, 0, str(tmp_path / 'cs') + '/', splits=10, task=1, label_dim=2)
        assert tuple(c[0]['graph'].shape) == (2, 3763) and c[0]['train']['pairs'].shape == (7345, 2)
    # existence task: two balanced classes
    p, l = undirected_label2directed_label(A, ds[0]['test']['pairs'], task=1)
    assert set(np.unique(l)) <= {0, 1} and abs(int((l == 0).sum()) - int((l == 1).sum())) <= 1


def test_link_driver_finds_split_caches_where_the_reference_keeps_them(tmp_path):
    """The link driver's split cache is looked up at data_path + dataset + '/' + subset (reference
    src/Edge_sparseMagnet.py:110), the place of the caches the reference ships; a cache found there is read
    without unpickling and used as it is, and only splits generated here go under log_root."""
    import os
    from magnet_amd.utils.edge_data import generate_dataset_3class, split_cache_prefix, split_cache_file
    data_path, log_root = str(tmp_path / 'data') + '/', str(tmp_path / 'logs') + '/'
    for dataset, shipped_name in (('citeseer/', 'citeseer/task1dim2prob15_5.pk'),
                                  ('WebKB/Cornell', 'WebKB/Cornell/Cornelltask1dim2prob15_5.pk')):
        assert split_cache_prefix(data_path, dataset, log_root, 1, 2, [0.15, 0.05]).startswith(log_root)     # nothing shipped yet
        # a cache in the reference's format at the reference's location
        g = torch.Generator().manual_seed(0)
        ei = torch.randint(0, 60, (2, 400), generator=g)
        made = generate_dataset_3class(ei, 60, str(tmp_path / 'made') + '/', splits=10, task=1, label_dim=2)
        os.makedirs(os.path.dirname(data_path + shipped_name), exist_ok=True)
        os.replace(split_cache_file(str(tmp_path / 'made') + '/', 1, 2, [0.15, 0.05]), data_path + shipped_name)
        prefix = split_cache_prefix(data_path, dataset, log_root, 1, 2, [0.15, 0.05])
        assert split_cache_file(prefix, 1, 2, [0.15, 0.05]) == data_path + dataset + ('/' if dataset.endswith('/') else '/Cornell') + \
            'task1dim2prob15_5.pk'
        assert os.path.samefile(split_cache_file(prefix, 1, 2, [0.15, 0.05]), data_path + shipped_name)
        # the driver's call: a different graph is passed, the cache wins and is returned as stored
        used = generate_dataset_3class(torch.zeros(2, 3, dtype=torch.long), 60, prefix, splits=10, probs=[0.15, 0.05], task=1, label_dim=2)
        assert torch.equal(used[3]['graph'], made[3]['graph']) and np.array_equal(used[3]['train']['pairs'], made[3]['train']['pairs'])
        assert split_cache_prefix(data_path, dataset, log_root, 2, 2, [0.15, 0.05]).startswith(log_root)     # other task: not shipped
    # the reference's own shipped cache through the same path (build container only)
    ref_root = '/root/reference/dataset/data/tmp/'
    if os.path.isfile(ref_root + 'citeseer/task1dim2prob15_5.pk'):
        prefix = split_cache_prefix(ref_root, 'citeseer/', log_root, 1, 2, [0.15, 0.05])
        assert prefix == ref_root + 'citeseer//'
        c = generate_dataset_3class(None, 0, prefix, splits=10, probs=[0.15, 0.05], task=1, label_dim=2)
        assert tuple(c[0]['graph'].shape) == (2, 3763) and c[0]['train']['pairs'].shape == (7345, 2)


def test_dsbm_meta_graphs():
    """The three meta-graph families of the reference's DSBM generators: which cluster pairs are
    linked and which way their edges point."""
    from magnet_amd import data as D
    m, c = 60, 5
    for meta in ('ordered', 'cyclic', 'fill'):
        g = D.dsbm(meta, n_per_cluster=m, clusters=c, p_in=0.2, p_inter=0.2, p_q=0.9, seed=3, train_per_class=5)
        src, dst = g.edge_index.numpy()
        assert g.x.shape == (m * c, 1) and src.max() < m * c and (src != dst).all()
        cnt = np.zeros((c, c), np.int64)
        np.add.at(cnt, (src // m, dst // m), 1)
        for a in range(c):
            for b in range(a + 1, c):
                tot, fwd = cnt[a, b] + cnt[b, a], cnt[a, b]
                closing = a == 0 and b == c - 1
                if meta == 'ordered' or b == a + 1:
                    assert tot > 0 and fwd / tot > 0.8
                elif closing:
                    assert tot > 0 and fwd / tot < 0.2           # last cluster -> first
                elif meta == 'fill':
                    assert tot > 0 and 0.35 < fwd / tot < 0.65
                else:
                    assert tot == 0
        a = D.dsbm(meta, n_per_cluster=m, clusters=c, seed=3).edge_index
        b = D.dsbm(meta, n_per_cluster=m, clusters=c, seed=3).edge_index
        assert torch.equal(a, b)                                  # seeded


def test_driver_dataset_names_and_saved_settings(tmp_path):
    """Dataset renaming and the best-setting override of the node driver
    (reference src/sparse_Magnet.py:245-303); the dictionaries are read without unpickling."""
    import pickle
    from magnet_amd import sparse_Magnet as drv
    args = drv.parse_args(['--dataset', 'syn/cyclic', '--p_q', '0.9', '--seed', '20', '--settings_dir', str(tmp_path)])
    drv.rename_syn_dataset(args)
    assert args.dataset == 'syn/cyclic90Seed20' and drv.syn_kind(args.dataset) == 'cyclic'
    assert drv.apply_saved_settings(args) is False and args.K == 2           # no file: command line stays
    table = {'syn/syn_tri_1': {'Magnet': 'log_root,../logs/,dataset,syn/syn_tri_1,epochs,3000,q,0.05,seed,0,K,1,layer,3,'
                                         'dropout,0.5,lr,0.01,l2,0.0005,activation,True,num_filter,16,,val_acc,0.9',
                               'GCN': 'lr,0.001'}}
    with open(tmp_path / 'Cyclic_setting_dict.pk', 'wb') as f:
        pickle.dump(table, f)
    assert drv.apply_saved_settings(args) is True
    assert (args.K, args.layer, args.num_filter, args.lr, args.q) == (1, 3, 16, 0.01, 0.05)
    a2 = drv.parse_args(['--dataset', 'syn/cyclic', '--p_q', '0.9', '-NS', '--settings_dir', str(tmp_path)])
    drv.rename_syn_dataset(a2)
    assert drv.apply_saved_settings(a2) is False and a2.lr == 5e-3            # -NS keeps the command line
    a3 = drv.parse_args(['--dataset', 'syn/syn', '--p_q', '-0.08', '--seed', '0'])
    drv.rename_syn_dataset(a3)
    assert a3.dataset == 'syn/syn2Seed0' and a3.p_inter == 0.08 and a3.gen_p_q == 0.95
    a4 = drv.parse_args(['--dataset', 'syn/fill', '--p_q', '0.8', '--seed', '5'])
    drv.rename_syn_dataset(a4)
    assert a4.dataset == 'syn/fill80Seed5' and drv.syn_kind(a4.dataset) == 'fill'
    # the reference's own dictionary, when the reference tree is present (build container only)
    ref = '/root/reference/src'
    import os
    if os.path.exists(os.path.join(ref, 'Cyclic_setting_dict.pk')):
        a5 = drv.parse_args(['--dataset', 'syn/cyclic', '--p_q', '0.95', '--settings_dir', ref])
        drv.rename_syn_dataset(a5)
        assert drv.apply_saved_settings(a5) is True
        assert (a5.K, a5.layer, a5.num_filter, a5.lr, a5.q) == (1, 3, 16, 0.01, 0.05)


def _write_geom_gcn_raw(folder, name, n=40, f=7, e=150, seed=0):
    import os
    rng = np.random.default_rng(seed)
    raw = os.path.join(folder, name.lower(), 'raw')
    os.makedirs(raw, exist_ok=True)
    x = (rng.random((n, f)) < 0.3).astype(int); y = rng.integers(0, 3, n)
    with open(os.path.join(raw, 'out1_node_feature_label.txt'), 'w') as fh:
        fh.write('node_id\tfeature\tlabel\n')
        for i in rng.permutation(n):                       # rows need not be in id order
            fh.write('%d\t%s\t%d\n' % (i, ','.join(map(str, x[i])), y[i]))
    src = rng.integers(0, n, e); dst = rng.integers(0, n, e)
    with open(os.path.join(raw, 'out1_graph_edges.txt'), 'w') as fh:
        fh.write('node_id\tnode_id\n')
        for a, b in zip(src, dst):
            fh.write('%d\t%d\n' % (a, b))
    for i in range(10):
        perm = np.random.default_rng(i).permutation(n)
        m = np.zeros((3, n), np.uint8); m[0, perm[:24]] = 1; m[1, perm[24:32]] = 1; m[2, perm[32:]] = 1
        np.savez(os.path.join(raw, '%s_split_0.6_0.2_%d.npz' % (name.lower(), i)), train_mask=m[0], val_mask=m[1], test_mask=m[2])
    return x, y, np.unique(np.stack([src, dst], 1), axis=0)


def test_geom_gcn_raw_loader(tmp_path):
    """WebKB-style raw files (the layout torch_geometric downloads, read by the reference through
    WebKB(root, name)): features, labels, directed de-duplicated edges, the ten 60/20/20 splits."""
    import os
    from magnet_amd import data as D
    x, y, edges = _write_geom_gcn_raw(str(tmp_path), 'Texas')
    g = D.load_geom_gcn_raw(str(tmp_path / 'WebKB' / 'Texas'), 'Texas', str(tmp_path))
    assert np.array_equal(g.x.numpy(), x.astype(np.float32)) and np.array_equal(g.y.numpy(), y)
    assert np.array_equal(g.edge_index.numpy().T, edges)
    assert g.train_mask.shape == (40, 10) and g.train_mask.dtype == torch.bool
    assert int(g.train_mask[:, 3].sum()) == 24 and not bool((g.train_mask & g.val_mask).any())
    with pytest.raises(FileNotFoundError):
        D.load_geom_gcn_raw(str(tmp_path), 'Cornell')
    ref = '/root/reference/dataset/data/tmp/'
    if os.path.isdir(ref + 'texas/raw'):                  # the copy the reference ships (build container only)
        t = D.load_geom_gcn_raw(ref + 'WebKB/Texas', 'Texas', ref)
        assert tuple(t.x.shape) == (183, 1703) and t.edge_index.size(1) == 325 and int(t.train_mask[:, 0].sum()) == 87


class _Bag:
    """Stand-in for torch_geometric.data.Data when writing a test pickle (never imported by the reader)."""


def test_load_syn_reads_data_pickles_without_unpickling(tmp_path):
    """The reference stores its synthetic and telegram datasets as pickled Data objects
    (src/utils/preprocess.py:103-111); load_syn walks the opcodes and rebuilds the tensors from the
    embedded storages with torch's restricted loader."""
    import os
    import pickle
    from magnet_amd import data as D
    g = torch.Generator().manual_seed(0)
    b = _Bag()
    b.x = torch.randn(30, 1, generator=g)
    b.edge_index = torch.randint(0, 30, (2, 200), generator=g)
    b.edge_attr = None
    b.y = torch.randint(0, 4, (30,), generator=g)
    b.edge_weight = np.arange(200, dtype=np.float64)
    base = torch.rand(30, 20, generator=g) < 0.5
    b.train_mask, b.val_mask, b.test_mask = base[:, :10], base[:, 10:], ~base[:, :10]     # non-contiguous views
    with open(tmp_path / 'toy.pk', 'wb') as f:
        pickle.dump(b, f)
    d = D.load_syn(str(tmp_path / 'toy'))[0]
    assert torch.equal(d.x, b.x) and torch.equal(d.edge_index, b.edge_index) and torch.equal(d.y, b.y)
    assert torch.equal(d.train_mask, b.train_mask) and torch.equal(d.val_mask, b.val_mask) and torch.equal(d.test_mask, b.test_mask)
    assert d.edge_weight.dtype == torch.float32 and torch.equal(d.edge_weight, torch.arange(200.0))
    ref = '/root/reference/dataset/data/tmp/syn/syn3Seed0'
    if os.path.isfile(ref + '.pk'):                       # build container only
        r = D.load_syn(ref)[0]
        assert tuple(r.x.shape) == (2500, 1) and r.edge_index.size(1) == 186984 and int(r.train_mask[:, 0].sum()) == 50


def test_random_splits_reproduce_the_reference_draws():
    """random_splits against masks drawn by the reference's train_test_split (src/utils/Citation.py:222-246)
    for the CiteSeer labels and DSBM-shaped labels, ten splits each: identical masks."""
    from magnet_amd.data import random_splits
    g = load_golden('split_cases.npz')
    for name in [str(s) for s in g['names']]:
        per_class, val = (int(v) for v in g[name + '/params'])
        tr, va, te = random_splits(g[name + '/labels'], 10, per_class, val)
        assert np.array_equal(tr.numpy(), g[name + '/train']), name
        assert np.array_equal(va.numpy(), g[name + '/val']), name
        assert np.array_equal(te.numpy(), g[name + '/test']), name


def test_pair_labelling_rule_matches_reference():
    """undirected_label2directed_label against outputs of the reference's function
    (src/utils/edge_data.py:122-161) for existence, three-class and direction labelling."""
    import scipy.sparse as sp
    from magnet_amd.utils.edge_data import undirected_label2directed_label
    g = load_golden('label_cases.npz')
    n = int(g['n'])
    A = sp.coo_matrix((np.ones(len(g['row'])), (g['row'], g['col'])), shape=(n, n)).tocsr()
    A.data[:] = 1.0
    for task in (1, 2, 3):
        p, l = undirected_label2directed_label(A, g['pairs'], task)
        assert np.array_equal(np.asarray(p), g['task%d/pairs' % task]) and np.array_equal(np.asarray(l), g['task%d/label' % task])


def _planted_digraph(n, n_comm, deg_in, deg_out, seed):
    """Communities with many edges inside and few between, vertex ids carrying no information about them."""
    g = torch.Generator().manual_seed(seed)
    comm = torch.randint(0, n_comm, (n,), generator=g)
    order = torch.argsort(comm, stable=True)
    counts = torch.bincount(comm, minlength=n_comm)
    starts = torch.cumsum(counts, 0) - counts
    s = torch.randint(0, n, (n * deg_in,), generator=g)
    r = torch.rand(n * deg_in, generator=g)
    d = order[starts[comm[s]] + (r * counts[comm[s]]).long().clamp(max=n - 1)]
    s2 = torch.randint(0, n, (n * deg_out,), generator=g)
    d2 = torch.randint(0, n, (n * deg_out,), generator=g)
    return torch.cat([s, s2]), torch.cat([d, d2]), comm


def test_relabelling_for_the_row_partition():
    """magnet_amd.partition: on a graph with communities hidden behind random vertex ids the balanced label
    propagation finds parts that cut far fewer edges than contiguous id blocks, within the load bound; the order
    derived from the parts is a permutation whose blocks are the parts; and the halo of every rank of the row
    partition shrinks accordingly."""
    from magnet_amd import partition as P
    from magnet_amd import dist as D
    n, world = 20000, 4
    src, dst, comm = _planted_digraph(n, world, 8, 1, 3)
    parts = P.balanced_label_propagation(src, dst, n, world)
    assert parts.shape == (n,) and int(parts.min()) >= 0 and int(parts.max()) < world
    again = P.balanced_label_propagation(src, dst, n, world)
    assert torch.equal(parts, again)                                            # no hidden random state
    blocks = torch.arange(n) * world // n
    cut_ids, cut_lp, cut_truth = P.cut_fraction(src, dst, blocks), P.cut_fraction(src, dst, parts), P.cut_fraction(src, dst, comm)
    assert cut_ids > 0.7 and cut_lp < 0.5 * cut_ids and cut_lp < cut_truth + 0.15
    w = torch.bincount(torch.cat([src, dst]), minlength=n).double() + 2
    load = torch.zeros(world, dtype=torch.float64).index_add_(0, parts, w)
    assert float(load.max() / load.mean()) <= 1.08

    new_of_old, old_of_new, bounds = P.order_from_parts(parts, world)
    assert torch.equal(torch.sort(new_of_old).values, torch.arange(n))
    assert torch.equal(new_of_old[old_of_new], torch.arange(n))
    assert bounds[0] == 0 and bounds[-1] == n and len(bounds) == world + 1
    for p in range(world):
        assert bool((parts[old_of_new[bounds[p]:bounds[p + 1]]] == p).all())

    def halo_rows(s, d, bnds):
        worst = 0
        for p in range(world):
            lo, hi = bnds[p], bnds[p + 1]
            mine_s, mine_d = (s >= lo) & (s < hi), (d >= lo) & (d < hi)
            other = torch.cat([d[mine_s & ~mine_d], s[mine_d & ~mine_s]])       # the Laplacian's support is symmetric
            worst = max(worst, int(torch.unique(other).numel()))
        return worst
    before = halo_rows(src, dst, [n * p // world for p in range(world + 1)])
    after = halo_rows(new_of_old[src], new_of_old[dst], bounds)
    assert after < 0.6 * before
    # a partitioner of one's own plugs in the same way: the true communities give the smallest halo
    t_new, _, t_bounds = P.order_from_parts(comm, world)
    assert halo_rows(t_new[src], t_new[dst], t_bounds) <= after


@pytest.mark.parametrize('dropout', [0.0, 0.5])
def test_shared_forward_epochs_on_a_toy_model(dropout):
    """train.SharedForwardEpochs against train.node_epoch on a small CPU model with the trunk / head split of ChebNet:
    identical losses, accuracies and parameters, one trunk pass per epoch (plus one) instead of two, and a kept pass
    that is dropped when a parameter changes behind its back."""
    from magnet_amd.train import node_epoch, SharedForwardEpochs, NodeTargets

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin, self.out, self.dropout, self.trunk_calls = torch.nn.Linear(6, 5), torch.nn.Linear(10, 3), dropout, 0

        def trunk(self, real, imag):
            self.trunk_calls += 1
            return torch.stack([torch.tanh(self.lin(real)), torch.tanh(self.lin(imag))], dim=1)

        def head(self, p):
            x = p.reshape(p.size(0), -1)
            if self.dropout > 0:
                x = torch.nn.functional.dropout(x, self.dropout, training=self.training)
            return torch.log_softmax(self.out(x), dim=1).t().unsqueeze(0)          # [1, classes, N]

        def forward(self, real, imag):
            return self.head(self.trunk(real, imag))

    g = torch.Generator().manual_seed(0)
    n = 40
    X, Y = torch.randn(n, 6, generator=g), torch.randn(n, 6, generator=g)
    label = torch.randint(0, 3, (1, n), generator=g)
    train = torch.rand(n, generator=g) < 0.6
    tr, va = NodeTargets(label, train), NodeTargets(label, ~train)

    def run(shared, epochs=6, poke_at=3):
        torch.manual_seed(1)
        net = Toy()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=5e-4)
        ep = SharedForwardEpochs(net, opt)
        torch.manual_seed(2)
        stats = []
        for i in range(epochs):
            if i == poke_at:
                with torch.no_grad():
                    net.lin.weight.mul_(1.05)                     # what load_state_dict does: same tensor, new version
            out = ep.node_epoch(X, Y, label, tr, va) if shared else node_epoch(net, opt, X, Y, label, tr, va)
            stats.append([float(t) for t in out])
        return stats, [p.detach().clone() for p in net.parameters()], net.trunk_calls

    s0, p0, calls0 = run(False)
    s1, p1, calls1 = run(True)
    assert s0 == s1
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))
    assert calls0 == 12 and calls1 == 6 + 1 + 1                  # one per epoch, the first epoch's own, one after the poke
