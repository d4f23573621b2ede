"""Graph inputs for the drivers: plain attribute bags with the fields the reference's glue reads
(edge_index [2,E], edge_weight, x, y, train/val/test masks [N,10]) — reference
src/utils/preprocess.py:165-171 — from files that can be read without unpickling, and seeded
synthetic generators for the configurations whose files the reference does not ship.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch


class GraphData:
    """Attribute bag standing in for torch_geometric.data.Data on this path."""

    def __init__(self, **kw):
        self.edge_weight = None
        for k, v in kw.items():
            setattr(self, k, v)


def random_splits(labels, n_splits=10, train_per_class=20, val_size=500, seed=0):
    """[N, n_splits] boolean train / val / test masks, drawn exactly as the reference draws them
    (src/utils/Citation.py:128-246, called at :56 and from src/utils/generate_cyclic.py:154): split s uses
    numpy's legacy RandomState(seed + s); per class in ascending order `train_per_class` nodes are chosen
    without replacement from the class's nodes in index order, then `val_size` nodes from the sorted rest,
    and everything left is the test set.  (A class smaller than `train_per_class` contributes all its
    nodes; the reference would raise there.)"""
    labels = np.asarray(labels).astype(np.int64)
    n = len(labels)
    tr = np.zeros((n, n_splits), bool); va = np.zeros((n, n_splits), bool); te = np.zeros((n, n_splits), bool)
    for s in range(n_splits):
        rs = np.random.RandomState(seed + s)
        chosen = []
        for c in range(int(labels.max()) + 1 if n else 0):
            idx = np.nonzero(labels == c)[0]
            chosen.append(rs.choice(idx, min(train_per_class, len(idx)), replace=False) if len(idx) else idx)
        chosen = np.concatenate(chosen) if chosen else np.zeros(0, np.int64)
        rest = np.setdiff1d(np.arange(n), chosen)
        v = rs.choice(rest, min(val_size, len(rest)), replace=False)
        tr[chosen, s] = True
        va[v, s] = True
        te[np.setdiff1d(rest, v), s] = True
    return torch.from_numpy(tr), torch.from_numpy(va), torch.from_numpy(te)


def dsbm_cyclic(n_per_cluster=500, clusters=5, p_in=0.1, p_inter=0.1, p_q=0.95, seed=10, train_per_class=50):
    """Cyclic directed stochastic block model (the model of reference
    src/utils/generate_cyclic.py:79-143 with cycle=True, 279-290): undirected SBM edges with
    probability p_in inside a cluster and p_inter between clusters that are adjacent on the cycle
    0 -> 1 -> ... -> c-1 -> 0; inside a cluster an edge gets a uniformly random direction, between
    adjacent clusters it points along the cycle with probability p_q.  Features are N(0,1) [N,1]
    (generate_cyclic.py:178-179).  Seeded with numpy's default_rng."""
    rng = np.random.default_rng(seed)
    n = n_per_cluster * clusters
    src, dst = [], []
    for a in range(clusters):
        # inside cluster a
        iu = np.triu(rng.random((n_per_cluster, n_per_cluster)) < p_in, 1)
        x, y = np.nonzero(iu)
        flip = rng.random(len(x)) < 0.5
        s = np.where(flip, y, x) + a * n_per_cluster; d = np.where(flip, x, y) + a * n_per_cluster
        src.append(s); dst.append(d)
        # between a and its successor on the cycle
        b = (a + 1) % clusters
        if clusters > 1 and not (clusters == 2 and a == 1):
            x, y = np.nonzero(rng.random((n_per_cluster, n_per_cluster)) < p_inter)
            fwd = rng.random(len(x)) < p_q
            s = np.where(fwd, x + a * n_per_cluster, y + b * n_per_cluster)
            d = np.where(fwd, y + b * n_per_cluster, x + a * n_per_cluster)
            src.append(s); dst.append(d)
    edge_index = torch.from_numpy(np.vstack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64))
    y = np.repeat(np.arange(clusters), n_per_cluster)
    x = torch.from_numpy(rng.normal(0, 1.0, (n, 1)).astype(np.float32))
    tr, va, te = random_splits(y, train_per_class=train_per_class, val_size=500)      # split s is seeded with s, as in the reference
    return GraphData(x=x, edge_index=edge_index, edge_weight=None, y=torch.from_numpy(y).long(),
                     train_mask=tr, val_mask=va, test_mask=te)


def dsbm(meta='ordered', n_per_cluster=500, clusters=5, p_in=0.1, p_inter=0.1, p_q=0.95, seed=0, train_per_class=10):
    """Directed stochastic block model with the reference's three meta-graphs
    (src/utils/generate_cyclic.py:79-143, src/utils/generate_fill.py:6-78):

      'ordered' — every pair of clusters a < b is linked with probability p_inter and an edge points
                  a -> b with probability p_q (the syn/syn* family, cycle=False);
      'cyclic'  — only clusters adjacent on the cycle 0 -> 1 -> ... -> c-1 -> 0 are linked, edges
                  point along the cycle with probability p_q (cycle=True);
      'fill'    — as 'cyclic', and the other cluster pairs are linked too, with a uniformly random
                  direction (cycle=True, fill=True).
    Inside a cluster (probability p_in) directions are uniformly random.  Features N(0,1) [N,1]."""
    if meta not in ('ordered', 'cyclic', 'fill'):
        raise ValueError('meta must be ordered, cyclic or fill')
    if meta == 'cyclic':
        return dsbm_cyclic(n_per_cluster, clusters, p_in, p_inter, p_q, seed, train_per_class)
    rng = np.random.default_rng(seed)
    m = n_per_cluster
    src, dst = [], []
    for a in range(clusters):
        x, y = np.nonzero(np.triu(rng.random((m, m)) < p_in, 1))
        flip = rng.random(len(x)) < 0.5
        src.append(np.where(flip, y, x) + a * m); dst.append(np.where(flip, x, y) + a * m)
        for b in range(a + 1, clusters):
            if meta == 'ordered' or b == a + 1:
                p_ab = p_q                                  # a -> b
            elif a == 0 and b == clusters - 1:
                p_ab = 1.0 - p_q                            # closing arc of the cycle: b -> a
            else:
                p_ab = 0.5                                  # 'fill': undirected noise between non-adjacent clusters
            x, y = np.nonzero(rng.random((m, m)) < p_inter)
            fwd = rng.random(len(x)) < p_ab
            src.append(np.where(fwd, x + a * m, y + b * m)); dst.append(np.where(fwd, y + b * m, x + a * m))
    n = m * clusters
    edge_index = torch.from_numpy(np.vstack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64))
    y = np.repeat(np.arange(clusters), m)
    x = torch.from_numpy(rng.normal(0, 1.0, (n, 1)).astype(np.float32))
    tr, va, te = random_splits(y, train_per_class=train_per_class, val_size=500)      # split s is seeded with s, as in the reference
    return GraphData(x=x, edge_index=edge_index, edge_weight=None, y=torch.from_numpy(y).long(),
                     train_mask=tr, val_mask=va, test_mask=te)


def load_npz_citation(path, n_splits=10):
    """`.npz` citation graphs (cora_ml, citeseer_npz; reference src/utils/Citation.py:36-73,96-110):
    adjacency, dense features, labels, 20-per-class / 500 / rest splits.  allow_pickle stays off."""
    if os.path.isdir(path):
        path = os.path.join(path, os.path.basename(os.path.normpath(path)) + '.npz')
    z = np.load(path, allow_pickle=False)
    A = sp.csr_matrix((z['adj_data'], z['adj_indices'], z['adj_indptr']), shape=tuple(z['adj_shape'])).tocoo()
    X = sp.csr_matrix((z['attr_data'], z['attr_indices'], z['attr_indptr']), shape=tuple(z['attr_shape']))
    labels = z['labels'].astype(np.int64)
    tr, va, te = random_splits(labels, n_splits, 20, 500, seed=0)
    return GraphData(x=torch.from_numpy(np.asarray(X.todense())).float(),
                     edge_index=torch.from_numpy(np.vstack([A.row, A.col]).astype(np.int64)), edge_weight=None,
                     y=torch.from_numpy(labels), train_mask=tr, val_mask=va, test_mask=te)


def load_telegram(folder, n_splits=10):
    """telegram_adj.npz (weighted adjacency) + telegram_labels.npy, the inputs of reference
    src/utils/telegram_prepare.py; features N(0,1) [N,1]; 60/20/20 random splits."""
    A = sp.load_npz(os.path.join(folder, 'telegram_adj.npz')).tocoo()
    labels = np.load(os.path.join(folder, 'telegram_labels.npy'), allow_pickle=False).astype(np.int64).ravel()
    n = A.shape[0]
    rng = np.random.default_rng(0)
    tr = np.zeros((n, n_splits), bool); va = np.zeros((n, n_splits), bool); te = np.zeros((n, n_splits), bool)
    for s in range(n_splits):
        perm = np.random.default_rng(s).permutation(n)
        a, b = int(0.6 * n), int(0.8 * n)
        tr[perm[:a], s] = True; va[perm[a:b], s] = True; te[perm[b:], s] = True
    return GraphData(x=torch.from_numpy(rng.normal(0, 1, (n, 1)).astype(np.float32)),
                     edge_index=torch.from_numpy(np.vstack([A.row, A.col]).astype(np.int64)),
                     edge_weight=A.data.astype(np.int64), y=torch.from_numpy(labels),
                     train_mask=torch.from_numpy(tr), val_mask=torch.from_numpy(va), test_mask=torch.from_numpy(te))


def load_syn(root, name=None):
    """`root + '.pk'`: a dataset the reference stores as a pickled torch_geometric Data object — its
    synthetic DSBM graphs `syn/*.pk` and `telegram/telegram.pk` (reference src/utils/preprocess.py:103-111).
    The file is read by walking the pickle opcodes (utils.inert_pickle): nothing in it is executed."""
    from .utils.inert_pickle import read_object_state
    d = read_object_state(root + '.pk')
    t = lambda v: v if v is None or torch.is_tensor(v) else torch.from_numpy(np.asarray(v))
    ew = d.get('edge_weight')
    g = GraphData(x=t(d['x']).float(), edge_index=t(d['edge_index']).long(), y=t(d['y']).long(),
                  train_mask=t(d['train_mask']).bool(), val_mask=t(d['val_mask']).bool(), test_mask=t(d['test_mask']).bool())
    g.edge_weight = None if ew is None else t(ew).float()
    return [g]


def load_geom_gcn_raw(root, name, data_path=None):
    """WebKB / WikipediaNetwork graphs in the Geom-GCN raw layout that torch_geometric downloads and the
    reference reads through it (src/sparse_Magnet.py:77-80; the reference tree ships texas and wisconsin):
        <dir>/<name>/raw/out1_node_feature_label.txt   node_id <tab> f0,f1,... <tab> label   (header line)
        <dir>/<name>/raw/out1_graph_edges.txt          src <tab> dst                         (header line)
        <dir>/<name>/raw/<name>_split_0.6_0.2_<i>.npz  train_mask / val_mask / test_mask, i = 0..9
    with <name> lower-cased and <dir> = root or data_path.  Edges stay directed; duplicates are dropped
    (torch_geometric coalesces them)."""
    low = name.lower()
    raw = None
    for base in (root, data_path):
        if base and os.path.isfile(os.path.join(base, low, 'raw', 'out1_graph_edges.txt')):
            raw = os.path.join(base, low, 'raw')
            break
    if raw is None:
        raise FileNotFoundError('no Geom-GCN raw files for %s under %s or %s (torch_geometric would download them; '
                                'there is no network here)' % (name, root, data_path))
    feats, labels = {}, {}
    with open(os.path.join(raw, 'out1_node_feature_label.txt')) as f:
        next(f)
        for line in f:
            parts = line.rstrip('\n').split('\t')
            if len(parts) < 3:
                continue
            i = int(parts[0])
            feats[i] = np.array(parts[1].split(','), dtype=np.float32)
            labels[i] = int(parts[2])
    n = max(feats) + 1
    x = np.stack([feats[i] for i in range(n)])
    y = np.array([labels[i] for i in range(n)], dtype=np.int64)
    e = np.loadtxt(os.path.join(raw, 'out1_graph_edges.txt'), dtype=np.int64, skiprows=1, ndmin=2)
    e = np.unique(e, axis=0)                                   # (src, dst) sorted, duplicates removed
    masks = {'train_mask': [], 'val_mask': [], 'test_mask': []}
    for i in range(10):
        z = np.load(os.path.join(raw, '%s_split_0.6_0.2_%d.npz' % (low, i)), allow_pickle=False)
        for k in masks:
            masks[k].append(z[k].astype(bool))
    tr, va, te = (np.stack(masks[k], axis=1) for k in ('train_mask', 'val_mask', 'test_mask'))
    return GraphData(x=torch.from_numpy(x), edge_index=torch.from_numpy(e.T.copy()), edge_weight=None,
                     y=torch.from_numpy(y), train_mask=torch.from_numpy(tr), val_mask=torch.from_numpy(va),
                     test_mask=torch.from_numpy(te))


def load_edge_csv(edges_path, labels_path=None):
    """`i,j,w` edge list with 1-based ids (dataset/cora/cora.edges) and optional `i,label` file."""
    e = np.loadtxt(edges_path, delimiter=',', dtype=np.int64)
    edge_index = torch.from_numpy((e[:, :2] - 1).T.copy())
    n = int(edge_index.max()) + 1
    y = None
    if labels_path is not None:
        l = np.loadtxt(labels_path, delimiter=',', dtype=np.int64)
        y = np.zeros(n, np.int64); y[l[:, 0] - 1] = l[:, 1]
        y = torch.from_numpy(y - y.min())
    return GraphData(edge_index=edge_index, edge_weight=None, y=y, x=None)
