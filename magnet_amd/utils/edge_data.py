"""Link-prediction helpers on the hot path's edge (reference src/utils/edge_data.py)."""
import os

import numpy as np
import torch


def in_out_degree(edge_index, size, as_written=True):
    """Degree features [size, 2] (reference src/utils/edge_data.py:115-120).

    as_written=True reproduces the reference exactly as its driver calls it
    (src/Edge_sparseMagnet.py:135 passes a [2, E] tensor): the function reads
    edge_index[:, 0] and edge_index[:, 1] as row / column lists and len(edge_index) as the
    number of entries, so for a [2, E] input only two entries (0-th and 1-st column pairs) are
    counted.  as_written=False gives the intended in/out degrees of a [2, E] edge list.
    The result lives on edge_index's device."""
    ei = edge_index if isinstance(edge_index, torch.Tensor) else torch.as_tensor(np.asarray(edge_index))
    dev = ei.device
    if as_written:
        rows, cols = ei[:, 0].long(), ei[:, 1].long()
        k = ei.size(0)
        rows, cols = rows[:k], cols[:k]
    else:
        rows, cols = ei[0].long(), ei[1].long()
    ones = torch.ones(rows.numel(), dtype=torch.float32, device=dev)
    # the reference's names are swapped too: "in_degree" is the row sum (out-degree), "out_degree" the column sum
    in_degree = torch.zeros(size, dtype=torch.float32, device=dev).index_add_(0, rows, ones)
    out_degree = torch.zeros(size, dtype=torch.float32, device=dev).index_add_(0, cols, ones)
    return torch.stack([in_degree, out_degree], dim=1)


def acc(pred, label):
    """Fraction of correct predictions (reference src/Edge_sparseMagnet.py:55-60)."""
    return float((pred == label).sum().item()) / max(len(pred), 1)


def link_prediction_evaluation(out_val, out_test, y_val, y_test):
    """Accuracy (all classes), then accuracy / AUC / micro-F1 / macro-F1 on the pairs whose label
    is 0 or 1, for validation and test (reference src/utils/edge_data.py:538-586; the metrics are
    host-side sklearn calls there too).  Inputs are log-probabilities [M, 2 or 3] and labels [M]."""
    from sklearn.metrics import accuracy_score, f1_score, roc_auc_score

    def one(out, y):
        out = torch.exp(out).detach().to('cpu').numpy()
        y = y.detach().to('cpu').numpy()
        acc_full = accuracy_score(np.argmax(out, axis=1), y)
        out = out[y < 2, :2]
        y = y[y < 2]
        with np.errstate(divide='ignore', invalid='ignore'):
            prob = out[:, 0] / (out[:, 0] + out[:, 1])
        prob = np.nan_to_num(prob, nan=0.5, posinf=0)
        pred = np.argmax(out, axis=1)
        return [acc_full, accuracy_score(pred, y), roc_auc_score(y, prob), f1_score(pred, y, average='micro'),
                f1_score(pred, y, average='macro')]

    return [one(out_val, y_val), one(out_test, y_test)]


# ---------------------------------------------------------------------------------------------
# link splits (reference src/utils/edge_data.py:122-265).  The reference delegates the edge
# removal / negative sampling to stellargraph's EdgeSplitter, which is not installable here; the
# procedure below restates what that call is documented to do ("global" method) with numpy's
# seeded generator, and keeps the reference's labelling rule and output dictionary layout.
# ---------------------------------------------------------------------------------------------
def _split_undirected(und, n, p, seed, keep_connected):
    """Remove int(p * |edges|) undirected edges (never an edge of a spanning forest when
    keep_connected) and draw as many node pairs that are not adjacent.  und: [U, 2] with a < b.
    Returns (remaining edges [U', 2], sampled pairs [2k, 2] = removed edges then non-edges)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import minimum_spanning_tree
    rng = np.random.default_rng(seed)
    u = len(und)
    k = int(p * u)
    removable = np.ones(u, bool)
    if keep_connected and u:
        g = sp.coo_matrix((np.ones(u), (und[:, 0], und[:, 1])), shape=(n, n)).tocsr()
        t = minimum_spanning_tree(g).tocoo()
        tree = set(zip(np.minimum(t.row, t.col).tolist(), np.maximum(t.row, t.col).tolist()))
        removable = np.array([(a, b) not in tree for a, b in und.tolist()], bool)
    cand = np.nonzero(removable)[0]
    k = min(k, len(cand))
    gone = rng.choice(cand, size=k, replace=False) if k else np.zeros(0, np.int64)
    keep = np.ones(u, bool); keep[gone] = False
    present = set(map(tuple, und.tolist()))
    neg = []
    seen = set()
    while len(neg) < k:
        a = rng.integers(0, n, size=2 * (k - len(neg)) + 8); b = rng.integers(0, n, size=len(a))
        for x, y in zip(a.tolist(), b.tolist()):
            if x == y:
                continue
            key = (min(x, y), max(x, y))
            if key in present or key in seen:
                continue
            seen.add(key); neg.append((x, y))
            if len(neg) == k:
                break
    pairs = np.concatenate([und[gone], np.array(neg, np.int64).reshape(-1, 2)]) if k else np.zeros((0, 2), np.int64)
    return und[keep], pairs


def undirected_label2directed_label(adj, edge_pairs, task):
    """Orientation labels of sampled pairs (reference :122-161): a pair joined by a one-way edge is
    labelled alternately 0 (listed along the edge) and 1 (listed against it), reciprocal pairs are
    dropped, absent pairs get 2; for the existence tasks (task != 2) absent pairs become class 1
    and class 1 is thinned to the size of class 0 with default_rng(1000)."""
    pairs = np.asarray(edge_pairs).reshape(-1, 2)
    a, b = pairs[:, 0], pairs[:, 1]
    if len(pairs):
        fwd = np.asarray(adj[a, b]).ravel() > 0
        bwd = np.asarray(adj[b, a]).ravel() > 0
    else:
        fwd = bwd = np.zeros(0, bool)
    one_way = fwd ^ bwd
    labels = np.full(len(pairs), 2, dtype=np.int32)            # absent pairs
    labels[fwd & bwd] = -1                                      # reciprocal pairs are dropped
    order = np.cumsum(one_way) - 1                              # running count over the one-way pairs, in list order
    labels[one_way] = (order[one_way] % 2).astype(np.int32)     # alternately 0 (along the edge) and 1 (against it)
    new_edge_pairs = pairs.copy()
    along = np.where(fwd[:, None], pairs, pairs[:, ::-1])       # the pair listed along its edge
    sel0, sel1 = one_way & (labels == 0), one_way & (labels == 1)
    new_edge_pairs[sel0] = along[sel0]
    new_edge_pairs[sel1] = along[sel1][:, ::-1]
    if task != 2:
        labels[labels == 2] = 1
        neg = np.where(labels == 1)[0]
        rng = np.random.default_rng(1000)
        neg_half = rng.choice(neg, size=max(len(neg) - int(np.sum(labels == 0)), 0), replace=False)
        labels[neg_half] = -1
    return new_edge_pairs[labels >= 0], labels[labels >= 0]


def split_cache_file(save_path, task, label_dim, probs):
    """Name of a link-split cache as the reference forms it (src/utils/edge_data.py:164)."""
    return save_path + 'task' + str(task) + 'dim' + str(label_dim) + 'prob' + str(int(probs[0] * 100)) + '_' + \
        str(int(probs[1] * 100)) + '.pk'


def split_cache_prefix(data_path, dataset, log_root, task, label_dim, probs):
    """Where the link driver keeps its splits.  The reference looks under the dataset folder,
    data_path + dataset + '/' + subset (src/Edge_sparseMagnet.py:110), which is where its shipped caches live
    (citeseer/task1dim2prob15_5.pk, WebKB/Cornell/Cornelltask2dim2prob15_5.pk ...): a cache found there is
    used as it is.  Splits generated here (the reference delegates the draw to stellargraph's EdgeSplitter;
    ours comes from a seeded numpy generator) are written under log_root instead, so that they can never be
    mistaken for, or overwrite, the reference's."""
    parts = dataset.split('/')
    subset = parts[1] if len(parts) > 1 else ''
    reference_prefix = data_path + dataset + '/' + subset
    if os.path.exists(split_cache_file(reference_prefix, task, label_dim, probs)):
        return reference_prefix
    return os.path.join(log_root, 'splits', dataset.replace('/', '_')) + '/' + subset


def generate_dataset_3class(edge_index, size, save_path, splits=10, probs=(0.15, 0.05), task=2, label_dim=2):
    """{split: {'graph': LongTensor [2, E_obs], 'undirected': ndarray, 'train' | 'validate' | 'test':
    {'pairs': int [M, 2], 'label': int32 [M]}}} — the layout of reference :221-239, cached under the
    reference's file name (:259-263).  An existing cache — written here or shipped with the reference,
    e.g. dataset/data/tmp/citeseer/task1dim2prob15_5.pk — is read by walking its pickle opcodes
    (utils.inert_pickle), never unpickled."""
    import pickle as pk
    import scipy.sparse as sp
    save_file = split_cache_file(save_path, task, label_dim, probs)
    if os.path.exists(save_file):
        from .inert_pickle import load_inert, to_numpy
        with open(save_file, 'rb') as f:
            cached = to_numpy(load_inert(f.read()))
        if isinstance(cached, dict) and all(i in cached for i in range(splits)):
            return cached
    ei = edge_index.cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    A = sp.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(size, size), dtype=np.float32).tocsr()
    S = (A + A.T).tocoo()
    und = np.stack([S.row, S.col], 1)
    und = und[und[:, 0] < und[:, 1]].astype(np.int64)
    out = {}
    for ind in range(splits):
        g_test, ids_test = _split_undirected(und, size, float(probs[0]), ind, True)
        ids_test, labels_test = undirected_label2directed_label(A, ids_test, task)
        g_val, ids_val = _split_undirected(g_test, size, float(probs[1]), ind, True)
        ids_val, labels_val = undirected_label2directed_label(A, ids_val, task)
        _, ids_train = _split_undirected(g_val, size, 0.99, ind, False)
        ids_train, labels_train = undirected_label2directed_label(A, ids_train, task)
        # back to the stored direction (reference :197-203; a reciprocal pair ends up listed as (b, a))
        ab = np.asarray(A[g_val[:, 0], g_val[:, 1]]).ravel() > 0 if len(g_val) else np.zeros(0, bool)
        ba = np.asarray(A[g_val[:, 1], g_val[:, 0]]).ravel() > 0 if len(g_val) else np.zeros(0, bool)
        observed = np.zeros((len(g_val), 2), dtype=np.int32)
        observed[ab] = g_val[ab]
        observed[ba] = g_val[ba][:, ::-1]
        if label_dim == 2:
            ids_train, labels_train = ids_train[labels_train < 2], labels_train[labels_train < 2]
            ids_test, labels_test = ids_test[labels_test < 2], labels_test[labels_test < 2]
            ids_val, labels_val = ids_val[labels_val < 2], labels_val[labels_val < 2]
        out[ind] = {'graph': torch.from_numpy(observed.T.copy()).long(),
                    'undirected': np.concatenate([und[:, ::-1], und]).astype(np.int32),
                    'train': {'pairs': ids_train, 'label': labels_train},
                    'validate': {'pairs': ids_val, 'label': labels_val},
                    'test': {'pairs': ids_test, 'label': labels_test}}
    os.makedirs(os.path.dirname(save_file) or '.', exist_ok=True)
    with open(save_file, 'wb') as f:
        pk.dump(out, f, protocol=pk.HIGHEST_PROTOCOL)
    return out
