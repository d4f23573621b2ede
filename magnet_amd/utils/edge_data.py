"""Link-prediction helpers on the hot path's edge (reference src/utils/edge_data.py)."""
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
