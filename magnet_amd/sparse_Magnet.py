"""Node-classification driver with the reference's command line (src/sparse_Magnet.py:26-53,64-239):

    python -m magnet_amd.sparse_Magnet --dataset syn/cyclic --p_q 0.95 --seed 10 --q 0.1 --K 2 -a \
           --num_filter 16 --layer 2 --epochs 300

Same flags, same loop (full-batch Adam, NLL loss on the training mask, validation forward, best /
latest checkpoints as model{split}.t7 / model_latest{split}.t7, early stopping after 500 epochs
without improvement, log{split}.csv, results array), on the HIP path.  Datasets that need
torch_geometric downloads are not available offline; `syn/...` graphs are generated (magnet_amd.data),
`telegram/telegram`, `citeseer_npz/`, `cora_ml/` and the WebKB / WikipediaNetwork graphs (Geom-GCN raw files,
e.g. --dataset WebKB/Texas) are read from --data_path when present.
"""
import argparse
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import data as D
from .layer.sparse_magnet import ChebNet
from .train import node_epoch, GraphedEpoch, NodeTargets, SharedForwardEpochs, GraphedSharedEpochs
from .utils.preprocess import geometric_dataset_sparse
from .utils.save_settings import write_log


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="MagNet Conv (sparse version)")
    parser.add_argument('--log_root', type=str, default='../logs/')
    parser.add_argument('--log_path', type=str, default='test')
    parser.add_argument('--data_path', type=str, default='../dataset/data/tmp/')
    parser.add_argument('--dataset', type=str, default='syn/cyclic')
    parser.add_argument('--epochs', type=int, default=3000)
    parser.add_argument('--q', type=float, default=0)
    parser.add_argument('--p_q', type=float, default=0.95)
    parser.add_argument('--p_inter', type=float, default=0.1)
    parser.add_argument('--method_name', type=str, default='Magnet')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--K', type=int, default=2)
    parser.add_argument('--layer', type=int, default=2)
    parser.add_argument('--dropout', type=float, default=0.0)
    parser.add_argument('--debug', '-D', action='store_true')
    parser.add_argument('--new_setting', '-NS', action='store_true')
    parser.add_argument('--lr', type=float, default=5e-3)
    parser.add_argument('--l2', type=float, default=5e-4)
    parser.add_argument('-activation', '-a', action='store_true')
    parser.add_argument('--num_filter', type=int, default=1)
    parser.add_argument('--randomseed', type=int, default=-1)
    parser.add_argument('--splits', type=int, default=10, help='how many of the 10 splits to run')
    parser.add_argument('--settings_dir', type=str, default='.',
                        help='where the saved best-setting dictionaries (syn_settings.pk, Cyclic_setting_dict.pk, '
                             'Cyclic_fill_setting_dict.pk) live; the reference opens them in the working directory')
    parser.add_argument('--separate_forwards', action='store_true',
                        help='(not in the reference) run the training and the validation forward of an epoch as two '
                             'passes, as the reference does; by default the validation pass through the convolutions '
                             'also serves the next training step (train.SharedForwardEpochs: same numbers)')
    parser.add_argument('--hip_graph', action='store_true',
                        help='(not in the reference) capture the epoch once and replay it as one HIP graph: the '
                             'small datasets spend their epoch in launch overhead')
    return parser.parse_args(argv)


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy sparse matrix -> float32 torch sparse COO tensor with int64 indices, in scipy's entry
    order and not coalesced (reference src/sparse_Magnet.py:55-62).  Only needed by callers that still hand
    the layers the reference's two lists of K+1 tensors; the drivers here pass the ChebBasis."""
    coo = sparse_mx.tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data), torch.Size(coo.shape))


def syn_kind(dataset):
    """Meta-graph family of a syn/... dataset name (reference src/sparse_Magnet.py:245-258)."""
    tail = dataset[4:]
    return 'ordered' if tail[:3] == 'syn' else 'cyclic' if tail[:6] == 'cyclic' else 'fill'


def rename_syn_dataset(args):
    """syn/<family> -> the reference's file stem with p_q and the seed in it
    (src/sparse_Magnet.py:245-258); the negative p_q codes select the p_inter = 0.08 / 0.05 variants."""
    if args.dataset[:3] != 'syn':
        return
    kind = syn_kind(args.dataset)
    args.gen_p_q = 0.95 if args.p_q in (-0.08, -0.05) else args.p_q      # the coded variants are p_q = 0.95 graphs
    if kind == 'ordered':
        if args.p_q == -0.08:
            args.p_inter = 0.08
            args.dataset = 'syn/syn2Seed' + str(args.seed)
        elif args.p_q == -0.05:
            args.p_inter = 0.05
            args.dataset = 'syn/syn3Seed' + str(args.seed)
        else:
            args.dataset = 'syn/syn' + str(int(100 * args.p_q)) + 'Seed' + str(args.seed)
    else:
        args.dataset = 'syn/' + kind + str(int(100 * args.p_q)) + 'Seed' + str(args.seed)


def apply_saved_settings(args):
    """Without --new_setting the reference replaces num_filter / layer / K / lr / q by the best setting it
    saved for the dataset and method (src/sparse_Magnet.py:262-303).  The dictionaries are pickles of
    plain strings; they are read by walking the opcodes, never unpickled.  A missing file, dataset or
    method leaves the command-line values in force (the reference would stop with an exception)."""
    if args.new_setting or args.dataset[:3] != 'syn':
        return False
    from .utils.inert_pickle import load_inert
    kind = syn_kind(args.dataset)
    try:
        if kind == 'ordered':
            fname = 'syn_settings.pk'
            if args.p_inter == 0.1:
                key = 'syn/syn' + str({0.95: 1, 0.9: 4, 0.85: 5, 0.8: 6, 0.75: 7, 0.7: 8, 0.65: 9, 0.6: 10}[args.p_q])
            elif args.p_inter in (0.08, 0.05):
                key = 'syn/syn2' if args.p_inter == 0.08 else 'syn/syn3'
            else:
                raise ValueError('Please input the correct p_q and p_inter values!')
        elif kind == 'cyclic':
            fname = 'Cyclic_setting_dict.pk'
            key = 'syn/syn_tri_' + str({0.95: 0, 0.9: 1, 0.85: 2, 0.8: 3, 0.75: 4, 0.7: 5, 0.65: 6}[args.p_q])
        else:
            fname = 'Cyclic_fill_setting_dict.pk'
            key = 'syn/syn_tri_' + str({0.95: 0, 0.9: 1, 0.85: 2, 0.8: 3}[args.p_q]) + '_fill'
        path = os.path.join(args.settings_dir, fname)
        if not os.path.exists(path):
            return False
        table = load_inert(open(path, 'rb').read())
        cur = table[key][args.method_name].split(',')
    except (KeyError, TypeError):
        return False
    for name, cast in (('num_filter', int), ('layer', int), ('K', int)):
        if name in cur:
            setattr(args, name, cast(cur[cur.index(name) + 1]))
    args.lr = float(cur[cur.index('lr') + 1])
    args.q = float(cur[cur.index('q') + 1])
    return True


def make_loader(args):
    """callable(root=..., name=...) -> [GraphData], the `dataset` argument of geometric_dataset_sparse."""
    head = args.dataset.split('/')[0]
    if os.path.isfile(args.data_path + args.dataset + '.pk') and head in ('syn', 'telegram'):
        return D.load_syn                                  # the reference's stored dataset, read inertly
    if head == 'syn':
        kind = syn_kind(args.dataset)
        per_class = {'ordered': 10, 'cyclic': 50, 'fill': 50}[kind]       # as the reference's generators
        return lambda root, name=None: [D.dsbm(kind, p_inter=args.p_inter, p_q=getattr(args, 'gen_p_q', args.p_q), seed=args.seed,
                                               train_per_class=per_class)]
    if head == 'telegram':
        return lambda root, name=None: [D.load_telegram(os.path.dirname(os.path.normpath(root)))]
    if head in ('cora_ml', 'citeseer_npz'):
        return lambda root, name=None: [D.load_npz_citation(root)]
    if head in ('WebKB', 'WikipediaNetwork'):
        return lambda root, name=None: [D.load_geom_gcn_raw(root, name, args.data_path)]
    raise ValueError('dataset %s needs torch_geometric downloads, which are not available here' % args.dataset)


def main(args):
    device = torch.device('cuda:0')
    if args.randomseed > 0:
        torch.manual_seed(args.randomseed)
    date_time = datetime.now().strftime('%m-%d-%H:%M:%S')
    log_path = os.path.join(args.log_root, args.log_path, args.save_name, date_time)
    os.makedirs(log_path, exist_ok=True)
    parts = args.dataset.split('/')
    subset = parts[1] if len(parts) > 1 else ''
    X, label, train_mask, val_mask, test_mask, L = geometric_dataset_sparse(
        args.q, args.K, root=args.data_path + args.dataset, subset=subset, dataset=make_loader(args),
        load_only=False, save_pk=False)
    _label_ = label - np.amin(label)
    cluster_dim = int(np.amax(_label_) + 1)
    label = torch.from_numpy(_label_[np.newaxis]).to(device)
    X_img = torch.FloatTensor(X).to(device)
    X_real = torch.FloatTensor(X).to(device)
    criterion = nn.NLLLoss()
    splits = min(train_mask.shape[1], args.splits)
    if len(test_mask.shape) == 1:
        test_mask = np.repeat(test_mask[:, np.newaxis], splits, 1)
    results = np.zeros((splits, 4))
    for split in range(splits):
        log_str_full = ''
        model = ChebNet(X_real.size(-1), L, None, K=args.K, label_dim=cluster_dim, layer=args.layer,
                        activation=args.activation, num_filter=args.num_filter, dropout=args.dropout).to(device)
        # the driver owns X_real / X_img and never writes them in place: keep the first layer's T_k(L) X across epochs
        model.Chebs[0].cache_input_propagation = True
        opt = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.l2, capturable=args.hip_graph)
        as_index = lambda m: torch.from_numpy(m).to(device).nonzero().squeeze(1) if m.dtype == np.bool_ \
            else torch.from_numpy(m).to(device).long()
        train_index, val_index, test_index = (as_index(m[:, split]) for m in (train_mask, val_mask, test_mask))
        # the split's labels with the other nodes ignored: loss and accuracy of the reference's
        # criterion(preds[:, :, index], label[:, index]) without the per-epoch gather and index-put
        train_targets, val_targets = NodeTargets(label, train_index), NodeTargets(label, val_index)
        body = lambda: node_epoch(model, opt, X_real, X_img, label, train_targets, val_targets, criterion)
        if not (args.hip_graph or args.separate_forwards):
            # same numbers, one pass through the convolutions per epoch: the validation forward is kept for the next
            # training step (same weights; dropout acts after the convolutions)
            shared = SharedForwardEpochs(model, opt)
            body = lambda: shared.node_epoch(X_real, X_img, label, train_targets, val_targets, criterion)
        if args.hip_graph:
            # the capture's warm-up epochs are real optimizer steps: rewind parameters and Adam state in place
            start = [p.detach().clone() for p in model.parameters()]
            if args.separate_forwards:
                body = GraphedEpoch(body)
            else:                                     # the shared-forward epoch as two HIP graphs
                body = GraphedSharedEpochs.for_nodes(model, opt, X_real, X_img, label, train_targets, val_targets, criterion)
            with torch.no_grad():
                for p, p0 in zip(model.parameters(), start):
                    p.copy_(p0)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        best_test_err, early_stopping = 1000.0, 0
        for epoch in range(args.epochs):
            start_time = time.time()
            train_loss, train_acc, test_loss, test_acc = body()
            train_acc, test_acc = train_acc.item(), test_acc.item()
            log_str_full += ("%d ,/, %d ,epoch," % (epoch, args.epochs)) + \
                'Train loss:, %.6f, acc:, %.3f,' % (train_loss.item(), train_acc) + \
                ' Test loss:, %.6f, acc:, %.3f,' % (test_loss.item(), test_acc) + \
                "---, %.4f, seconds ---" % (time.time() - start_time) + '\n'
            save_perform = test_loss.item()
            if save_perform <= best_test_err:
                early_stopping, best_test_err = 0, save_perform
                torch.save(model.state_dict(), log_path + '/model' + str(split) + '.t7')
            else:
                early_stopping += 1
            if early_stopping > 500 or epoch == (args.epochs - 1):
                torch.save(model.state_dict(), log_path + '/model_latest' + str(split) + '.t7')
                break
        body = shared = None                           # the kept pass (saved activations of one forward) goes with them
        write_log(vars(args), log_path)

        def evaluate(path):
            model.load_state_dict(torch.load(path, weights_only=True))
            model.eval()
            with torch.no_grad():
                pred = model(X_real, X_img).max(dim=1)[1]
            a = (pred[:, val_index] == label[:, val_index]).float().mean().item()
            b = (pred[:, test_index] == label[:, test_index]).float().mean().item()
            return pred, a, b

        pred, acc_val, acc_test = evaluate(log_path + '/model' + str(split) + '.t7')
        np.save(log_path + '/pred' + str(split), pred.to('cpu'))
        pred, acc_val_latest, acc_test_latest = evaluate(log_path + '/model_latest' + str(split) + '.t7')
        np.save(log_path + '/pred_latest' + str(split), pred.to('cpu'))
        logstr = 'val_acc: ' + str(np.round(acc_val, 3)) + ' test_acc: ' + str(np.round(acc_test, 3)) + \
            ' val_acc_latest: ' + str(np.round(acc_val_latest, 3)) + ' test_acc_latest: ' + str(np.round(acc_test_latest, 3))
        print(logstr)
        results[split] = [acc_val, acc_test, acc_val_latest, acc_test_latest]
        with open(log_path + '/log' + str(split) + '.csv', 'w') as f:
            f.write(log_str_full + logstr + '\n')
    return results


def run(argv=None):
    args = parse_args(argv)
    if args.debug:
        args.epochs = 1
    rename_syn_dataset(args)
    # the reference keeps result arrays beside its logs: <src>/../result_arrays next to the default ../logs
    dir_name = os.path.join(args.log_root, '..', 'result_arrays', args.log_path, args.dataset + '/')
    args.log_path = os.path.join(args.log_path, args.method_name, args.dataset)
    apply_saved_settings(args)
    args.save_name = args.method_name + 'lr' + str(int(args.lr * 1000)) + 'num_filters' + str(int(args.num_filter)) + \
        'q' + str(int(100 * args.q)) + 'layer' + str(int(args.layer))
    results = main(args)
    os.makedirs(dir_name, exist_ok=True)
    np.save(dir_name + args.save_name, results)
    return results


if __name__ == "__main__":
    run()
