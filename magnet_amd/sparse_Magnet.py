"""Node-classification driver with the reference's command line (src/sparse_Magnet.py:26-53,64-239):

    python -m magnet_amd.sparse_Magnet --dataset syn/cyclic --p_q 0.95 --seed 10 --q 0.1 --K 2 -a \
           --num_filter 16 --layer 2 --epochs 300

Same flags, same loop (full-batch Adam, NLL loss on the training mask, validation forward, best /
latest checkpoints as model{split}.t7 / model_latest{split}.t7, early stopping after 500 epochs
without improvement, log{split}.csv, results array), on the HIP path.  Datasets that need
torch_geometric downloads are not available offline; `syn/cyclic` is generated (magnet_amd.data),
`telegram/telegram`, `citeseer_npz/`, `cora_ml/` are read from --data_path when present.
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
from .utils.preprocess import geometric_dataset_sparse


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
    return parser.parse_args(argv)


def make_loader(args):
    """callable(root=..., name=...) -> [GraphData], the `dataset` argument of geometric_dataset_sparse."""
    head = args.dataset.split('/')[0]
    if head == 'syn':
        return lambda root, name=None: [D.dsbm_cyclic(p_inter=args.p_inter, p_q=args.p_q, seed=args.seed)]
    if head == 'telegram':
        return lambda root, name=None: [D.load_telegram(os.path.dirname(os.path.normpath(root)))]
    if head in ('cora_ml', 'citeseer_npz'):
        return lambda root, name=None: [D.load_npz_citation(root)]
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
        opt = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.l2)
        train_index = torch.from_numpy(train_mask[:, split]).to(device)
        val_index = torch.from_numpy(val_mask[:, split]).to(device)
        test_index = torch.from_numpy(test_mask[:, split]).to(device)
        best_test_err, early_stopping = 1000.0, 0
        for epoch in range(args.epochs):
            start_time = time.time()
            model.train()
            preds = model(X_real, X_img)
            train_loss = criterion(preds[:, :, train_index], label[:, train_index])
            pred_label = preds.max(dim=1)[1]
            train_acc = (pred_label[:, train_index] == label[:, train_index]).float().mean().item()
            opt.zero_grad()
            train_loss.backward()
            opt.step()
            model.eval()
            with torch.no_grad():
                preds = model(X_real, X_img)
            pred_label = preds.max(dim=1)[1]
            test_loss = criterion(preds[:, :, val_index], label[:, val_index])
            test_acc = (pred_label[:, val_index] == label[:, val_index]).float().mean().item()
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
    args.log_path = os.path.join(args.log_path, args.method_name, args.dataset)
    args.save_name = args.method_name + 'lr' + str(int(args.lr * 1000)) + 'num_filters' + str(int(args.num_filter)) + \
        'q' + str(int(100 * args.q)) + 'layer' + str(int(args.layer))
    return main(args)


if __name__ == "__main__":
    run()
