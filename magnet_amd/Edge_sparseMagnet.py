"""Link-prediction driver with the reference's command line (src/Edge_sparseMagnet.py:25-53,73-309):

    python -m magnet_amd.Edge_sparseMagnet --dataset citeseer/ --task 2 --q 0.1 --K 2 -a --num_filter 16

Per split: Laplacian of the observed graph, degree features, ChebNet_Edge, full-batch Adam on the
training pairs, evaluation on the validation pairs each epoch, best / latest checkpoints, accuracy
(task 1, 3) or accuracy / AUC / F1 (task 2) at the end — all on the HIP path.
"""
import argparse
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from . import data as D
from .layer.sparse_magnet import ChebNet_Edge
from .utils.edge_data import in_out_degree, generate_dataset_3class, link_prediction_evaluation, acc, split_cache_prefix
from .utils.preprocess import to_edge_dataset_sparse
from .utils.save_settings import write_log
from .train import link_epoch, GraphedEpoch, SharedForwardEpochs, GraphedSharedEpochs


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="link prediction of MagNet (sparse version)")
    parser.add_argument('--log_root', type=str, default='../logs/')
    parser.add_argument('--log_path', type=str, default='test')
    parser.add_argument('--data_path', type=str, default='../dataset/data/tmp/')
    parser.add_argument('--dataset', type=str, default='syn/cyclic')
    parser.add_argument('--split_prob', type=lambda s: [float(item) for item in s.split(',')], default="0.15,0.05")
    parser.add_argument('--task', type=int, default=2, help='2: 3-class, 1: direction, 3: existence')
    parser.add_argument('--method_name', type=str, default='Magnet')
    parser.add_argument('--epochs', type=int, default=3000)
    parser.add_argument('--q', type=float, default=0)
    parser.add_argument('--K', type=int, default=2)
    parser.add_argument('--layer', type=int, default=2)
    parser.add_argument('--dropout', type=float, default=0.5)
    parser.add_argument('--debug', '-D', action='store_true')
    parser.add_argument('--not_norm', '-n', action='store_false')
    parser.add_argument('--lr', type=float, default=5e-3)
    parser.add_argument('--l2', type=float, default=5e-4)
    parser.add_argument('--num_class_link', type=int, default=3)
    parser.add_argument('-activation', '-a', action='store_true')
    parser.add_argument('--num_filter', type=int, default=1)
    parser.add_argument('--p_q', type=float, default=0.95)
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--splits', type=int, default=10)
    parser.add_argument('--true_degrees', action='store_true',
                        help='use real in/out degrees as features instead of the reference\'s as-written call')
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
    order and not coalesced (reference src/Edge_sparseMagnet.py:63-70).  Only needed by callers that still hand
    the layers the reference's two lists of K+1 tensors; the drivers here pass the ChebBasis."""
    coo = sparse_mx.tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data), torch.Size(coo.shape))


def load_graph(args):
    head = args.dataset.split('/')[0]
    if os.path.isfile(args.data_path + args.dataset + '.pk'):
        return D.load_syn(args.data_path + args.dataset)[0]      # the reference's stored dataset, read inertly
    if head == 'syn':
        return D.dsbm_cyclic(p_q=args.p_q, seed=args.seed)
    if head in ('cora_ml', 'citeseer'):
        name = 'cora_ml' if head == 'cora_ml' else 'citeseer_npz'
        return D.load_npz_citation(os.path.join(args.data_path, name, name + '.npz'))
    if head == 'telegram':
        return D.load_telegram(os.path.join(args.data_path, 'telegram'))
    if head in ('WebKB', 'WikipediaNetwork'):
        return D.load_geom_gcn_raw(args.data_path + args.dataset, args.dataset.split('/')[1], args.data_path)
    raise ValueError('dataset %s needs torch_geometric downloads, which are not available here' % args.dataset)


def main(args):
    device = torch.device('cuda:0')
    date_time = datetime.now().strftime('%m-%d-%H:%M:%S')
    log_path = os.path.join(args.log_root, args.log_path, args.save_name, date_time)
    os.makedirs(log_path, exist_ok=True)
    data = load_graph(args)
    edge_index = data.edge_index
    size = int(torch.max(edge_index).item()) + 1
    save_file = split_cache_prefix(args.data_path, args.dataset, args.log_root, args.task, args.num_class_link, args.split_prob)
    datasets = generate_dataset_3class(edge_index, size, save_file, splits=args.splits, probs=args.split_prob,
                                       task=args.task, label_dim=args.num_class_link)
    results = np.zeros((args.splits, 4)) if args.task != 2 else np.zeros((args.splits, 4, 5))
    for i in range(args.splits):
        edges = datasets[i]['graph']
        L = to_edge_dataset_sparse(args.q, edges, args.K, i, size, root=args.data_path + args.dataset, laplacian=True,
                                   norm=args.not_norm, gcn_appr=False)
        X_img = in_out_degree(edges, size, as_written=not args.true_degrees).to(device)
        X_real = X_img.clone()
        model = ChebNet_Edge(X_real.size(-1), L, None, K=args.K, label_dim=args.num_class_link, layer=args.layer,
                             activation=args.activation, num_filter=args.num_filter, dropout=args.dropout).to(device)
        # the driver owns X_real / X_img and never writes them in place: keep the first layer's T_k(L) X across epochs
        model.Chebs[0].cache_input_propagation = True
        opt = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.l2, capturable=args.hip_graph)
        y = {k: torch.from_numpy(datasets[i][k]['label']).long().to(device) for k in ('train', 'validate', 'test')}
        idx = {k: torch.from_numpy(np.asarray(datasets[i][k]['pairs'])).long().to(device) for k in ('train', 'validate', 'test')}
        best_test_err, early_stopping, log_str_full = 1000.0, 0, ''
        # one epoch of reference src/Edge_sparseMagnet.py:168-199: training step, then the validation forward
        body = lambda: link_epoch(model, opt, X_real, X_img, idx['train'], y['train'], idx['validate'], y['validate'])
        if not (args.hip_graph or args.separate_forwards):
            # same numbers, one pass through the convolutions per epoch: the validation forward is kept for the next
            # training step (same weights; pair gather and dropout act after the convolutions)
            shared = SharedForwardEpochs(model, opt)
            body = lambda: shared.link_epoch(X_real, X_img, idx['train'], y['train'], idx['validate'], y['validate'])
        if args.hip_graph:
            # the capture's warm-up epochs are real optimizer steps: rewind parameters and Adam state in place
            start = [p.detach().clone() for p in model.parameters()]
            if args.separate_forwards:
                body = GraphedEpoch(body)
            else:                                     # the shared-forward epoch as two HIP graphs
                body = GraphedSharedEpochs.for_links(model, opt, X_real, X_img, idx['train'], y['train'], idx['validate'],
                                                     y['validate'])
            with torch.no_grad():
                for p, p0 in zip(model.parameters(), start):
                    p.copy_(p0)
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        for epoch in range(args.epochs):
            start_time = time.time()
            if early_stopping > 500:
                break
            train_loss, train_acc, test_loss, test_acc = body()
            train_acc, test_acc = train_acc.item(), test_acc.item()
            log_str_full += ("%d / %d epoch" % (epoch, args.epochs)) + 'Train loss: %.6f, acc: %.3f' % (train_loss.item(), train_acc) + \
                ' Test loss: %.6f, acc: %.3f' % (test_loss.item(), test_acc) + "--- %.4f seconds ---" % (time.time() - start_time) + '\n'
            torch.save(model.state_dict(), log_path + '/model_latest' + str(i) + '.t7')
            save_perform = test_loss.item()
            if save_perform <= best_test_err:
                early_stopping, best_test_err = 0, save_perform
                torch.save(model.state_dict(), log_path + '/model' + str(i) + '.t7')
            else:
                early_stopping += 1
        body = shared = None                           # the kept pass (saved activations of one forward) goes with them
        write_log(vars(args), log_path)

        def outputs(path):
            model.load_state_dict(torch.load(path, weights_only=True))
            model.eval()
            with torch.no_grad():
                return model(X_real, X_img, idx['validate']), model(X_real, X_img, idx['test'])

        row = []
        for name in ('/model', '/model_latest'):
            ov, ot = outputs(log_path + name + str(i) + '.t7')
            if args.task != 2:
                row += [acc(ov[:, :2].max(dim=1)[1], y['validate']), acc(ot[:, :2].max(dim=1)[1], y['test'])]
            else:
                v, t = link_prediction_evaluation(ov, ot, y['validate'], y['test'])
                row += [v, t]
        results[i] = row
        print('split %d:' % i, np.round(np.asarray(row), 4).tolist())
        with open(log_path + '/log' + str(i) + '.csv', 'w') as f:
            f.write(log_str_full + '\n')
    return results


def run(argv=None):
    args = parse_args(argv)
    if args.debug:
        args.epochs = 1
    args.save_name = args.method_name + 'lr' + str(int(args.lr * 1000)) + 'num_filters' + str(int(args.num_filter)) + \
        'q' + str(int(100 * args.q)) + 'task' + str(int(args.task)) + 'link' + str(int(args.num_class_link))
    args.log_path = os.path.join(args.log_path, args.method_name, args.dataset)
    # reference src/Edge_sparseMagnet.py:317-327; kept beside the logs (<src>/../result_arrays next to ../logs)
    dir_name = os.path.join(args.log_root, '..', 'result_arrays', args.log_path, args.dataset + '/')
    results = main(args)
    os.makedirs(dir_name, exist_ok=True)
    np.save(dir_name + args.save_name, results)
    return results


if __name__ == "__main__":
    run()
