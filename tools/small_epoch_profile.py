"""Configuration-2 epochs (2 500-node cyclic DSBM, K = 2, 16 filters) replayed as HIP graphs and nothing else, for
`rocprofv3 --kernel-trace --stats -- python3 tools/small_epoch_profile.py`: which kernels a launch-bound epoch consists of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import data as D
from magnet_amd.layer.sparse_magnet import ChebNet
from magnet_amd.utils.hermitian import hermitian_decomp_sparse_device, cheb_poly_sparse
from magnet_amd.train import GraphedSharedEpochs, NodeTargets
dev = 'cuda'
g = D.dsbm_cyclic(seed=10); n = g.x.size(0)
L = hermitian_decomp_sparse_device(g.edge_index[0], g.edge_index[1], n, 0.1)
X = g.x.to(dev); y = g.y[None].to(dev)
tr = NodeTargets(y, g.train_mask[:, 0].to(dev)); va = NodeTargets(y, g.val_mask[:, 0].to(dev))
torch.manual_seed(0)
net = ChebNet(1, cheb_poly_sparse(L, 2), None, num_filter=16, K=2, label_dim=5, activation=True, layer=2).to(dev)
net.Chebs[0].cache_input_propagation = len(sys.argv) > 1 and sys.argv[1] == 'cached'
opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4, capturable=True)
ep = GraphedSharedEpochs.for_nodes(net, opt, X, X, y, tr, va)
for _ in range(20):
    ep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300):
    ep()
torch.cuda.synchronize()
print('epoch %.3f ms' % ((time.perf_counter() - t0) / 300 * 1e3))
