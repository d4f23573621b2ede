"""The multi-rank code with REAL processes and the REAL HIP kernels on one GPU: world_size 2 and 3 ranks share
cuda:0 and talk over gloo (RCCL needs one GPU per rank; magnet_amd.dist stages device tensors through host memory
on that backend).  What runs is what bench.py --gpus N runs on an 8-GPU node, transport aside: every rank builds
only its row block (mh_lap_block_*), plans its halo (mh_halo_plan_*), exchanges need lists, propagates with the
packed halo exchange — and the column-sharded conv — against the single-GPU result."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

N, E, F, TERMS, SEED = 30_000, 600_000, 32, 4, 12


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _chunks(dev, weighted):
    from magnet_amd.laplacian import powerlaw_edges

    def gen():
        for first in range(0, E, 250_000):                      # several chunks, as the bench streams them
            src, dst = powerlaw_edges(N, min(250_000, E - first), SEED, first_edge=first, device=dev)
            if weighted:
                w = ((src.long() * 7 + dst.long() * 3 + first) % 5 + 1).float() * 0.5
                yield src, dst, w
            else:
                yield src, dst
    return gen


def _worker(rank, world, port, out_dir, weighted):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from magnet_amd import dist as D
        from magnet_amd.laplacian import MagneticLaplacian
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        blk, bounds = D.build_row_block(_chunks(dev, weighted), N, rank, world, q=0.2)
        make = lambda rp, ci, v, nc: MagneticLaplacian(rp, ci, v, nc)
        part = D.RowPartition(blk.row_ptr, blk.col_idx, blk.val, N, bounds, rank, world, make)
        x = torch.randn(N, 2, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        xb = part.new_buffer(F, dev); xb.fill_(float('nan'))
        part.block(xb).copy_(x[part.r0:part.r1])
        zs = part.cheb_terms(xb, TERMS)
        # column-sharded conv on the same graph (every rank holds the whole CSR in that layout)
        full, _ = D.build_row_block(_chunks(dev, weighted), N, 0, 1, q=0.2)
        g = torch.Generator().manual_seed(5)
        W = torch.randn(TERMS, F, 8, generator=g).to(dev); b = torch.randn(1, 8, generator=g).to(dev)
        sh = D.ColumnSharded(full, N, F, rank, world)
        out = sh.conv(x[:, :, sh.c0:sh.c1].contiguous(), W, b)
        torch.cuda.synchronize()
        torch.save({'bounds': bounds, 'row_ptr': blk.row_ptr.cpu(), 'col_idx': blk.col_idx.cpu(), 'val': blk.val.cpu(),
                    'blocks': [part.block(z).cpu() for z in zs], 'n_halo': part.n_halo, 'sent': part.sent_rows(),
                    'cols_out': out.cpu(), 'rb': sh.rb}, os.path.join(out_dir, 'r%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,weighted', [(2, False), (3, True)])
def test_ranks_sharing_one_gpu_match_the_single_gpu_result(world, weighted, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), weighted), nprocs=world, join=True)
    from magnet_amd.laplacian import build_magnetic_laplacian
    dev = torch.device('cuda', 0)
    parts = [torch.cat(t) for t in zip(*_chunks(dev, weighted)())]
    L = build_magnetic_laplacian(parts[0], parts[1], N, q=0.2, edge_weight=parts[2] if weighted else None)
    outs = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    bounds = outs[0]['bounds']
    assert bounds[0] == 0 and bounds[-1] == N and all(o['bounds'] == bounds for o in outs)
    # the per-rank blocks, stacked, are the single-GPU CSR bit for bit
    rp, ci, v = L.row_ptr.cpu(), L.col_idx.cpu(), L.val.cpu()
    for r, o in enumerate(outs):
        a, z = int(rp[bounds[r]]), int(rp[bounds[r + 1]])
        assert torch.equal(o['row_ptr'], rp[bounds[r]:bounds[r + 1] + 1] - rp[bounds[r]])
        assert torch.equal(o['col_idx'], ci[a:z]) and torch.equal(o['val'], v[a:z])
    assert sum(o['sent'] for o in outs) == sum(o['n_halo'] for o in outs) > 0
    # propagation
    x = torch.randn(N, 2, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    ref = [x, L.spmm(x)]
    for k in range(2, TERMS):
        ref.append(L.spmm(ref[k - 1], alpha=2.0, z1=ref[k - 2], beta1=-1.0))
    for k in range(TERMS):
        got = torch.cat([o['blocks'][k] for o in outs]).to(dev)
        assert float((got - ref[k]).abs().max()) <= 1e-5 * float(ref[k].abs().max()), 'order %d' % k
    # column-sharded conv
    g = torch.Generator().manual_seed(5)
    W = torch.randn(TERMS, F, 8, generator=g).to(dev); b = torch.randn(1, 8, generator=g).to(dev)
    want = sum(torch.matmul(ref[k], W[k]) for k in range(TERMS)) + b.view(1, 1, -1)
    got = torch.cat([o['cols_out'] for o in outs]).to(dev)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


MOCK_RCCL = os.path.join(ROOT, 'tests', 'abi', 'libmock_rccl.so')


def _worker_native(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['MAGNET_RCCL_LIBRARY'] = MOCK_RCCL            # RCCL refuses ranks that share a device; see tests/abi/mock_rccl.cpp
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from magnet_amd import dist as D
        from magnet_amd.laplacian import MagneticLaplacian, build_magnetic_laplacian, powerlaw_edges
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        make = lambda rp, ci, v, nc: MagneticLaplacian(rp, ci, v, nc)
        report = {}
        for name in ('sparse', 'dense'):
            if name == 'sparse':        # peers take subsets of the block: packed sends
                n = N
                blk, bounds = D.build_row_block(_chunks(dev, False), n, rank, world, q=0.2)
                part = D.RowPartition(blk.row_ptr, blk.col_idx, blk.val, n, bounds, rank, world, make)
            else:                       # every rank references every remote row: the block itself is sent, in place
                n = 96
                src, dst = powerlaw_edges(n, 6000, 4, device=dev)
                src = (src + dst * 7) % n
                L = build_magnetic_laplacian(src, dst, n, q=0.2)
                part = D.RowPartition.from_global(L.row_ptr, L.col_idx, L.val, n, rank, world, make,
                                                  bounds=[n * r // world for r in range(world + 1)])
            x = torch.randn(n, 2, F, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

            def run():
                xb = part.new_buffer(F, dev); xb.fill_(float('nan'))
                part.block(xb).copy_(x[part.r0:part.r1])
                zs = [part.block(z).clone() for z in part.cheb_terms(xb, TERMS)]          # contiguous operands
                block = torch.full((part.n_local, 2, TERMS, F), float('nan'), device=dev)
                block[:, :, 0] = x[part.r0:part.r1]
                part.cheb_block(block)                                                     # strided operands: one order of a block
                back = part.clenshaw([block[:, :, k] for k in range(TERMS)])
                torch.cuda.synchronize()
                return zs, block.clone(), back.clone()

            part.set_mode('overlap')
            want = run()
            part.set_mode('extension')
            got = run()
            got2 = run()                                                                   # the object is reused call after call
            same = all(torch.equal(a, b) for a, b in zip(want[0], got[0])) and torch.equal(want[1], got[1]) \
                and torch.equal(want[2], got[2]) and torch.equal(want[1], got2[1]) and torch.equal(want[2], got2[2])
            # the object's all-gather: every rank's vector, in rank order, on every rank
            from magnet_amd._lib import ops
            mine = torch.full((1000,), float(rank + 1), device=dev) + torch.arange(1000, device=dev) * 1e-3
            everyone = torch.full((world * 1000,), float('nan'), device=dev)
            ops.exchange_all_gather(part._exchange, mine, everyone, world)
            ops.exchange_end(part._exchange, everyone)
            torch.cuda.synchronize()
            expect = torch.cat([torch.full((1000,), float(r + 1), device=dev) + torch.arange(1000, device=dev) * 1e-3 for r in range(world)])
            same = same and torch.equal(everyone, expect)
            in_place = sum(int(g.numel() == part.n_local) for p, g in enumerate(part.give) if p != rank and g is not None)
            report[name] = {'same': bool(same), 'finite': bool(torch.isfinite(got[1]).all()), 'in_place_peers': in_place,
                            'n_halo': part.n_halo}
            part.close()
            assert part.mode == 'overlap' and part._exchange is None
        torch.save(report, os.path.join(out_dir, 'n%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_exchange_posted_by_the_library_between_ranks_sharing_one_gpu(world, tmp_path):
    """mh_exchange_* (packing + grouped send/receive posted from native code, on the library's own stream) with 2 and 3
    real processes: forward recursion on contiguous and on strided operands and the Clenshaw backward give bit for bit
    what the torch.distributed path gives.  The transport under the library is the shared-memory stand-in for RCCL
    (one GPU here); the real RCCL is exercised by the single-rank test below."""
    assert os.path.exists(MOCK_RCCL), 'tests/abi/libmock_rccl.so is not built (make -C magnet_amd/csrc)'
    mp.spawn(_worker_native, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        rep = torch.load(os.path.join(str(tmp_path), 'n%d.pt' % r))
        assert rep['sparse']['same'] and rep['sparse']['finite'] and rep['sparse']['n_halo'] > 0, rep
        assert rep['dense']['same'] and rep['dense']['finite'] and rep['dense']['in_place_peers'] == world - 1, rep


def test_exchange_posted_by_the_library_over_the_real_rccl_single_rank():
    """The same entry points over the real RCCL with a communicator of one rank that serves itself: rows gathered from
    a strided block land in the halo buffer, a whole contiguous block is sent in place, an all-gather of one rank copies
    — the loading of the library, the communicator, the grouped calls and the stream ordering are the production ones."""
    script = r"""
import torch
from magnet_amd._lib import ops
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
g = torch.Generator(device=dev).manual_seed(1)
n, f, terms = 5000, 48, 3
block = torch.randn(n, 2, terms, f, device=dev, generator=g)
rows = torch.randperm(n, device=dev, generator=g)[:1234].sort().values.to(torch.int32)
ident = ops.exchange_new_id()
h = ops.exchange_create(ident, 0, 1, [rows], [0, rows.numel()], n, 0)
halo = torch.full((rows.numel(), 2, f), float('nan'), device=dev)
for k in (1, 2, 1):
    own = block[:, :, k]
    ops.exchange_begin(h, own, halo)
    busy = torch.mm(torch.randn(512, 512, device=dev), torch.randn(512, 512, device=dev))     # overlaps the transfer
    ops.exchange_end(h, halo)
    assert torch.equal(halo, own[rows.long()]), 'packed rows, order %d' % k
ops.exchange_destroy(h)
everything = torch.arange(n, device=dev, dtype=torch.int32)
h = ops.exchange_create(ops.exchange_new_id(), 0, 1, [everything], [0, n], n, 0)
own = torch.randn(n, 2, f, device=dev, generator=g)
halo = torch.empty(n, 2, f, device=dev)
ops.exchange_begin(h, own, halo)
ops.exchange_end(h, halo)
assert torch.equal(halo, own), 'block sent in place'
out = torch.empty(n * 2 * f, device=dev)
ops.exchange_all_gather(h, own.view(-1), out, 1)
ops.exchange_end(h, out)
assert torch.equal(out, own.view(-1)), 'all-gather of one rank'
ops.exchange_destroy(h)
torch.cuda.synchronize()
print('native exchange ok')
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MAGNET_RCCL_LIBRARY', None)
    res = subprocess.run([sys.executable, '-c', script], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'native exchange ok' in res.stdout, (res.stdout[-500:], res.stderr[-2000:])


def _train_setup(dev):
    g = torch.Generator().manual_seed(8)
    cin, ncls = 6, 4
    X = torch.randn(N, cin, generator=g).to(dev)
    y = torch.randint(0, ncls, (1, N), generator=g).to(dev)
    train = (torch.rand(N, generator=g) < 0.5).to(dev)
    return cin, ncls, X, y, train


def _worker_train(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from magnet_amd import dist as D
        from magnet_amd.laplacian import MagneticLaplacian
        from magnet_amd.layer.sparse_magnet import ChebNet
        from magnet_amd.utils.hermitian import ChebBasis
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        blk, bounds = D.build_row_block(_chunks(dev, False), N, rank, world, q=0.2)
        part = D.RowPartition(blk.row_ptr, blk.col_idx, blk.val, N, bounds, rank, world,
                              lambda rp, ci, v, nc: MagneticLaplacian(rp, ci, v, nc))
        cin, ncls, X, y, train = _train_setup(dev)
        torch.manual_seed(0)
        net = ChebNet(cin, ChebBasis(blk, 2), None, num_filter=8, K=2, label_dim=ncls, activation=True, layer=2).to(dev)
        sh = D.RowShardedChebNet(net, part)
        r0, r1 = sh.r0, sh.r1
        xp = torch.stack([X[r0:r1], X[r0:r1]], dim=1).contiguous()
        opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
        tr = train[r0:r1].nonzero().squeeze(1); va = (~train[r0:r1]).nonzero().squeeze(1)
        losses = []
        for _ in range(2):
            tl, vl = sh.epoch(opt, xp, y[:, r0:r1].contiguous(), tr, va, int(train.sum()), int((~train).sum()))
            losses.append((tl.item(), vl.item()))
        torch.save({'losses': losses, 'state': {k: v.cpu() for k, v in net.state_dict().items()}}, os.path.join(out_dir, 't%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_row_partitioned_training_on_ranks_sharing_one_gpu(tmp_path):
    """Two Adam epochs of a ChebNet over the ROW partition (every rank: its own row block, built by itself, compact
    [own | halo] features, halo exchanges in forward and backward) by three processes on this GPU, against the same
    epochs of the single-GPU network."""
    world = 3
    mp.spawn(_worker_train, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from magnet_amd.laplacian import build_magnetic_laplacian
    from magnet_amd.layer.sparse_magnet import ChebNet
    from magnet_amd.utils.hermitian import ChebBasis
    from magnet_amd.train import node_epoch
    dev = torch.device('cuda', 0)
    parts = [torch.cat(t) for t in zip(*_chunks(dev, False)())]
    L = build_magnetic_laplacian(parts[0], parts[1], N, q=0.2)
    cin, ncls, X, y, train = _train_setup(dev)
    torch.manual_seed(0)
    net = ChebNet(cin, ChebBasis(L, 2), None, num_filter=8, K=2, label_dim=ncls, activation=True, layer=2).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
    ref = []
    for _ in range(2):
        tl, _, vl, _ = node_epoch(net, opt, X, X, y, train.nonzero().squeeze(1), (~train).nonzero().squeeze(1))
        ref.append((tl.item(), vl.item()))
    for r in range(world):
        o = torch.load(os.path.join(str(tmp_path), 't%d.pt' % r))
        for (a, b), (c, d) in zip(o['losses'], ref):
            assert abs(a - c) <= 1e-5 * abs(c) and abs(b - d) <= 1e-5 * abs(d), (o['losses'], ref)
        for k, v in net.state_dict().items():
            assert float((o['state'][k].to(dev) - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-7, k


@pytest.mark.parametrize('strategy', ['rows', 'cols'])
def test_bench_two_rank_rehearsal_on_one_gpu(strategy):
    """bench.py launched as the driver launches it for --gpus 2 (torch.distributed.run, one process per rank), both
    ranks on this GPU over gloo: the whole multi-rank control flow (per-rank build, strategy record, timed steps,
    compute-only steps, the row-partition variants, the epoch in both layouts, the CPU baseline on rank 0) ends with
    one JSON line on rank 0 that carries every field the first run on several GPUs is meant to decide from."""
    # the leg whose exchanges the library posts itself runs over the shared-memory stand-in for RCCL (one GPU here)
    env = dict(os.environ, MH_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MAGNET_RCCL_LIBRARY=MOCK_RCCL,
               MOCK_RCCL_SLOT_BYTES=str(128 << 20))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--nodes', '200000',
           '--edges', '4000000', '--features', '64', '--K', '3', '--strategy', strategy,
           '--cpu-sample-nodes', '20000', '--cpu-sample-edges', '200000']
    cmd += ['--epoch-nodes', '50000', '--epoch-edges', '1000000']
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == strategy + '2' and d['value'] > 0
    m = d['multi_gpu']
    assert m['ranks_initialised'] == 2 and m['strategy'] == strategy and m['backend'] == 'gloo'
    rl = m['row_layout']
    assert rl['largest_halo_rows'] > 0 and rl['bytes_sent_per_rank_per_order'] > 0
    assert 0 < rl['per_rank_block_build_ms'] < rl['edge_stream_and_block_build_ms']
    assert m['compute_only_ms_per_step'] > 0 and np.isfinite(m['exposed_comm_ms_per_step'])
    assert 0 < rl['largest_block_nnz'] < d['config']['nnz']
    # the whole-graph builder figures exist only where a whole-graph build ran
    assert (d['builder_edges_per_s'] is None) == (strategy == 'rows')
    # every arrangement of the row partition's exchange, with the per-order wait of the compute stream
    v = m['row_variants']
    for mode in ('overlap', 'overlap_capped', 'sequential', 'all_gather'):
        assert v[mode]['ms_per_step'] > 0 and v[mode]['edges_per_s'] > 0, mode
        assert len(v[mode]['exposed_wait_ms_per_order']) == 3 and all(w >= 0 for w in v[mode]['exposed_wait_ms_per_order']), mode
        assert v[mode]['host_issue_ms_per_exchange_rank0'] > 0
    assert v['overlap_capped']['own_column_product_workgroups'] > 0 and v['compute_only_ms_per_step'] > 0
    assert v['all_gather']['rows_received_per_rank_per_order'] >= rl['largest_halo_rows']
    assert d['row_partition']['ms_per_step'] == v['overlap']['ms_per_step']
    x = v['extension']
    assert x['ms_per_step'] > 0 and x['max_abs_difference_from_overlap'] == 0.0 and len(x['exposed_wait_ms_per_order']) == 3, x
    # the epoch in both layouts, the row one over blocks the ranks built for themselves
    ep = d['epoch']
    assert ep['ms'] > 0 and ep['n_gpus'] == 2 and set(ep['layouts']) == {'rows', 'cols'}
    for lay in ('rows', 'cols'):
        e = ep['layouts'][lay]
        assert e['ms'] > 0 and e['layout'] == lay, e
        # the variants after the plain epoch are timed while the leg's time budget lasts (a slow box may drop them, and says so)
        for key, what in (('ms_with_cached_input_propagation', 'cached_input_propagation'), ('ms_with_shared_forward', 'shared_forward'),
                          ('ms_with_shared_forward_and_cached_input_propagation', 'shared_forward')):
            assert (e[key] is not None and e[key] > 0) or what in e.get('skipped_for_time', []), (key, e)
    assert 'RowShardedChebNet' in ep['layouts']['rows']['config'] and ep['layouts']['rows']['block_build_ms_rank0'] > 0
    assert ep['layouts']['rows']['nnz'] == ep['layouts']['cols']['nnz']
    assert ep['ms'] == min(ep['layouts'][l]['ms'] for l in ep['layouts'])
    # the CPU baseline is reported at every N
    assert d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['kind'] == 'port'


def test_bench_line_survives_a_stuck_library_exchange():
    """The leg whose exchanges the library posts runs last and under its own time limit: with a communication library
    that never completes (the stand-in, told to stall) rank 0 still prints the whole line — every other field, the CPU
    baseline included — records the leg as unfinished, and the run ends with status 0."""
    env = dict(os.environ, MH_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MAGNET_RCCL_LIBRARY=MOCK_RCCL,
               MOCK_RCCL_STALL='1', MH_EXTENSION_LIMIT_S='8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--nodes', '100000',
           '--edges', '2000000', '--features', '32', '--K', '2', '--strategy', 'rows',
           '--cpu-sample-nodes', '20000', '--cpu-sample-edges', '200000', '--epoch-nodes', '30000', '--epoch-edges', '600000']
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-500:]
    d = json.loads(lines[0])
    v = d['multi_gpu']['row_variants']
    assert 'error' in v['extension'] and v['overlap']['ms_per_step'] > 0
    assert d['epoch']['ms'] > 0 and d['cpu_baseline']['value'] > 0 and d['value'] > 0


@pytest.mark.parametrize('strategy', ['rows', 'auto'])
def test_bench_eight_rank_rehearsal_on_one_gpu(strategy):
    """bench.py --gpus 8 launched as the driver launches it, all eight ranks on this GPU (gloo, host-staged transfers, the
    library-posted exchange over the shared-memory stand-in): the eight-rank bookkeeping — bounds, column slices of 64 / 8
    features, seven peers per exchange, both epoch layouts — runs to the end and fills every field."""
    env = dict(os.environ, MH_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MAGNET_RCCL_LIBRARY=MOCK_RCCL,
               MOCK_RCCL_SLOT_BYTES=str(32 << 20), OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1', '--nodes', '200000',
           '--edges', '4000000', '--features', '64', '--K', '3', '--strategy', strategy,
           '--cpu-sample-nodes', '20000', '--cpu-sample-edges', '200000', '--epoch-nodes', '50000', '--epoch-edges', '1000000']
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-500:]
    d = json.loads(lines[0])
    m = d['multi_gpu']
    assert d['n_gpus'] == 8 and m['ranks_initialised'] == 8 and d['value'] > 0
    assert d['config']['parallelism'] == ('rows8' if strategy == 'rows' else m['strategy'] + '8')
    v = m['row_variants']
    for mode in ('overlap', 'overlap_capped', 'sequential', 'all_gather', 'extension'):
        assert v[mode]['ms_per_step'] > 0 and len(v[mode]['exposed_wait_ms_per_order']) == 3, (mode, v[mode])
    assert v['extension']['max_abs_difference_from_overlap'] == 0.0
    assert set(d['epoch']['layouts']) == {'rows', 'cols'} and all(e['ms'] > 0 for e in d['epoch']['layouts'].values())
    assert d['cpu_baseline']['value'] > 0
