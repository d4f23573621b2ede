#!/usr/bin/env python3
"""Headline benchmark: MagNetConv (fused complex Chebyshev SpMM) edge rate on a synthetic
power-law digraph, with the HBM roofline of the dominant kernel and a CPU baseline.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 5, the one the metric's roofline target is quoted on; it fits one
288 GB GPU): N = 10M nodes, E = 200M edges (src = floor(N u1^2), dst = floor(N u2^2), Philox seed
43), 256 complex features, K = 3, q = 0.25.  One step = one MagNetConv forward: K fused complex
SpMM orders + (K+1) real weight GEMMs + bias, inputs resident in HBM.  value = nnz(L) * K / t.
With --gpus N > 1 the same graph is spread over the ranks (strong scaling), see magnet_amd/dist.py:
--strategy rows: every rank owns a contiguous block of rows balanced by nonzeros and each order
exchanges the freshly computed row blocks over RCCL; --strategy cols: every rank holds the whole
Laplacian and a slice of the feature columns, the recursion runs without communication and one
all-to-all per conv regroups the orders by rows for the weight GEMM; auto picks the layout that moves fewer bytes.
Whatever the headline layout, the row-partitioned conv is also timed in every arrangement of its halo exchange
(multi_gpu.row_variants) and the epoch leg runs in both layouts.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--nodes', type=int, default=10_000_000)
    p.add_argument('--edges', type=int, default=200_000_000)
    p.add_argument('--features', type=int, default=256)
    p.add_argument('--K', type=int, default=3)
    p.add_argument('--q', type=float, default=0.25)
    p.add_argument('--seed', type=int, default=43)
    p.add_argument('--strategy', choices=['auto', 'rows', 'cols'], default='auto',
                   help='multi-GPU decomposition (see magnet_amd/dist.py)')
    p.add_argument('--force-dist', action='store_true', help='take the multi-GPU code path even with one rank (rehearsal)')
    p.add_argument('--relabel', action='store_true',
                   help='apply a random vertex relabelling (seed 99) to the generated graph: the locality-free stress variant')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-epoch', action='store_true', help='skip the secondary epoch-time measurement')
    p.add_argument('--epoch-layout', choices=['both', 'cols', 'rows'], default='both',
                   help='multi-GPU epoch leg: cols = ShardedChebNet (every rank holds the whole CSR), rows = RowShardedChebNet '
                        '(every rank builds and holds its row block only, halo exchanges in forward and backward), both = one '
                        'after the other, each on record')
    p.add_argument('--epoch-nodes', type=int, default=2_000_000, help='graph of the secondary epoch measurement (config 4)')
    p.add_argument('--epoch-edges', type=int, default=40_000_000)
    p.add_argument('--cpu-sample-nodes', type=int, default=625_000, help='CPU baseline sample: 1/16 of the workload by default')
    p.add_argument('--cpu-sample-edges', type=int, default=12_500_000)
    return p.parse_args()


def algorithmic_bytes(nnz, n_rows, f, n_addends):
    """HBM bytes one SpMM order must move when no gathered row is reused: CSR streamed once
    (int32 column + complex64 value per nonzero, row_ptr), one gathered complex feature row per
    nonzero, one output row per node, one more row read per axpy addend."""
    row_bytes = 8 * f
    return nnz * 12 + 4 * (n_rows + 1) + nnz * row_bytes + n_rows * row_bytes * (1 + n_addends)


def compulsory_bytes(nnz, n_rows, f, n_addends):
    """Lower bound of the same order's HBM traffic if every feature row had to be read only once (perfect
    reuse of gathered rows): CSR once, X once, the addends once, the output once."""
    row_bytes = 8 * f
    return nnz * 12 + 4 * (n_rows + 1) + n_rows * row_bytes * (2 + n_addends)


def builder_bytes(e, n, nnz, passes):
    """Bytes the Laplacian builder streams for E edges (m = 2E directed records of 8 bytes, unweighted, int32 ids):
    `passes` radix passes of histogram read + scatter read + write (the first pass reads the edge lists, 8 bytes per
    record, and makes the records on the fly: no unsorted record is ever written), the binary searches for the group
    starts (their probes share cache lines; counted as the N + 1 words they write, twice), segments (keys read, one
    flag BIT written per record), the scan of the flag words (read twice, one 4-byte prefix written per 64 records),
    row_ptr (N + 1 lookups of a prefix and a flag word, N + 1 words written), emission (keys, flag words and prefixes
    read; column and value, 12 bytes per nonzero, written).  Runs longer than 64 records of one directed pair go through
    a side list (not counted: none on this workload).  d^-1/2 of an entry's column comes from the record count its key
    carries (unweighted whole-graph builds with whole column passes; a random 4-byte read per nonzero otherwise, not
    counted)."""
    return sum(builder_bytes_split(e, n, nnz, passes))


def builder_bytes_split(e, n, nnz, passes):
    """builder_bytes in two parts: bytes of the kernels that essentially only read (histograms, group starts, segments,
    flag scan, row_ptr) and bytes of those that read and write in equal measure (the scatters, the emission).  The
    first kind can approach the device's read-only streaming rate, the second its copy rate."""
    m = 2 * e
    histograms = passes * 8 * m
    scatters = passes * 16 * m
    starts = 8 * (n + 1)
    segments = 8 * m + m // 8
    flag_scan = 2 * (m // 8) + m // 16
    row_ptr = 20 * (n + 1)
    emission = 8 * m + m // 8 + m // 16 + 12 * nnz
    return histograms + starts + segments + flag_scan + row_ptr, scatters + emission


def file_sha256(path):
    import hashlib
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def stream_ceilings(dev, nbytes=8 << 30, iters=5):
    """Measured streaming ceilings of this device in GB/s: a device-to-device copy of nbytes (torch,
    read + write counted) and a read-only pass over the same buffer (mh_stream_read).  Reported
    beside the spec peak: the SpMM is read-dominated, so the read-only figure is the one it can approach."""
    from magnet_amd import _lib
    a = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    sink = torch.zeros(1, device=dev)
    read = lambda: _lib.ops.stream_read(a, sink)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    b.copy_(a); read()
    ev[0].record()
    for _ in range(iters):
        b.copy_(a)
    ev[1].record()
    for _ in range(iters):
        read()
    ev[2].record(); torch.cuda.synchronize()
    return {'copy': 2.0 * nbytes * iters / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9,
            'read': 1.0 * nbytes * iters / (ev[1].elapsed_time(ev[2]) * 1e-3) / 1e9}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args):
    """The reference formulation (uncoalesced COO, four real spmm per order) on the host cores,
    on a bounded sample of the same workload."""
    from oracle import magnet_oracle as O
    n, e, f, K = args.cpu_sample_nodes, args.cpu_sample_edges, args.features, args.K
    src, dst = O.powerlaw_edges(n, e, seed=args.seed)
    t0 = time.time()
    L = O.hermitian_decomp_sparse(src, dst, n, args.q)
    t_build = time.time() - t0
    Lr, Li = O.laplacian_lists([L])
    Lr, Li = Lr[0], Li[0]
    g = torch.Generator().manual_seed(1)
    Xr = torch.randn(n, f, generator=g); Xi = Xr.clone()
    W = torch.randn(K + 1, f, f, generator=g) / f ** 0.5
    b = torch.zeros(1, f)

    def conv(lr, li):
        # feature-side recursion with the reference's four-spmm bundle per order
        # (precomputed sparse T_k as in the reference is infeasible at this degree: fill-in ~ deg^k)
        zr, zi = [Xr], [Xi]
        for k in range(1, K + 1):
            pr, pi = zr[-1], zi[-1]
            nr = torch.spmm(lr, pr) - torch.spmm(li, pi)
            ni = torch.spmm(li, pr) + torch.spmm(lr, pi)
            if k >= 2:
                nr = 2.0 * nr - zr[-2]; ni = 2.0 * ni - zi[-2]
            zr.append(nr); zi.append(ni)
        real = sum(torch.matmul(z, W[k]) for k, z in enumerate(zr)) + b
        imag = sum(torch.matmul(z, W[k]) for k, z in enumerate(zi)) + b
        return real, imag

    t0 = time.time(); conv(Lr, Li); t_coo = time.time() - t0
    csr_r, csr_i = Lr.coalesce().to_sparse_csr(), Li.coalesce().to_sparse_csr()
    conv(csr_r, csr_i)
    t0 = time.time(); conv(csr_r, csr_i); t_csr = time.time() - t0
    return {
        'value': L.nnz * K / t_coo, 'unit': 'edges/s', 'cores': torch.get_num_threads(), 'kind': 'port',
        'extrapolated': True,
        'extrapolation': 'rate measured on a sample %.4g times smaller in nnz than the GPU workload (nnz %d of %d) and taken as '
                         'the rate of the full problem, i.e. time assumed linear in nnz * F' % (args.full_nnz / max(L.nnz, 1), L.nnz,
                                                                                                 args.full_nnz),
        'sample': 'same generator at N=%d, E=%d (nnz=%d), F=%d, K=%d: feature-side recursion with the reference\'s '
                  '4x torch.spmm (uncoalesced COO) per order + weight matmuls, torch CPU; %.1f s' % (n, e, L.nnz, f, K, t_coo),
        'strong_csr_value': L.nnz * K / t_csr,
        'strong_csr_note': 'same math with to_sparse_csr() operands (MKL), %.2f s' % t_csr,
        'builder_edges_per_s': e / t_build,
        'builder_note': 'numpy restatement of hermitian_decomp_sparse, 1 thread, %.2f s' % t_build,
        'host_cpu_count': os.cpu_count(),
        'cpu_model': _cpu_model(),
    }


def epoch_time(rank, world, dev, multi, n=2_000_000, e=40_000_000, layout='cols', budget_s=None):
    """Secondary metric: one training epoch (train forward + backward + Adam, then an evaluation
    forward: reference src/sparse_Magnet.py:143-176) of a 2-layer ChebNet, 128 -> 128 -> 128 filters,
    K = 2, q = 0.25, complex ReLU, 5 classes, on BASELINE.json config 4 (power-law digraph, 2M nodes /
    40M edges, 128 input features).  With several ranks: layout 'rows' = magnet_amd.dist.RowShardedChebNet over row
    blocks that every rank builds for itself (build_row_block: no rank holds the whole Laplacian), layout 'cols' =
    ShardedChebNet (every rank holds the whole CSR, layers alternate between row blocks and column slices).
    `budget_s`: seconds this measurement may take; the plain epoch is always timed, the variants after it (first layer's
    orders kept, shared forward) only while the time already spent plus their predicted cost stays inside the budget —
    decided from numbers reduced over the ranks, so every rank decides alike."""
    import torch.distributed as dist
    t_begin = time.time()
    from magnet_amd.laplacian import build_magnetic_laplacian, powerlaw_edges, MagneticLaplacian
    from magnet_amd.layer.sparse_magnet import ChebNet, ChebConv
    from magnet_amd.utils.hermitian import ChebBasis
    from magnet_amd.train import node_epoch, NodeTargets, SharedForwardEpochs
    f, K, ncls = 128, 2, 5
    gen = torch.Generator(device=dev).manual_seed(3)
    X = torch.randn(n, f, device=dev, generator=gen)
    y = torch.randint(0, ncls, (1, n), device=dev, generator=gen)
    train = torch.rand(n, device=dev, generator=gen) < 0.6
    extra = {}
    if multi and layout == 'rows':
        from magnet_amd import dist as D

        def chunks(chunk=20_000_000):
            for first in range(0, e, chunk):
                yield powerlaw_edges(n, min(chunk, e - first), 42, first_edge=first, device=dev)
        timing = {}
        lap, bounds = D.build_row_block(chunks, n, rank, world, q=0.25, timing=timing)
        part = D.RowPartition(lap.row_ptr, lap.col_idx, lap.val, n, bounds, rank, world,
                              lambda rp, ci, v, nc: MagneticLaplacian(rp, ci, v, nc))
        nnz = torch.tensor([lap.nnz], device=dev, dtype=torch.int64)
        D.all_reduce(nnz)
        nnz = int(nnz.item())
        extra = {'block_build_ms_rank0': timing.get('block_build_ms'), 'halo_rows_rank0': part.n_halo, 'own_rows_rank0': part.n_local}
    else:
        src, dst = powerlaw_edges(n, e, 42, device=dev)
        lap = build_magnetic_laplacian(src, dst, n, q=0.25)
        nnz = lap.nnz
        del src, dst
    torch.manual_seed(2)
    net = ChebNet(f, ChebBasis(lap, K), None, num_filter=f, K=K, label_dim=ncls, activation=True, layer=2).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=5e-4)
    if multi:
        from magnet_amd import dist as D
        sh = D.RowShardedChebNet(net, part) if layout == 'rows' else D.ShardedChebNet(net, lap, n, rank, world)
        r0, r1 = sh.r0, sh.r1
        xp = torch.stack([X[r0:r1], X[r0:r1]], dim=1).contiguous()
        yl = y[:, r0:r1].contiguous()
        # labels of the rank's training / validation rows with the others ignored: the loss reads all local rows
        tr, va = NodeTargets(yl, train[r0:r1]), NodeTargets(yl, ~train[r0:r1])
        n_tr, n_va = int(train.sum()), int((~train).sum())
        del X
        run = lambda: sh.epoch(opt, xp, yl, tr, va, n_tr, n_va)
    else:
        tr, va = NodeTargets(y, train), NodeTargets(y, ~train)
        run = lambda: node_epoch(net, opt, X, X, y, tr, va)

    def timed(iters=5, run=run):
        for _ in range(2):
            run()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            from magnet_amd.dist import all_reduce
            all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt / iters * 1e3

    skipped = []

    def affordable(n_timed, what):
        """Whether n_timed more calls of timed() (7 epochs each, at most as slow as the plain epoch) fit the budget;
        the same answer on every rank."""
        if budget_s is None:
            return True
        v = torch.tensor([time.time() - t_begin, -float(budget_s)], device=dev, dtype=torch.float64)
        if multi:
            from magnet_amd.dist import all_reduce
            all_reduce(v, op=dist.ReduceOp.MAX)          # longest time spent, smallest budget
        ok = v[0].item() + n_timed * 7 * ms / 1e3 <= -v[1].item()
        if not ok:
            skipped.append(what)
        return ok

    ChebConv.cache_input_propagation = False       # every forward recomputes T_k(L) X, as the reference does
    ms = timed()
    ms_cached = shared_ms = shared_cached_ms = None
    ChebConv.cache_input_propagation = True        # first layer's weight-independent T_k(L) X kept across forwards
    if multi:
        sh.cache_input_propagation = True
    if affordable(1, 'cached_input_propagation'):
        ms_cached = timed()
    if multi:
        if affordable(2, 'shared_forward'):
            sh.shared_forward = True               # the same over the ranks (ShardedChebNet.shared_forward)
            shared_cached_ms = timed()
            sh.cache_input_propagation = False
            sh._kept_pass = sh._input_cache = None
            shared_ms = timed()
            sh.shared_forward, sh._kept_pass = False, None
    elif affordable(2, 'shared_forward'):
        # the evaluation's pass through the convolutions kept for the next epoch's training step (same weights, same
        # tensor: magnet_amd.train.SharedForwardEpochs): one forward per epoch instead of two, identical results
        shared = SharedForwardEpochs(net, opt)
        run_shared = lambda: shared.node_epoch(X, X, y, tr, va)
        shared_cached_ms = timed(run=run_shared)
        ChebConv.cache_input_propagation = False
        shared.reset()
        shared_ms = timed(run=run_shared)
        shared.reset()
    ChebConv.cache_input_propagation = False       # back to the default
    kind = {'rows': ', RowShardedChebNet over per-rank built row blocks', 'cols': ', ShardedChebNet (whole CSR on every rank)'}
    out = {'ms': ms, 'ms_with_cached_input_propagation': ms_cached, 'ms_with_shared_forward': shared_ms,
           'ms_with_shared_forward_and_cached_input_propagation': shared_cached_ms,
           'shared_forward_note': 'the evaluation forward of epoch e and the training forward of epoch e+1 run under the same '
                                  'weights and differ only in the dropout after the convolutions: one pass through the '
                                  'convolutions serves both (train.SharedForwardEpochs; same losses, accuracies and '
                                  'parameters as the plain epoch, which is what `ms` times)' if shared_ms is not None else None,
           'n_gpus': world, 'nnz': nnz, 'layout': layout if multi else 'single',
           'config': 'config 4: power-law digraph N=%d E=%d, ChebNet 128->128->128, K=2, q=0.25, 2 layers, '
                     'train fwd+bwd+Adam + eval fwd' % (n, e) + (kind[layout] if multi else '')}
    if skipped:
        out['skipped_for_time'] = skipped
    out.update(extra)
    return out


def time_variant(conv, mode, K, nnz_total, dev, multi, steps, cap=0):
    """`steps` steps of the row-partitioned conv in one arrangement of its exchange: wall time per step (max over ranks),
    the compute stream's wait for each order's exchange (HIP events around the wait, max over ranks), the host time
    spent issuing one exchange."""
    import torch.distributed as dist
    from magnet_amd import dist as D
    part = conv.part
    part.set_mode(mode, int_grid_cap=cap)
    conv.step()
    part.wait_events, part.issue_seconds = [], []
    t = timed_steps(conv.step, steps, dev, multi)
    ev, issue = part.wait_events, part.issue_seconds
    part.wait_events = part.issue_seconds = None
    waits = [a.elapsed_time(b) for a, b in ev]
    # every rank takes part in the reduction whatever it recorded (a rank without halo records no wait)
    mine = [sum(waits[i::K]) / steps for i in range(K)] if (len(waits) == K * steps and K > 0) else [0.0] * K
    w = torch.tensor(mine + [float(len(waits))], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        D.all_reduce(w, op=dist.ReduceOp.MAX)
    per_order = w[:K].tolist() if w[K].item() > 0 else None
    res = {'ms_per_step': t / steps * 1e3, 'edges_per_s': nnz_total * K / (t / steps),
           'exposed_wait_ms_per_order': per_order,
           'exposed_wait_ms_per_step': None if per_order is None else sum(per_order),
           'host_issue_ms_per_exchange_rank0': (sum(issue) / len(issue) * 1e3) if issue else None}
    part._gathered.clear()
    return res


def extension_variant(conv, K, nnz_total, dev, multi, steps=3):
    """The same conv with every exchange posted by the library itself (mh_exchange_*: packing launches and the grouped
    RCCL send/receive from native code on the library's own communicator and stream), timed like the other
    arrangements, and its output compared with the torch.distributed arrangement's on the same input."""
    import torch.distributed as dist
    from magnet_amd import dist as D
    part = conv.part
    part.set_mode('overlap')
    want = conv.step().clone()
    res = time_variant(conv, part.NATIVE_MODE, K, nnz_total, dev, multi, steps)
    got = conv.step()
    diff = torch.tensor([float((got - want).abs().max())], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        D.all_reduce(diff, op=dist.ReduceOp.MAX)
    res['max_abs_difference_from_overlap'] = diff.item()
    res['note'] = 'exchange posted by the library (mh_exchange_begin / mh_exchange_end) instead of torch.distributed'
    part.set_mode('overlap')
    part.close()
    return res


def row_variants(conv, K, nnz_total, dev, multi, steps=3):
    """The row-partitioned conv timed in every arrangement of its halo exchange (magnet_amd.dist.RowPartition.set_mode),
    `steps` steps each, and with the transfers switched off.  Per arrangement: wall time per step (max over ranks) and
    the time the compute stream sat waiting for each order's exchange (HIP events around the wait, max over ranks) —
    what the overlap did not hide.  Also the host time spent issuing one exchange."""
    part = conv.part
    cap = 6 * torch.cuda.get_device_properties(dev).multi_processor_count
    out = {}
    for mode in part.MODES:
        out[mode] = time_variant(conv, mode, K, nnz_total, dev, multi, steps, cap)
    out['overlap_capped']['own_column_product_workgroups'] = cap
    out['all_gather']['rows_received_per_rank_per_order'] = getattr(part, 'max_rows', 0) * (part.world - 1)
    out['overlap']['rows_received_per_rank_per_order_rank0'] = part.n_halo
    part.set_mode('overlap')
    part.comm_enabled = False
    conv.step()
    t = timed_steps(conv.step, steps, dev, multi)
    part.comm_enabled = True
    out['compute_only_ms_per_step'] = t / steps * 1e3
    out['steps_each'] = steps
    out['how_to_read'] = ('overlap vs compute_only: cost of the exchange after overlap; overlap vs sequential: what the overlap '
                          'buys; overlap_capped vs overlap: whether the copy kernels were starved of compute units; all_gather vs '
                          'overlap: one RCCL collective against a batch of point-to-point transfers; exposed_wait: per order, the '
                          'part of the exchange that the own-column product did not cover; extension (added after the epochs, one process per '
                          'GPU over RCCL only): overlap with the exchange posted by the library instead of torch.distributed')
    return out


def edge_source(args, n, e, dev, chunk=50_000_000):
    """The benchmark graph as a re-iterable stream of (src, dst) chunks: the Philox generator is addressed by
    edge id, so every rank reproduces any chunk without holding the whole list."""
    perm = None
    if args.relabel:
        perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(99)).to(torch.int32)

    def chunks():
        from magnet_amd.laplacian import powerlaw_edges
        for first in range(0, e, chunk):
            src, dst = powerlaw_edges(n, min(chunk, e - first), args.seed, first_edge=first, device=dev)
            if perm is not None:
                src, dst = perm[src.long()], perm[dst.long()]
            yield src, dst
    return chunks


def note(rank, msg):
    """Progress on stderr (rank 0): a long run shows where it is; stdout stays reserved for the JSON line."""
    if rank == 0:
        sys.stderr.write('[bench %6.1fs] %s\n' % (time.time() - T_START, msg))
        sys.stderr.flush()


T_START = time.time()


def timed_steps(step, steps, dev, multi):
    """Wall time of `steps` calls bracketed by barrier + synchronize, maximum over the ranks."""
    import torch.distributed as dist
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        from magnet_amd.dist import all_reduce
        all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def main():
    args = parse_args()
    # stdout carries the one JSON line and nothing else: libraries that print to file descriptor 1
    # (RCCL writes a version banner there) are pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    # MH_DIST_BACKEND=gloo: rehearsal of the multi-rank code on fewer GPUs than ranks (ranks share devices, transfers are
    # staged through host memory by magnet_amd.dist.start_transfers); its timings say nothing about xGMI
    backend = os.environ.get('MH_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    multi = world > 1 or args.force_dist
    if multi:
        import datetime
        if world == 1:                                  # --force-dist started without a launcher: a one-rank group
            for k, v in (('RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533')):
                os.environ.setdefault(k, v)
        # a collective that cannot complete ends the run after 5 minutes instead of the 10-minute default
        dist.init_process_group(backend, timeout=datetime.timedelta(minutes=5), **({'device_id': dev} if backend == 'nccl' else {}))

    from magnet_amd import laplacian as lapmod
    from magnet_amd.laplacian import build_magnetic_laplacian
    from magnet_amd.layer.sparse_magnet import ChebConv
    from magnet_amd.utils.hermitian import ChebBasis

    n, e, f, K = args.nodes, args.edges, args.features, args.K
    chunks = edge_source(args, n, e, dev)
    torch.manual_seed(1)
    strategy = 'single'
    dist_info = None
    lap = lap_blk = None
    ev_b0, ev_b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    build_cold_ms = [None]

    def build_full():
        """The whole-graph build, twice: the first call also pays the allocation of its workspace and the loading of its
        kernels (recorded as the cold time); the second is the builder itself."""
        src, dst = (torch.cat(t) for t in zip(*chunks()))
        for attempt in range(2):
            full = None
            torch.cuda.synchronize()
            t0 = time.time()
            ev_b0.record()
            full = build_magnetic_laplacian(src, dst, n, q=args.q)
            ev_b1.record()
            torch.cuda.synchronize()
            if attempt == 0:
                build_cold_ms[0] = ev_b0.elapsed_time(ev_b1)
        return full, time.time() - t0, ev_b0.elapsed_time(ev_b1)

    if multi:
        from magnet_amd import dist as D
        # row layout set-up: every rank builds only its own row block, from the edges incident to its rows
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        blk_timing = {}
        lap_blk, bounds = D.build_row_block(chunks, n, rank, world, q=args.q, timing=blk_timing)
        torch.cuda.synchronize()
        note(rank, 'row block built (%d of %d rows)' % (lap_blk.n_rows, n))
        t_setup = time.time() - t0                  # edge stream generated twice + degree count + selection + block build
        t_build = None                              # no whole-graph build in the row layout
        build_device_ms = None
        counts = torch.tensor([lap_blk.nnz], device=dev, dtype=torch.int64)
        D.all_reduce(counts)
        nnz_total = int(counts.item())
        part_probe = D.RowPartition(lap_blk.row_ptr, lap_blk.col_idx, lap_blk.val, n, bounds, rank, world,
                                    lambda rp, ci, v, nc: lapmod.MagneticLaplacian(rp, ci, v, nc))
        worst = torch.tensor([part_probe.n_halo, part_probe.sent_rows(), lap_blk.nnz, part_probe.n_local], device=dev,
                             dtype=torch.int64)
        D.all_reduce(worst, op=dist.ReduceOp.MAX)
        halo_max, sent_max, nnz_max, rows_max = (int(v) for v in worst.tolist())
        rows_floats = D.row_partition_floats(halo_max, f, K)
        cols_floats = D.column_sharding_floats(n, f, K, world)
        strategy = args.strategy
        if strategy == 'auto':
            ok_cols = world > 1 and K >= 1 and f % (4 * world) == 0
            strategy = 'cols' if ok_cols and cols_floats < rows_floats else 'rows'
        del part_probe
        dist_info = {
            'backend': dist.get_backend(), 'ranks_initialised': dist.get_world_size(), 'strategy': strategy,
            'strategy_choice': 'floats received per rank per conv: row partition %d (largest halo %d rows x K=%d orders), column '
                               'sharding %d (one all-to-all of the K+1 orders); --strategy %s' % (
                                   rows_floats, halo_max, K, cols_floats, args.strategy),
            'row_layout': {'largest_block_rows': rows_max, 'largest_block_nnz': nnz_max, 'largest_halo_rows': halo_max,
                           'largest_sent_rows_per_exchange': sent_max,
                           'bytes_sent_per_rank_per_order': sent_max * 8 * f, 'exchanges_per_conv': K,
                           'per_rank_block_build_ms': blk_timing.get('block_build_ms'),
                           'per_rank_block_build_note': 'mh_lap_block_* of rank 0 + the d^-1/2 all-reduce, HIP events',
                           'edge_stream_and_block_build_ms': t_setup * 1e3,
                           'edge_stream_note': 'the whole set-up of rank 0: edge stream generated twice (degrees, then '
                                               'records), bincount, selection of the incident edges, block build'},
        }
        if strategy == 'cols':
            lap, t_build, build_device_ms = build_full()       # every rank holds the whole CSR in this layout
            layer = D.ColumnShardedConv(lap, f, f, K, rank, world, dev)
            dist_info['column_layout'] = {'bytes_sent_per_rank_per_conv': cols_floats * 4, 'all_to_alls_per_conv': K + 1}
        else:
            layer = D.RowPartitionedConv(lap_blk, bounds, f, f, K, rank, world, dev)
        torch.cuda.empty_cache()
        step = layer.step
        launches = layer.launches
    else:
        lap, t_build, build_device_ms = build_full()
        nnz_total = lap.nnz
        torch.cuda.empty_cache()
        conv = ChebConv(f, f, K, ChebBasis(lap, K), None).to(dev)
        # features resident in the layout a conv inside a network receives them in: order 0 of the
        # [N, 2, K+1, F] block whose other orders the conv fills (the layer before writes them there)
        block = torch.empty(n, 2, K + 1, f, device=dev)
        xp = block[:, :, 0, :]
        xp.normal_()
        launches = []

        def step():
            with torch.no_grad():
                return conv.forward_packed(xp, in_block=block)

    def observer(ev0, ev1, info):
        launches.append((ev0, ev1, info))

    note(rank, 'set-up done (nnz %d, strategy %s); warm-up' % (nnz_total, strategy))
    for _ in range(args.warmup):
        step()
    launches.clear()
    lapmod.SPMM_OBSERVER = observer
    elapsed = timed_steps(step, args.steps, dev, multi)
    lapmod.SPMM_OBSERVER = None
    note(rank, '%d timed steps: %.1f ms per step' % (args.steps, elapsed / args.steps * 1e3))

    # dominant kernel: the fused complex SpMM, timed by HIP events on its own stream
    k_ms = sum(a.elapsed_time(b) for a, b, _ in launches)
    k_bytes = sum(algorithmic_bytes(i['nnz'], i['n_rows'], i['F'], i['n_addends']) for _, _, i in launches)
    k_min = sum(compulsory_bytes(i['nnz'], i['n_rows'], i['F'], i['n_addends']) for _, _, i in launches)
    n_launches_total = len(launches)
    n_launch = max(n_launches_total, 1)
    achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, traffic_note = None, 'no counter file for this workload'
    tpath = os.path.join(ROOT, 'profiles', 'spmm_traffic.json')
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            if tj.get('nodes') == n and tj.get('edges') == e and tj.get('features') == f:
                if tj.get('kernel_source_sha256') == file_sha256(os.path.join(ROOT, 'magnet_amd', 'csrc', 'cspmm.hip')):
                    traffic, traffic_note = tj.get('hbm_bytes_per_launch'), tj.get('source')
                else:
                    traffic_note = 'profiles/spmm_traffic.json was collected on a different cspmm.hip: refused as stale'
        except Exception as ex:
            traffic_note = 'unreadable counter file: %r' % (ex,)

    # multi-GPU: the same steps with the transfers switched off = compute alone; the difference is what the
    # exchange costs after overlap
    if multi and dist_info is not None:
        target = layer.part if strategy == 'rows' else layer.shard
        target.comm_enabled = False
        step()
        compute_only = timed_steps(step, args.steps, dev, multi)
        target.comm_enabled = True
        dist_info['compute_only_ms_per_step'] = compute_only / args.steps * 1e3
        dist_info['exposed_comm_ms_per_step'] = (elapsed - compute_only) / args.steps * 1e3
        dist_info['peak_memory_gb_rank0'] = torch.cuda.max_memory_allocated(dev) / 1e9
        dist_info['unmeasured_note'] = ('figures of a run with ranks_initialised > 1 on backend nccl are the first measurements of '
                                        'this code on xGMI; gloo runs stage transfers through host memory and say nothing about it')
        if world == 1:
            dist_info['note'] = 'single-rank rehearsal of the multi-GPU code path: no bytes cross a link'

    ceiling = stream_ceilings(dev) if rank == 0 else None
    note(rank, 'headline complete; secondary measurements follow')

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        passes = (2 * max(1, (n - 1).bit_length()) + 7) // 8
        b_bytes = builder_bytes(e, n, nnz_total, passes)
        b_read, b_copy = builder_bytes_split(e, n, nnz_total, passes)
        b_floor_ms = (b_read / ceiling['read'] + b_copy / ceiling['copy']) / 1e6 if ceiling else None
        out = {
            'metric': 'MagNetConv edges/s', 'value': nnz_total * K / (elapsed / args.steps), 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'power-law digraph N=%d E=%d (seed %d), %d complex features, K=%d, q=%g; step = one '
                            'MagNetConv forward (K fused complex SpMM orders + one weight GEMM over the K+1 orders + bias)%s' % (
                                n, e, args.seed, f, K, args.q, '; vertices randomly relabelled (seed 99)' if args.relabel else ''),
                'nnz': nnz_total, 'parallelism': '%s%d' % (strategy, world) if multi else 'single',
            },
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'traffic_kind': 'L2 <-> fabric bytes per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE): reads served by the '
                                'Infinity Cache are inside this figure, so it bounds HBM bytes from above',
                'traffic_source': traffic_note,
                'kernel': 'cspmm_vec4 (fused complex CSR SpMM)', 'launches': n_launches_total,
                'avg_launch_ms': k_ms / n_launch, 'algorithmic_bytes_per_launch': k_bytes / n_launch,
                'b_min_bytes_per_launch': k_min / n_launch,
                'b_min_note': 'compulsory bytes if every gathered feature row were read from HBM once per order',
                'measured_stream_ceilings': ceiling,
                'frac_of_measured_read_ceiling': achieved / ceiling['read'] if ceiling else None,
            },
            'spmm_edges_per_s': nnz_total * K * args.steps / (k_ms * 1e-3) if k_ms > 0 and world == 1 else None,
            'input_edges_per_s': e * K / (elapsed / args.steps),
            'builder_edges_per_s': None if t_build is None else e / t_build,
            'builder_ms': None if t_build is None else t_build * 1e3,
            'builder_roofline': None if build_device_ms is None else {
                'bound': 'hbm', 'device_ms': build_device_ms, 'first_build_device_ms': build_cold_ms[0],
                'bytes_model': b_bytes, 'sort_passes': passes,
                'achieved': b_bytes / (build_device_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': b_bytes / (build_device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'floor_ms_at_measured_ceilings': b_floor_ms,
                'frac_of_floor_at_measured_ceilings': None if b_floor_ms is None else b_floor_ms / build_device_ms,
                'floor_note': 'read-mostly kernels at the read-only streaming rate measured in this run, read+write kernels at '
                              'its copy rate (roofline.measured_stream_ceilings)',
                'note': 'streamed bytes of all builder kernels (bench.py: builder_bytes) / HIP-event time of the whole build (second '
                        'build of the process; first_build_device_ms includes workspace allocation and kernel loading); '
                        'd^-1/2 of the columns comes from counts carried in the keys on this workload (no random reads)'},
            'multi_gpu': dist_info,
            'device': {'name': torch.cuda.get_device_properties(dev).name, 'uuid': str(getattr(torch.cuda.get_device_properties(dev), 'uuid', '')),
                       'hbm_gb': torch.cuda.get_device_properties(dev).total_memory / 1e9},
            'epoch': None,
        }

    def emit():
        sys.stdout.flush()
        if rank == 0:
            os.write(json_fd, (json.dumps(out) + '\n').encode())

    # The headline above is complete.  The secondary GPU measurements that follow must not be able to lose it:
    # if they have not finished after 6 minutes (a collective that never completes, a hung kernel), rank 0
    # prints the line without them, says why on stderr, and every rank ends with a non-zero status.
    stage = ['secondary']

    def give_up():
        if stage[0] == 'extension':
            if out is not None and isinstance(dist_info.get('row_variants'), dict):
                dist_info['row_variants']['extension'] = {'error': 'did not finish within its time limit'}
            sys.stderr.write('bench.py: the library-posted exchange did not finish within its time limit; line printed without it\n')
            sys.stderr.flush()
            if rank == 0 and not args.no_cpu_baseline:      # the rest of the line is owed regardless (this thread is free:
                args.full_nnz = nnz_total                   # a stuck GPU call holds no interpreter lock)
                try:
                    out['cpu_baseline'] = cpu_baseline(args)
                except Exception as ex:
                    out['cpu_baseline'] = {'error': repr(ex)[:300]}
            emit()
            os._exit(0)
        if out is not None:
            out['epoch'] = {'error': 'secondary measurements did not finish within 360 s'}
            if epoch_results:                           # what had finished by then stays on record
                out['epoch']['layouts'] = dict(epoch_results)
        sys.stderr.write('bench.py: secondary measurements did not finish within 360 s; headline printed, exiting with status 3\n')
        sys.stderr.flush()
        emit()
        os._exit(3)

    import threading
    rconv = None
    epoch_results = {}
    t_secondary = time.time()
    watchdog = threading.Timer(360.0, give_up)
    watchdog.daemon = True
    watchdog.start()

    # The row-partitioned conv (row blocks + one halo exchange per order, the layout BASELINE configs 4/5 name) in every
    # arrangement of its exchange, whatever layout the headline ran in: both layouts and all variants are on record.
    if multi and not args.no_epoch:
        try:
            if strategy == 'cols':
                del step
                layer = None
                torch.cuda.empty_cache()
                rconv = D.RowPartitionedConv(lap_blk, bounds, f, f, K, rank, world, dev)
            else:
                rconv = layer
            note(rank, 'row-partition variants')
            variants = row_variants(rconv, K, nnz_total, dev, multi)
            if out is not None:
                dist_info['row_variants'] = variants
                ov = variants['overlap']
                out['row_partition'] = {'ms_per_step': ov['ms_per_step'], 'edges_per_s': ov['edges_per_s'],
                                        'compute_only_ms_per_step': variants['compute_only_ms_per_step'],
                                        'exposed_comm_ms_per_step': ov['ms_per_step'] - variants['compute_only_ms_per_step'],
                                        'note': 'same conv, rows of L and X partitioned, compact [own | halo] buffers, K halo '
                                                'exchanges per conv (the overlap variant of multi_gpu.row_variants)'}
        except Exception as ex:
            rconv = None
            if out is not None:
                dist_info['row_variants'] = {'error': repr(ex)[:300]}
        step = None

    if not args.no_epoch:
        step = launches = None
        layer = conv = xp = block = lap = lap_blk = None
        torch.cuda.empty_cache()
        layouts = (['rows', 'cols'] if args.epoch_layout == 'both' else [args.epoch_layout]) if multi else ['single']
        results = epoch_results
        for i, lay in enumerate(layouts):
            note(rank, 'epoch measurement (config 4, layout %s)' % lay)
            # an equal share of what is left of the 360 s (minus a reserve): a slow transport (the gloo rehearsal stages
            # every transfer through the host) then drops the optional variants instead of tripping the watchdog
            budget = max(300.0 - (time.time() - t_secondary), 0.0) / (len(layouts) - i)
            try:
                results[lay] = epoch_time(rank, world, dev, multi, args.epoch_nodes, args.epoch_edges, lay, budget_s=budget)
            except Exception as ex:                         # the headline metric must still be reported
                results[lay] = {'error': repr(ex)[:300]}
            torch.cuda.empty_cache()
        good = [r for r in results.values() if 'ms' in r]
        epoch = dict(min(good, key=lambda r: r['ms'])) if good else dict(next(iter(results.values())))
        if multi:
            epoch['layouts'] = results
        if out is not None:
            out['epoch'] = epoch
    # Last, because it is the one leg no machine of the build could run with RCCL between several GPUs: the row-partitioned
    # conv with its exchanges posted by the library.  Everything above is already in `out`; a failure or a hang here
    # costs this leg only (its own time limit; the line is printed with what is there and the status stays 0).
    watchdog.cancel()
    if multi and rconv is not None and (backend == 'nccl' or os.environ.get('MAGNET_RCCL_LIBRARY')):
        stage[0] = 'extension'
        watchdog = threading.Timer(float(os.environ.get('MH_EXTENSION_LIMIT_S', '150')), give_up)
        watchdog.daemon = True
        watchdog.start()
        note(rank, 'row partition with the exchange posted by the library')
        try:
            res = extension_variant(rconv, K, nnz_total, dev, multi)
        except Exception as ex:
            res = {'error': repr(ex)[:300]}
        if out is not None and isinstance(dist_info.get('row_variants'), dict):
            dist_info['row_variants']['extension'] = res
        watchdog.cancel()
    rconv = None
    if multi:
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline:          # at every N: rank 0's host cores, after the ranks have parted
        args.full_nnz = nnz_total
        note(rank, 'CPU baseline on a sample (host cores)')
        try:
            out['cpu_baseline'] = cpu_baseline(args)
        except Exception as ex:
            out['cpu_baseline'] = {'error': repr(ex)[:300]}
    emit()
    os.close(json_fd)


if __name__ == '__main__':
    main()
