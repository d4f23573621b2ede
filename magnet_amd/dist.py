"""Propagation over the GPUs of one node (one process per GPU, torch.distributed: backend "nccl" =
RCCL over xGMI on MI355X, "gloo" for CPU rehearsal of the exchange logic).

The reference is single-device (src/sparse_Magnet.py:23-24); this module is the graph-scale
extension of its propagation (src/layer/sparse_magnet.py:50-72).  Two decompositions:

* RowPartition  — rows of the Laplacian and of the features are cut into blocks; every order
  exchanges the halo rows (on a graph without locality that is nearly every row).  Memory and
  set-up scale with 1/world; the natural choice when a partition keeps most neighbours local.
* ColumnSharded — every rank holds the whole Laplacian and a slice of the feature columns; the
  recursion needs no communication and one all-to-all per layer turns column slices into row
  blocks for the weight GEMM.  K*world/(K+1) times fewer bytes on a graph without locality.

Row partition
-------------
* Rows of the Laplacian (and of every feature matrix) are cut into `world` contiguous blocks
  balanced by work (nonzeros + rows).  Rank p computes rows [b_p, b_{p+1}) of each order and holds
  nothing else of the graph: it BUILDS only its own row block, from the edges incident to its rows
  (build_row_block -> mh_lap_block_*; the ranks exchange one vector of d^-1/2).
* Feature buffers are COMPACT, [n_local + n_halo, 2, F]: the rank's rows, then its halo (the remote rows
  its block references, ascending by global id and therefore grouped by owner).  The block's columns are
  renumbered to that buffer by the halo plan (mh_halo_plan_*).  After computing its rows of Z_k a rank
  sends every peer the rows that peer references (packed by a row-gather kernel; in place when the peer
  references the whole block) and receives its own halo rows straight into their slice of the buffer.
  The transfers are point-to-point sends/receives issued as one batch: on the fully connected xGMI mesh
  every pair of GPUs has its own link, so the batch is meant to serve all 7 peers concurrently (a ring would push
  7x the bytes through one link).  Unmeasured on xGMI so far: the code has run on one GPU only (gloo staging and
  one-rank RCCL groups); bench.py --gpus N records the variants of RowPartition.set_mode side by side.
* To overlap the exchange with compute the block is split by COLUMN into L_int (own columns: operands
  already local) and L_ext (halo columns).  Order k runs  Z_k[own] = a L_int Z_{k-1} + b Z_{k-2}[own]
  while the exchange of Z_{k-1} is still in flight, waits for it, then adds  a L_ext Z_{k-1}.  That covers
  more than the interior-row split (rows without any remote column): every nonzero with a local column
  overlaps, whatever row it sits in.
* L_ext's columns count from the start of the halo, so own rows and halo rows need not share a tensor: a conv keeps
  its own rows of all K+1 orders side by side in one [n_local, 2, K+1, F] block (cheb_block), which one GEMM with the
  stacked weights reads, and every order's halo in a contiguous [n_halo, 2, F] tensor of its own.
* The weight GEMM and the bias act on rows independently, so the output block needs no exchange;
  a conv of K orders performs K exchanges (its input's halo and K-1 intermediate orders).  Weight
  gradients are all-reduced (tiny).
"""
import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------
# partition plan (plain tensor code: runs on CPU tensors for the gloo tests and on the GPU)
# ---------------------------------------------------------------------------------------------
def partition_bounds(row_ptr, world, row_cost=2):
    """world+1 row boundaries; blocks balanced by nnz + row_cost * rows (a row costs about two
    feature-row accesses besides its gathers: the output store and the axpy operand)."""
    n = row_ptr.numel() - 1
    cost = row_ptr.to(torch.int64) + row_cost * torch.arange(n + 1, device=row_ptr.device, dtype=torch.int64)
    total = int(cost[-1])
    targets = torch.tensor([total * p // world for p in range(1, world)], device=row_ptr.device, dtype=torch.int64)
    inner = torch.searchsorted(cost, targets).clamp_(0, n).tolist() if world > 1 else []
    bounds = [0] + [int(b) for b in inner] + [n]
    for i in range(1, len(bounds)):                      # keep it monotone for degenerate inputs
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def slice_rows(row_ptr, col_idx, val, r0, r1):
    """CSR of rows [r0, r1) (column ids unchanged)."""
    a, b = int(row_ptr[r0]), int(row_ptr[r1])
    rp = (row_ptr[r0:r1 + 1] - row_ptr[r0]).to(torch.int32)
    return rp, col_idx[a:b], val[a:b]


def split_columns(row_ptr, col_idx, val, c0, c1):
    """Split a CSR into (entries with c0 <= col < c1, the rest); both keep all rows."""
    n_rows = row_ptr.numel() - 1
    inside = (col_idx >= c0) & (col_idx < c1)
    counts = (row_ptr[1:] - row_ptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n_rows, device=row_ptr.device), counts)
    out = []
    for m in (inside, ~inside):
        cnt = torch.bincount(rows[m], minlength=n_rows)
        rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=row_ptr.device)
        rp[1:] = torch.cumsum(cnt, 0)
        out.append((rp.to(torch.int32), col_idx[m].contiguous(), val[m].contiguous()))
    return out[0], out[1]


def wait_all(works):
    for w in works:
        w.wait()


class _StagedTransfer:
    """A point-to-point transfer of a device tensor carried by a host copy (gloo moves host memory only)."""

    def __init__(self, work, host, target):
        self.work, self.host, self.target = work, host, target

    def wait(self):
        self.work.wait()
        if self.target is not None:
            self.target.copy_(self.host)


def all_reduce(t, op=None, group=None):
    """dist.all_reduce in place; over gloo a device tensor is reduced through a host copy (see start_transfers)."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class _StagedGather:
    """all_gather_into_tensor of a device tensor carried by host copies (gloo moves host memory only)."""

    def __init__(self, work, host_out, target):
        self.work, self.host_out, self.target = work, host_out, target

    def wait(self):
        self.work.wait()
        self.target.copy_(self.host_out)


def start_all_gather(out, mine, group=None):
    """ONE collective: every rank's `mine` (equal shapes, contiguous) lands in out[rank * len(mine) : ...] on all
    ranks (RCCL all-gather over xGMI with the "nccl" backend; staged through host memory over gloo).  Returns a
    handle with .wait()."""
    if dist.get_backend(group) == 'gloo' and mine.is_cuda:
        host_in = mine.cpu()
        host_out = torch.empty(out.shape, dtype=out.dtype)
        return _StagedGather(dist.all_gather_into_tensor(host_out, host_in, group=group, async_op=True), host_out, out)
    return dist.all_gather_into_tensor(out, mine, group=group, async_op=True)


def start_transfers(ops, group=None):
    """ops: [('send' | 'recv', tensor, peer), ...] issued as ONE batch of point-to-point transfers (on the fully
    connected xGMI mesh every pair of GPUs has its own link, so a batch to all peers runs concurrently).  Returns
    handles with .wait().  Over RCCL (backend "nccl") the tensors travel device to device.  The gloo backend —
    the CPU rehearsal of this logic, and the one-GPU rehearsal of the multi-rank code, where several ranks share a
    device and RCCL cannot be used — moves host memory only: device tensors are staged through host copies there."""
    if not ops:
        return []
    if not (dist.get_backend(group) == 'gloo' and any(t.is_cuda for _, t, _ in ops)):
        return dist.batch_isend_irecv([dist.P2POp(dist.isend if kind == 'send' else dist.irecv, t, peer, group)
                                       for kind, t, peer in ops])
    works = []
    for kind, t, peer in ops:
        if kind == 'send':
            host = t.cpu()                                   # waits for the stream that produced t
            works.append(_StagedTransfer(dist.isend(host, peer, group), host, None))
        else:
            host = torch.empty(t.shape, dtype=t.dtype)
            works.append(_StagedTransfer(dist.irecv(host, peer, group), host, t))
    return works


class _NativeWait:
    """Handle of an exchange posted by the library (mh_exchange_begin): wait() queues the compute stream's wait."""

    def __init__(self, exchange, on_device_of):
        self.exchange, self.on_device_of = exchange, on_device_of

    def wait(self):
        from ._lib import ops
        ops.exchange_end(self.exchange, self.on_device_of)


class TorchRowMover:
    """Index-based stand-in for the halo staging kernels (CPU rehearsal over gloo)."""

    @staticmethod
    def gather(x, idx):
        return x[idx.long()].contiguous()

    @staticmethod
    def scatter(buf, idx, x):
        x[idx.long()] = buf


class HipRowMover:
    """Halo staging on the GPU (mh_gather_rows / mh_scatter_rows)."""

    @staticmethod
    def gather(x, idx):
        from .laplacian import gather_rows
        return gather_rows(x, idx)

    @staticmethod
    def scatter(buf, idx, x):
        from .laplacian import scatter_rows
        scatter_rows(buf, idx, x)


def halo_plan_torch(col_idx, n_cols, c0, c1, bounds):
    """Plain tensor restatement of mh_halo_plan_* (what the gloo rehearsal on CPU tensors uses, and what the
    GPU tests compare the kernels with): ascending remote column ids, their grouping by owning rank, and the
    columns renumbered to [own rows | halo rows]."""
    remote = (col_idx < c0) | (col_idx >= c1)
    halo_ids = torch.unique(col_idx[remote].long())
    peer_ptr = torch.searchsorted(halo_ids, bounds.long())
    slot = torch.searchsorted(halo_ids, col_idx.long())
    col_local = torch.where(remote, (c1 - c0) + slot, col_idx.long() - c0)
    return halo_ids.to(torch.int32), peer_ptr.to(torch.int32), col_local.to(torch.int32)


def _halo_plan(col_idx, n_cols, c0, c1, bounds):
    if col_idx.is_cuda:
        from .laplacian import halo_plan
        return halo_plan(col_idx, n_cols, c0, c1, bounds)
    return halo_plan_torch(col_idx, n_cols, c0, c1, bounds)


def exchange_index_lists(send, rank, world, group=None):
    """send[p]: int32 tensor for rank p (entry `rank` ignored) -> list of what every rank sent to this one.
    Lengths travel first (one all_gather), then the lists as batched point-to-point transfers."""
    dev = send[(rank + 1) % world].device if world > 1 else None
    recv = [None] * world
    if world == 1:
        return recv
    cdev = 'cpu' if dist.get_backend(group) == 'gloo' else dev       # gloo gathers host tensors
    counts = torch.tensor([0 if p == rank else int(send[p].numel()) for p in range(world)], dtype=torch.int64, device=cdev)
    table = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(table, counts, group=group)
    ops = []
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world
        recv[frm] = torch.empty(int(table[frm][rank]), dtype=torch.int32, device=dev)
        if send[to].numel():
            ops.append(('send', send[to].contiguous(), to))
        if recv[frm].numel():
            ops.append(('recv', recv[frm], frm))
    wait_all(start_transfers(ops, group))
    return recv


def exchange_volumes(row_ptr, col_idx, n, n_features, K, world):
    """Feature floats one rank receives per conv under the two decompositions, largest rank:
    (row partition: K exchanges of the remote rows its block references — the input's halo and K-1 intermediate
    orders; column sharding: one all-to-all of the K+1 orders).  Plain tensor code on a replicated CSR."""
    bounds = partition_bounds(row_ptr, world)
    rows_worst = 0
    for p in range(world):
        a, b = int(row_ptr[bounds[p]]), int(row_ptr[bounds[p + 1]])
        cols = col_idx[a:b]
        remote = torch.unique(cols[(cols < bounds[p]) | (cols >= bounds[p + 1])])   # rows of other ranks that rank p gathers
        rows_worst = max(rows_worst, int(remote.numel()))
    return row_partition_floats(rows_worst, n_features, K), column_sharding_floats(n, n_features, K, world)


def row_partition_floats(n_halo, n_features, K):
    """Floats a rank receives per conv in the row partition: its halo rows, once per propagated order."""
    return max(K, 0) * n_halo * 2 * n_features


def column_sharding_floats(n, n_features, K, world):
    return (K + 1) * n * 2 * n_features * (world - 1) // (world * world)


def choose_strategy(row_ptr, col_idx, n, n_features, K, world):
    """'rows' or 'cols' for a conv on `world` GPUs: the row partition moves little on graphs whose neighbours
    have nearby ids (only boundary rows travel); on a graph without locality every row is a boundary row and
    column sharding moves about K world / (K+1) times less."""
    if world == 1 or K < 1 or n_features % (4 * world) != 0:
        return 'rows'
    rows_floats, cols_floats = exchange_volumes(row_ptr, col_idx, n, n_features, K, world)
    return 'rows' if rows_floats <= cols_floats else 'cols'


class RowPartition:
    """One rank's share of a row-partitioned Laplacian, in COMPACT form: memory and set-up scale with 1/world.

    The rank holds the CSR of its row block [r0, r1) only.  Feature matrices are [n_local + n_halo, 2, F]:
    first the rank's own rows, then its halo — the remote rows its block references, ascending by global id
    and therefore grouped by owning rank.  The block's columns are renumbered to that buffer (mh_halo_plan_*:
    own column c -> c - r0, remote c -> n_local + slot), and split into L_int (own columns: operands local)
    and L_ext (halo columns, numbered from the start of the halo: its operand is the halo part alone, which
    may live in the same buffer behind the own rows or in a tensor of its own).  `make_operator(rp, ci, val, n_cols)` turns CSR tensors into an object with
    .spmm(X, out=, alpha=, z1=, beta1=) — the HIP MagneticLaplacian in production.

    Exchange of an order: every peer p gets the rows of this rank it references (`give[p]`, learned from p's
    halo plan at set-up; packed by the row-gather kernel, or sent in place when p references the whole
    block) and this rank receives, from every peer, its halo rows of that peer straight into their slice of
    the buffer — contiguous, no unpacking.  Point-to-point transfers issued as one batch: on the fully
    connected xGMI mesh every pair of GPUs has its own link.  Order k runs
    Z_k[own] = a L_int Z_{k-1} + b Z_{k-2}[own] while the halo of Z_{k-1} is in flight, waits, then adds
    a L_ext Z_{k-1}."""

    def __init__(self, rp_blk, ci_blk, val_blk, n, bounds, rank, world, make_operator, group=None, row_mover=None,
                 exchange_lists=exchange_index_lists):
        self.n, self.rank, self.world, self.group = n, rank, world, group
        self.mover = row_mover or (HipRowMover if rp_blk.is_cuda else TorchRowMover)
        self.bounds = b = [int(x) for x in bounds]
        self.r0, self.r1 = b[rank], b[rank + 1]
        self.n_local = self.r1 - self.r0
        if rp_blk.numel() != self.n_local + 1:
            raise ValueError('the CSR block has %d rows, the bounds give this rank %d' % (rp_blk.numel() - 1, self.n_local))
        dev = rp_blk.device
        self.local_nnz = int(ci_blk.numel())
        halo_ids, peer_ptr, col_local = _halo_plan(ci_blk, n, self.r0, self.r1, torch.tensor(b, dtype=torch.int32, device=dev))
        self.halo_ids, self.peer_ptr = halo_ids, [int(x) for x in peer_ptr.tolist()]
        self.n_halo = int(halo_ids.numel())
        self.n_buf = self.n_local + self.n_halo
        (rpi, cii, vi), (rpe, cie, ve) = split_columns(rp_blk, col_local, val_blk, 0, self.n_local)
        self.L_int = make_operator(rpi, cii, vi, self.n_local)
        self.L_ext = make_operator(rpe, (cie - self.n_local).to(cie.dtype), ve, self.n_halo)
        # tell every peer which of its rows this rank gathers (as row numbers inside the peer's block)
        self.need = [None if p == rank else (halo_ids[self.peer_ptr[p]:self.peer_ptr[p + 1]] - b[p]).to(torch.int32)
                     for p in range(world)]
        # give[p]: the rows of this rank's block that peer p gathers = need[rank] of peer p
        self.give = exchange_lists(self.need, rank, world, group)
        self.comm_enabled = True          # False: skip the transfers (timing the compute alone)
        self._make_operator, self._ext_csr = make_operator, (rpe, cie, ve)
        self.mode = 'overlap'
        self.int_grid_cap = 0             # workgroups of the own-column product in mode 'overlap_capped'
        self.wait_events = None           # a list: (event before, event after) of every wait for an exchange
        self.issue_seconds = None         # a list: host time spent issuing each exchange (packing launches + posting)
        self._gathered = {}               # all_gather mode: feature width -> (staging block, gathered buffer)
        self.L_ext_gathered = None
        self._exchange = None             # extension mode: handle of the library's exchange object (mh_exchange_*)

    MODES = ('overlap', 'overlap_capped', 'sequential', 'all_gather')
    NATIVE_MODE = 'extension'             # not in MODES: needs RCCL, i.e. one process per GPU

    def set_mode(self, mode, int_grid_cap=None):
        """How an order's halo exchange and its two products are arranged:
          'overlap'         the own-column product runs while the halo is in flight, the halo-column product after it
          'overlap_capped'  the same with the own-column product launched on `int_grid_cap` workgroups only, so that
                            the communication library's copy kernels find free compute units next to it
          'sequential'      the exchange completes first, then both products (no overlap)
          'all_gather'      the exchange is ONE all-gather of the ranks' own blocks (padded to the largest block) into a
                            buffer of all rows, which the halo-column product reads through columns renumbered to that
                            buffer; every row travels whether a peer references it or not, so it suits graphs without
                            locality, where the halo is nearly everything anyway; overlapped like 'overlap'
          'extension'       arranged like 'overlap', but an order's packing launches and its grouped RCCL send/receive
                            are posted by ONE call into the library (mh_exchange_begin, on a communicator and a stream
                            the library owns) and the wait by another (mh_exchange_end): no torch.distributed objects
                            and no interpreter time between the kernels and the communication library.  GPU tensors
                            and one process per GPU only (RCCL refuses two ranks on one device); the first use is a
                            collective (the communicator is created then)."""
        if mode not in self.MODES and mode != self.NATIVE_MODE:
            raise ValueError('mode must be one of %s' % (self.MODES + (self.NATIVE_MODE,),))
        if mode == self.NATIVE_MODE and self.world > 1 and self._exchange is None:
            self._exchange = self._create_native_exchange()
        self.mode = mode
        if int_grid_cap is not None:
            self.int_grid_cap = int(int_grid_cap)
        if mode == 'all_gather' and self.L_ext_gathered is None and self.world > 1:
            # halo slot -> global row id -> (owner, row inside the owner's block) -> owner * max_rows + row
            b = torch.tensor(self.bounds, dtype=torch.int64, device=self.halo_ids.device)
            self.max_rows = max(self.bounds[p + 1] - self.bounds[p] for p in range(self.world))
            gid = self.halo_ids.long()
            owner = torch.searchsorted(b, gid, right=True) - 1
            slot_to_gathered = (owner * self.max_rows + (gid - b[owner])).to(torch.int32)
            rpe, cie, ve = self._ext_csr
            self.L_ext_gathered = self._make_operator(rpe, slot_to_gathered[(cie - self.n_local).long()], ve,
                                                      self.world * self.max_rows)
        return self

    def _create_native_exchange(self):
        from ._lib import ops
        dev = self.halo_ids.device
        if dev.type != 'cuda':
            raise ValueError("mode 'extension' posts RCCL transfers of GPU tensors; this partition lives on %s" % dev)
        # the communicator's id is made on rank 0 and carried to the others over the group the partition already has
        id_ = ops.exchange_new_id().to(torch.int32) if self.rank == 0 else torch.zeros(128, dtype=torch.int32)
        id_ = all_reduce(id_.to(dev), group=self.group).to('cpu', torch.uint8)
        empty = torch.empty(0, dtype=torch.int32, device=dev)
        self._give_rows = [empty if (p == self.rank or g is None) else g.contiguous() for p, g in enumerate(self.give)]
        return ops.exchange_create(id_, self.rank, self.world, self._give_rows, self.peer_ptr, self.n_local, dev.index)

    def close(self):
        """Destroys the library's exchange object, if one was made (communicator, stream, staging memory)."""
        if self._exchange is not None:
            from ._lib import ops
            handle, self._exchange = self._exchange, None
            if self.mode == self.NATIVE_MODE:
                self.mode = 'overlap'
            ops.exchange_destroy(handle)

    def _gather_buffers(self, f, device, dtype):
        bufs = self._gathered.get(f)
        if bufs is None:
            stage = torch.zeros((self.max_rows, 2, f), dtype=dtype, device=device)
            bufs = self._gathered[f] = (stage, torch.empty((self.world * self.max_rows, 2, f), dtype=dtype, device=device))
        return bufs

    @classmethod
    def from_global(cls, row_ptr, col_idx, val, n, rank, world, make_operator, group=None, row_mover=None, bounds=None):
        """The rank's partition cut out of a CSR that the rank holds whole."""
        bounds = bounds or partition_bounds(row_ptr, world)
        rp, ci, v = slice_rows(row_ptr, col_idx, val, bounds[rank], bounds[rank + 1])
        return cls(rp, ci, v, n, bounds, rank, world, make_operator, group, row_mover)

    def new_buffer(self, f, device, dtype=torch.float32):
        return torch.empty((self.n_buf, 2, f), dtype=dtype, device=device)

    def block(self, buf):
        """The rank's own rows of a feature buffer."""
        return buf[:self.n_local]

    def sent_rows(self):
        """Rows this rank sends per exchange (over all peers)."""
        return sum(int(g.numel()) for p, g in enumerate(self.give) if p != self.rank and g is not None)

    def exchange(self, buf):
        """Start sending the peers their halo rows of buf's own block and receiving this rank's halo.
        Returns a handle for finish_exchange()."""
        return self.exchange_split(buf[:self.n_local], buf[self.n_local:])

    def exchange_split(self, own, halo):
        """The same exchange with the two parts given separately: `own` [n_local, 2, F] (any row stride: e.g. one
        order of a side-by-side block) is what the peers are served from, `halo` [n_halo, 2, F] (contiguous) is
        where this rank's halo rows arrive."""
        if self.world == 1 or not self.comm_enabled:
            return None
        if self.issue_seconds is not None:
            import time
            t0 = time.perf_counter()
            try:
                return self._start_exchange(own, halo)
            finally:
                self.issue_seconds.append(time.perf_counter() - t0)
        return self._start_exchange(own, halo)

    def _start_exchange(self, own, halo):
        if self.mode == 'all_gather':
            stage, gathered = self._gather_buffers(own.size(2), own.device, own.dtype)
            stage[:self.n_local].copy_(own)
            return ([start_all_gather(gathered, stage, self.group)], [stage])
        if self.mode == self.NATIVE_MODE:
            from ._lib import ops as native
            native.exchange_begin(self._exchange, own, halo)
            return ([_NativeWait(self._exchange, own)], [own, halo])
        ops, staged = [], []
        nl, pp = self.n_local, self.peer_ptr
        for step in range(1, self.world):
            to, frm = (self.rank + step) % self.world, (self.rank - step) % self.world
            g = self.give[to]
            if g.numel():
                # ascending + complete = the block itself: sent in place when it is contiguous
                mine = own if (g.numel() == nl and own.is_contiguous()) else self.mover.gather(own, g)
                ops.append(('send', mine, to))
                staged.append(mine)                                  # keep the send buffer alive
            if pp[frm + 1] > pp[frm]:
                ops.append(('recv', halo[pp[frm]:pp[frm + 1]], frm))
        return (start_transfers(ops, self.group), staged)

    def finish_exchange(self, handle):
        """Makes the stream wait for the exchange (RCCL: the compute stream waits for the communication stream; the
        host does not block).  With `wait_events` set, HIP events on the compute stream bracket the wait: their
        distance is how long the stream sat idle for the halo, i.e. what the overlap failed to hide."""
        if handle:
            ev = self.wait_events
            if ev is not None and handle[0] and torch.cuda.is_available():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                wait_all(handle[0])
                e1.record()
                ev.append((e0, e1))
            else:
                wait_all(handle[0])

    def propagate(self, z_prev, z_out, alpha=1.0, z_prev2=None, beta=0.0, pending=None, add=None, add_beta=1.0):
        """z_out[own] = alpha * (L z_prev)[own] + beta * z_prev2[own] (+ add_beta * add, a matrix of the rank's own
        rows).  `pending` is the handle of the exchange that completes z_prev's halo; only the halo-column part
        waits for it."""
        nl = self.n_local
        return self.propagate_split(z_prev[:nl], z_prev[nl:], z_out[:nl], alpha, None if z_prev2 is None else z_prev2[:nl],
                                    beta, pending, add, add_beta)

    def propagate_split(self, prev_own, prev_halo, out_own, alpha=1.0, prev2_own=None, beta=0.0, pending=None, add=None,
                        add_beta=1.0):
        """propagate() on operands given as (own rows, halo rows) pairs; the own-row matrices may be strided views."""
        mode = self.mode
        if mode == 'sequential':
            self.finish_exchange(pending)
            pending = None
        capped = mode == 'overlap_capped' and self.int_grid_cap > 0 and pending is not None
        if capped:
            from ._lib import ops
            ops.set_spmm_grid_cap(self.int_grid_cap)
        try:
            self.L_int.spmm(prev_own, out=out_own, alpha=alpha, z1=prev2_own, beta1=beta,
                            z2=add, beta2=add_beta if add is not None else 0.0)
        finally:
            if capped:
                ops.set_spmm_grid_cap(0)
        self.finish_exchange(pending)
        if self.n_halo:
            if mode == 'all_gather' and self.world > 1 and self.comm_enabled:
                gathered = self._gathered[prev_own.size(2)][1]
                self.L_ext_gathered.spmm(gathered, out=out_own, alpha=alpha, z1=out_own, beta1=1.0)
            else:
                self.L_ext.spmm(prev_halo, out=out_own, alpha=alpha, z1=out_own, beta1=1.0)
        return out_own

    def cheb_block(self, block, halos=None, halo_valid=False):
        """Fills block[:, :, k] = own rows of T_k(L) x for k >= 1, given block [n_local, 2, n_terms, F] (contiguous)
        whose order 0 holds the rank's rows of x: the orders side by side, the layout the weight GEMM reads in one
        go.  The halo rows of the orders live in `halos` (n_terms - 1 contiguous [n_halo, 2, F] tensors, made here
        when not given): they arrive there straight from the peers, and only the halo-column part of the operator
        reads them.  `halo_valid`: halos[0] already holds x's halo."""
        n_terms, f = block.size(2), block.size(3)
        if halo_valid and self.mode == 'all_gather':
            raise ValueError("halo_valid is not available in mode 'all_gather' (the halo lives in the gathered buffer)")
        if halos is None:
            halos = [torch.empty((self.n_halo, 2, f), dtype=block.dtype, device=block.device) for _ in range(max(n_terms - 1, 0))]
        order = lambda k: block[:, :, k, :]
        pending = None if (halo_valid or n_terms < 2) else self.exchange_split(order(0), halos[0])
        for k in range(1, n_terms):
            if k == 1:
                self.propagate_split(order(0), halos[0], order(1), 1.0, pending=pending)
            else:
                self.propagate_split(order(k - 1), halos[k - 1], order(k), 2.0, order(k - 2), -1.0, pending=pending)
            pending = self.exchange_split(order(k), halos[k]) if k < n_terms - 1 else None
        return block

    def cheb_terms(self, x_buf, n_terms, buffers=None, halo_valid=False):
        """[T_0 x, ..., T_{n-1} x] as compact buffers whose own rows are valid (and, for all but the last
        order, the halo).  x_buf holds the rank's rows of x; its halo is fetched first unless `halo_valid`."""
        zs = [x_buf]
        if halo_valid and self.mode == 'all_gather':
            raise ValueError("halo_valid is not available in mode 'all_gather' (the halo lives in the gathered buffer)")
        pending = None if (halo_valid or n_terms < 2) else self.exchange(x_buf)
        for k in range(1, n_terms):
            z = buffers[k - 1] if buffers is not None else torch.empty_like(x_buf)
            if k == 1:
                self.propagate(zs[0], z, 1.0, pending=pending)
            else:
                self.propagate(zs[k - 1], z, 2.0, zs[k - 2], -1.0, pending=pending)
            zs.append(z)
            pending = self.exchange(z) if k < n_terms - 1 else None
        return zs


    def clenshaw(self, ys):
        """Own rows of sum_k T_k(L) Y_k for Y given by its own rows, ys[k] = [n_local, 2, F] (any strides):
        b_K = Y_K, b_k = Y_k + 2 L b_{k+1} - b_{k+2}, result = Y_0 + L b_1 - b_2, every product with its halo
        exchange.  This is also the backward of cheb_terms (T_k of a Hermitian L is Hermitian), so the gradient uses
        the same blocks, the same plan and the same exchange."""
        n_terms = len(ys)
        f, dev = ys[0].size(2), ys[0].device
        if n_terms == 1:
            return ys[0].clone()
        b1 = self.new_buffer(f, dev)
        self.block(b1).copy_(ys[n_terms - 1])
        b2 = None
        for k in range(n_terms - 2, -1, -1):
            pending = self.exchange(b1)
            nb = self.new_buffer(f, dev)
            self.propagate(b1, nb, 2.0 if k > 0 else 1.0, b2, -1.0 if b2 is not None else 0.0, pending=pending, add=ys[k])
            b2, b1 = b1, nb
        return self.block(b1)


class _RowTermsFn(torch.autograd.Function):
    """Own rows x [n_local, 2, F] of a row-partitioned matrix -> the orders T_k(L) x side by side, own rows only,
    [n_local, 2, K+1, F].  Forward: K propagation orders with K halo exchanges.  Backward: the Clenshaw sum of the
    incoming gradients over the same partition (K more orders, K more exchanges)."""

    @staticmethod
    def forward(ctx, x_own, part, n_terms):
        block = torch.empty((part.n_local, 2, n_terms, x_own.size(2)), dtype=x_own.dtype, device=x_own.device)
        block[:, :, 0, :].copy_(x_own)
        part.cheb_block(block)
        ctx.part = part
        return block

    @staticmethod
    def backward(ctx, dz):
        return ctx.part.clenshaw([dz[:, :, k, :] for k in range(dz.size(2))]), None, None


# ---------------------------------------------------------------------------------------------
# distributed set-up: every rank builds its own row block
# ---------------------------------------------------------------------------------------------
def bounds_from_degrees(deg, world, row_cost=2):
    """Row boundaries balanced by (records per row + row_cost): `deg[i]` = out-degree + in-degree of node i,
    an upper bound of the row's nonzeros that is known before the Laplacian exists."""
    n = deg.numel()
    rp = torch.zeros(n + 1, dtype=torch.int64, device=deg.device)
    rp[1:] = torch.cumsum(deg.to(torch.int64), 0)
    return partition_bounds(rp, world, row_cost)


def build_row_block(edge_chunks, n, rank, world, q=0.25, group=None, bounds=None, max_eigen=2.0, timing=None):
    """This rank's row block of the magnetic Laplacian, built from the edges incident to its rows
    (mh_lap_block_*), plus the bounds of all ranks.  `edge_chunks()` returns an iterator over
    (row, col[, weight]) device tensors that together list every edge, in the same order on every rank
    (a replicated list or a seeded generator); it is called once for the degrees when `bounds` is not given and
    once for the records.  Ranks exchange only d^-1/2 (an all-reduce of n floats in which every node is
    non-zero on exactly one rank, so the sum is exact).  Returns (MagneticLaplacian block with global
    column ids, bounds).  `timing`: a dict that receives 'block_build_ms' (mh_lap_block_* + the d^-1/2 all-reduce,
    HIP events) apart from the streaming, counting and selection of the edges around it."""
    from .laplacian import LaplacianBlockBuild
    dev = None
    deg = None                                       # records per node (out-degree + in-degree over all edges)
    if bounds is None:
        for chunk in edge_chunks():
            row, col = chunk[0], chunk[1]
            dev = row.device
            d = torch.bincount(row.long(), minlength=n) + torch.bincount(col.long(), minlength=n)
            deg = d if deg is None else deg + d
        if deg is None:
            raise ValueError('edge_chunks() yielded nothing')
        bounds = bounds_from_degrees(deg, world)
    count_here = deg is None                         # bounds were given: the records are counted in the pass below
    r0, r1 = bounds[rank], bounds[rank + 1]
    parts = [[], [], [], [], [], []]                 # fwd row, col, w ; bwd row, col, w
    weighted = None
    for chunk in edge_chunks():
        row, col = chunk[0], chunk[1]
        w = chunk[2] if len(chunk) > 2 else None
        if weighted is None:
            weighted = w is not None
        elif weighted != (w is not None):
            raise ValueError('edge_chunks() mixes chunks with and without weights')
        dev = row.device
        if count_here and row.is_cuda:
            d = torch.bincount(row.long(), minlength=n) + torch.bincount(col.long(), minlength=n)
            deg = d if deg is None else deg + d
        f = (row >= r0) & (row < r1)
        b = (col >= r0) & (col < r1)
        parts[0].append(row[f]); parts[1].append(col[f])
        parts[3].append(row[b]); parts[4].append(col[b])
        if w is not None:
            parts[2].append(w[f]); parts[5].append(w[b])
    if weighted is None:
        raise ValueError('edge_chunks() yielded nothing (it must be re-iterable: it is called once per pass)')
    cat = lambda xs: torch.cat(xs) if len(xs) > 1 else xs[0]
    fwd = (cat(parts[0]), cat(parts[1])) + ((cat(parts[2]),) if weighted else ())
    bwd = (cat(parts[3]), cat(parts[4])) + ((cat(parts[5]),) if weighted else ())
    del parts
    timed = timing is not None and dev.type == 'cuda'
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    # every node's record count lets the block build carry the columns' degrees through its sort (unweighted builds)
    records = None if (weighted or deg is None) else deg.to(torch.int32)
    del deg
    build = LaplacianBlockBuild(fwd, bwd, n, r0, r1, max_eigen=max_eigen, node_records=records)
    del fwd, bwd, records
    dinv = torch.zeros(n, dtype=torch.float32, device=dev)
    dinv[r0:r1] = build.dinv_block
    if world > 1:
        all_reduce(dinv, group=group)
    blk = build.finish(dinv, q)
    if timed:
        ev1.record()
        ev1.synchronize()
        timing['block_build_ms'] = ev0.elapsed_time(ev1)
    return blk, bounds


class RowPartitionedConv:
    """One MagNetConv forward on a row-partitioned graph (what bench.py times with --gpus N > 1 in the row
    layout): the input's halo exchange, K propagation orders with K-1 more exchanges, then the row-local weight
    GEMM over the orders side by side and the bias.  `lap_block` is the rank's row block (global column ids),
    e.g. from build_row_block."""

    def __init__(self, lap_block, bounds, in_c, out_c, K, rank, world, device, seed=1, group=None):
        from .laplacian import MagneticLaplacian

        def make(rp, ci, v, n_cols):
            return MagneticLaplacian(rp, ci, v, n_cols)

        self.part = RowPartition(lap_block.row_ptr, lap_block.col_idx, lap_block.val, lap_block.n_cols, bounds, rank, world,
                                 make, group)
        self.K = K
        g = torch.Generator(device='cpu').manual_seed(seed)
        stdv = 1.0 / out_c ** 0.5
        self.weight = ((torch.rand(K + 1, in_c, out_c, generator=g) * 2 - 1) * stdv).to(device)
        self.bias = torch.zeros(1, out_c, device=device)
        part = self.part
        gen = torch.Generator(device=device).manual_seed(seed + 1000 * rank)
        # the rank's rows of the input and of the K orders side by side (one GEMM reads them), halo rows per order
        self.block = torch.empty((part.n_local, 2, K + 1, in_c), device=device)
        self.x = self.block[:, :, 0, :]
        self.x.copy_(torch.randn(part.n_local, 2, in_c, device=device, generator=gen))
        self.halos = [torch.zeros((part.n_halo, 2, in_c), device=device) for _ in range(K)]
        self.launches = []

    def step(self):
        part = self.part
        part.cheb_block(self.block, self.halos)
        rows, (k1, cin, cout) = part.n_local, self.weight.shape
        out = torch.addmm(self.bias, self.block.view(rows * 2, k1 * cin), self.weight.view(k1 * cin, cout))
        return out.view(rows, 2, cout)


# ---------------------------------------------------------------------------------------------
# feature-column sharding: no exchange inside the recursion
# ---------------------------------------------------------------------------------------------
def column_bounds(n_features, world, quantum=4):
    """world+1 column boundaries, multiples of `quantum` where the width allows (the vector SpMM
    path wants slices that are multiples of 4 floats)."""
    if n_features % (quantum * world) == 0 or n_features >= quantum * world:
        units = n_features // quantum
        b = [quantum * (units * p // world) for p in range(world)] + [n_features]
    else:
        b = [n_features * p // world for p in range(world)] + [n_features]
    return b


def all_to_all_blocks(send, recv, rank, world, group=None, enabled=True):
    """send[p] goes to rank p, recv[p] is filled by rank p (lists of contiguous tensors; entry `rank`
    is ignored).  Batched point-to-point: one direct transfer per GPU pair on the xGMI mesh."""
    ops = []
    if not enabled:
        return ops
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world
        if send[to].numel():
            ops.append(('send', send[to], to))
        if recv[frm].numel():
            ops.append(('recv', recv[frm], frm))
    return start_transfers(ops, group)


class ColumnSharded:
    """Every rank holds the WHOLE Laplacian (4.8 GB at 400M nonzeros: cheap next to 288 GB) and a
    slice of the feature columns.  The K propagation orders then need no communication at all:
    column f of L X depends on column f of X only.  The per-order weight GEMM mixes columns, so
    before it every order is regrouped from column slices to row blocks with an all-to-all (each
    rank sends 1/world of what it holds to every peer); the all-to-all of order k is in flight while
    order k+1 is computed.  Per conv that moves
    (K+1) * N * 8F * (world-1)/world^2 bytes per rank, against K * N * 8F * (world-1)/world
    for the row partition on a graph without locality (K exchanges per conv: row_partition_floats) —
    K * world / (K+1) times less.  Volumes only: no transfer of either layout has been timed on xGMI yet."""

    def __init__(self, operator, n, n_features, rank, world, group=None):
        self.op, self.n, self.rank, self.world, self.group = operator, n, rank, world, group
        self.cb = column_bounds(n_features, world)
        self.c0, self.c1 = self.cb[rank], self.cb[rank + 1]
        self.rb = [n * p // world for p in range(world)] + [n]         # output row blocks
        self.r0, self.r1 = self.rb[rank], self.rb[rank + 1]
        self.comm_enabled = True          # False: skip the transfers (timing the compute alone)

    def cheb_terms(self, x_slice, n_terms, after_each=None):
        """Column slices of [T_0 x, ..., T_{n-1} x]; `after_each(k, z)` is called as soon as order k
        has been launched (used to start that order's regrouping while the next one computes)."""
        zs = [x_slice]
        if after_each is not None:
            after_each(0, x_slice)
        for k in range(1, n_terms):
            z = torch.empty_like(x_slice)
            if k == 1:
                self.op.spmm(zs[0], out=z)
            else:
                self.op.spmm(zs[k - 1], out=z, alpha=2.0, z1=zs[k - 2], beta1=-1.0)
            zs.append(z)
            if after_each is not None:
                after_each(k, z)
        return zs

    def start_regroup(self, z):
        """Begin the all-to-all of ONE order: z [N, 2, fc_me] -> my rows of every rank's columns.
        Row slices of z are contiguous, so they are sent in place.  Returns a handle for finish_regroup."""
        rows_me = self.r1 - self.r0
        send = [z[self.rb[p]:self.rb[p + 1]] for p in range(self.world)]
        recv = [send[p] if p == self.rank else
                torch.empty((rows_me, 2, self.cb[p + 1] - self.cb[p]), dtype=z.dtype, device=z.device)
                for p in range(self.world)]
        return all_to_all_blocks(send, recv, self.rank, self.world, self.group, self.comm_enabled), recv, z

    def finish_regroup(self, handle):
        """-> [rows_me, 2, F] (my rows, all columns) of that order."""
        works, recv, _ = handle
        wait_all(works)
        return recv[0] if self.world == 1 else torch.cat(recv, dim=-1)

    def to_row_blocks(self, zs):
        """All orders at once: list over orders of [rows_me, 2, F]."""
        return [self.finish_regroup(h) for h in [self.start_regroup(z) for z in zs]]

    def _row_operator(self, p):
        """The operator restricted to the output rows of rank p's block (built on first use)."""
        ops = getattr(self, '_row_ops', None)
        if ops is None:
            ops = self._row_ops = {}
        if p not in ops:
            ops[p] = self.op.rows(self.rb[p], self.rb[p + 1])
        return ops[p]

    def last_order_staged(self, z_prev, z_prev2):
        """The LAST order computed block by block — the rows each peer will own first, this rank's own
        rows last — with every peer's block sent the moment it is done.  Nothing follows the last order
        that its all-to-all could hide under, so it is hidden under the order's own remaining blocks.
        Step s pairs rank r with r+s (send) and r-s (receive) on every rank at once, so each batch of
        transfers is a permutation.  Returns a handle for finish_regroup."""
        z = torch.empty_like(z_prev)
        rows_me = self.r1 - self.r0
        recv = [None] * self.world
        works = []

        def block(p):
            a, b = self.rb[p], self.rb[p + 1]
            if b > a:
                if z_prev2 is None:
                    self._row_operator(p).spmm(z_prev, out=z[a:b])
                else:
                    self._row_operator(p).spmm(z_prev, out=z[a:b], alpha=2.0, z1=z_prev2[a:b], beta1=-1.0)
            return z[a:b]

        for step in range(1, self.world):
            to, frm = (self.rank + step) % self.world, (self.rank - step) % self.world
            part = block(to)
            recv[frm] = torch.empty((rows_me, 2, self.cb[frm + 1] - self.cb[frm]), dtype=z.dtype, device=z.device)
            ops = []
            if part.numel():
                ops.append(('send', part, to))
            if recv[frm].numel():
                ops.append(('recv', recv[frm], frm))
            if self.comm_enabled:
                works += start_transfers(ops, self.group)
        recv[self.rank] = block(self.rank)
        return works, recv, z

    def conv(self, x_slice, weight, bias):
        """Rows [r0, r1) of sum_k T_k X W_k + bias, given this rank's column slice of X.  The
        regrouping of order k travels while order k+1 is being computed; the last order is computed in
        row blocks whose transfers overlap the blocks still to come (last_order_staged)."""
        k1, cin, cout = weight.shape
        handles = []
        staged = self.world > 1 and k1 > 1 and hasattr(self.op, 'rows')
        zs = self.cheb_terms(x_slice, k1 - 1 if staged else k1,
                             after_each=lambda k, z: handles.append(self.start_regroup(z)))
        if staged:
            handles.append(self.last_order_staged(zs[-1], zs[-2] if k1 > 2 else None))
        rows_me = self.r1 - self.r0
        # the peers' column slices of every order land side by side in one block, which ONE GEMM with the stacked
        # weights then reads (the copy out of the receive buffers is the one a concatenation would make)
        block = torch.empty((rows_me, 2, k1, cin), dtype=x_slice.dtype, device=x_slice.device)
        for k, (works, recv, _) in enumerate(handles):
            wait_all(works)
            for p in range(self.world):
                block[:, :, k, self.cb[p]:self.cb[p + 1]].copy_(recv[p])
        bf, wf = block.view(rows_me * 2, k1 * cin), weight.reshape(k1 * cin, cout)
        out = torch.addmm(bias, bf, wf) if bias is not None else torch.mm(bf, wf)
        return out.view(rows_me, 2, cout)


class ColumnShardedConv:
    """bench.py's multi-GPU step in the column-sharded form (same interface as RowPartitionedConv)."""

    def __init__(self, lap, in_c, out_c, K, rank, world, device, seed=1):
        self.shard = ColumnSharded(lap, lap.n_rows, in_c, rank, world)
        self.K = K
        g = torch.Generator(device='cpu').manual_seed(seed)
        stdv = 1.0 / out_c ** 0.5
        self.weight = ((torch.rand(K + 1, in_c, out_c, generator=g) * 2 - 1) * stdv).to(device)
        self.bias = torch.zeros(1, out_c, device=device)
        gen = torch.Generator(device=device).manual_seed(seed + 1000 * rank)
        self.x = torch.randn(lap.n_rows, 2, self.shard.c1 - self.shard.c0, device=device, generator=gen)
        self.launches = []

    def step(self):
        return self.shard.conv(self.x, self.weight, self.bias)


# ---------------------------------------------------------------------------------------------
# training over several GPUs: ChebNet with the layers alternating between the two layouts
# ---------------------------------------------------------------------------------------------
def _regroup(x, src_bounds, dst_bounds, rank, world, group, to_cols):
    """to_cols=True : x [rows_me, 2, k, C]  (my rows, all columns)  ->  [N, 2, k, fc_me]
       to_cols=False: x [N, 2, k, fc_me]    (all rows, my columns)  ->  [rows_me, 2, k, C]
    src_bounds / dst_bounds are (row bounds, column bounds)."""
    rb, cb = src_bounds, dst_bounds
    if to_cols:
        # every peer's rows of my columns arrive in their row range of the result (row ranges are contiguous)
        out = torch.empty((rb[-1],) + tuple(x.shape[1:-1]) + (cb[rank + 1] - cb[rank],), dtype=x.dtype, device=x.device)
        send = [None if p == rank else x[..., cb[p]:cb[p + 1]].contiguous() for p in range(world)]
        recv = [out[rb[p]:rb[p + 1]] for p in range(world)]
        recv[rank].copy_(x[..., cb[rank]:cb[rank + 1]])
        send[rank] = recv[rank]
        wait_all(all_to_all_blocks(send, recv, rank, world, group))
        return out
    send = [x[rb[p]:rb[p + 1]].contiguous() for p in range(world)]
    recv = [send[p] if p == rank else
            torch.empty((rb[rank + 1] - rb[rank],) + tuple(x.shape[1:-1]) + (cb[p + 1] - cb[p],), dtype=x.dtype, device=x.device)
            for p in range(world)]
    wait_all(all_to_all_blocks(send, recv, rank, world, group))
    return torch.cat(recv, dim=-1)


class _RegroupFn(torch.autograd.Function):
    """All-to-all between the row-block layout and the column-slice layout; its backward is the
    opposite regrouping of the gradient."""

    @staticmethod
    def forward(ctx, x, rb, cb, rank, world, group, to_cols):
        ctx.args = (rb, cb, rank, world, group, to_cols)
        return _regroup(x, rb, cb, rank, world, group, to_cols)

    @staticmethod
    def backward(ctx, g):
        rb, cb, rank, world, group, to_cols = ctx.args
        return _regroup(g.contiguous(), rb, cb, rank, world, group, not to_cols), None, None, None, None, None, None


class _OrdersGemmFn(torch.autograd.Function):
    """Row-local tail of a conv: orders side by side z [R, 2, k1, C_in] -> sum_k z_k W_k + bias, complex ReLU fused
    (one GEMM, the bias / ReLU kernel in place).  Backward: the ReLU mask applied in place of a copy, the weight
    gradient as a split-K gram of two tall matrices (a single GEMM would have too few output tiles for 256 CUs), the
    input gradient only when the orders need one."""

    @staticmethod
    def forward(ctx, z, w, b, relu):
        from .layer import sparse_magnet as M
        rows, _, k1, cin = z.shape
        cout = w.size(2)
        zf = z.reshape(rows * 2, k1 * cin)
        out = torch.mm(zf, w.reshape(k1 * cin, cout)).view(rows, 2, cout)
        mask = M._bias_relu_(out, b, relu)
        ctx.save_for_backward(zf, w, mask)
        ctx.has_bias, ctx.shape = b is not None, (rows, k1, cin, cout)
        return out

    @staticmethod
    def backward(ctx, g):
        from .layer import sparse_magnet as M
        zf, w, mask = ctx.saved_tensors
        rows, k1, cin, cout = ctx.shape
        g = M._relu_bwd(g, mask) if mask is not None else g.contiguous()
        g2 = g.view(rows * 2, cout)
        dz = torch.mm(g2, w.reshape(k1 * cin, cout).t()).view(rows, 2, k1, cin) if ctx.needs_input_grad[0] else None
        dw = M._gram(zf, g2).view(k1, cin, cout) if ctx.needs_input_grad[1] else None
        db = g2.sum(0).view(1, cout) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dz, dw, db, None


class ShardedChebNet:
    """Runs a ChebNet (magnet_amd.layer.sparse_magnet.ChebNet, same parameters / state_dict) on
    `world` GPUs.  Every rank holds the whole Laplacian; node features enter as row blocks.  Layers
    alternate between two forms so that each needs ONE all-to-all:
      rows in  -> GEMM first: Y = X [W_0|..|W_K] on the local rows, regroup to column slices, Clenshaw
                  recurrence on the slices (no communication), bias + complex ReLU fused -> columns out
      cols in  -> SpMM first: T_k X on the column slices (no communication), regroup to row blocks,
                  sum_k Z_k W_k + bias, complex ReLU -> rows out
    The classification head is row-local.  Parameter gradients are summed over ranks (all-reduce);
    the loss is the global mean over the training nodes."""

    # As on one GPU (ChebConv.cache_input_propagation), the propagated input T_k(L) X of the FIRST layer does
    # not depend on any weight.  With this switch on it is computed once on the column slices, regrouped to
    # this rank's row block and kept for as long as the same, unmodified input tensor is passed: the first
    # layer then costs one row-local GEMM per forward, no SpMM and no communication.
    cache_input_propagation = False

    # The evaluation forward of an epoch and the training forward of the next run under the same weights and differ only
    # in the dropout of the head.  With this switch on, epoch() records the evaluation's pass through the convolutions and
    # the next call backpropagates through it: one pass (and one set of exchanges) per epoch instead of two, same losses
    # and parameters (magnet_amd.train.SharedForwardEpochs is the single-GPU form).
    shared_forward = False
    _kept_pass = None

    def __init__(self, net, operator, n, rank, world, group=None):
        from .layer.sparse_magnet import ChebConv, complex_relu_layer
        self.net, self.op, self.n, self.rank, self.world, self.group = net, operator, n, rank, world, group
        self._input_cache = None
        self.rb = [n * p // world for p in range(world)] + [n]
        self.r0, self.r1 = self.rb[rank], self.rb[rank + 1]
        mods = list(net.Chebs)
        self.layers = []
        i = 0
        while i < len(mods):
            assert isinstance(mods[i], ChebConv)
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], complex_relu_layer)
            self.layers.append((mods[i], relu))
            i += 2 if relu else 1

    def _regroup(self, x, cb, to_cols):
        return _RegroupFn.apply(x, self.rb, cb, self.rank, self.world, self.group, to_cols)

    def trunk(self, x_rows):
        """x_rows: [rows_me, 2, C_in] row block of the packed input -> [rows_me, 2, F] row block."""
        from .layer.sparse_magnet import _ChebTermsFn, _ClenshawFn, _ComplexReluFn
        x, layout = x_rows, 'rows'
        rows_me = self.r1 - self.r0
        layers = self.layers
        conv0, relu0 = layers[0]
        if not x_rows.requires_grad and conv0.weight.size(1) >= self.world:
            # An input that needs no gradient (the node features): its propagated orders T_k(L) X are constants of the
            # step.  They are computed on column slices (one all-to-all of X there, one of the K+1 orders back) and the
            # layer is the row-local GEMM on them; the weight gradient is then a product with these saved orders,
            # where the GEMM-first form below would need K more SpMMs in backward for T_k(L) dOut.  Same number of
            # orders on the wire per training step as the GEMM-first form (whose all-to-all also runs in backward).
            from .layer.sparse_magnet import _cheb_terms
            w, b = conv0.weight, conv0.bias
            k1, cin, cout = w.shape
            c = self._input_cache if self.cache_input_propagation else None
            if c is None or c[0] is not x_rows or c[1] != x_rows._version or c[3] != k1:
                with torch.no_grad():
                    cb = column_bounds(cin, self.world)
                    xc = _regroup(x_rows.unsqueeze(2), self.rb, cb, self.rank, self.world, self.group, True).squeeze(2)
                    z = _cheb_terms(self.op, xc, k1)                                   # [N, 2, k1, my columns]
                    r = _regroup(z, self.rb, cb, self.rank, self.world, self.group, False)   # [rows_me, 2, k1, C_in]
                c = (x_rows, x_rows._version, r, k1)
                if self.cache_input_propagation:
                    self._input_cache = c
            x = _OrdersGemmFn.apply(c[2], w, b, relu0)
            layers = layers[1:]
        for conv, relu in layers:
            w, b = conv.weight, conv.bias
            k1, cin, cout = w.shape
            if layout == 'rows':
                y = torch.matmul(x.reshape(rows_me * 2, cin), w.permute(1, 0, 2).reshape(cin, k1 * cout))
                cb = column_bounds(cout, self.world)
                yc = self._regroup(y.view(rows_me, 2, k1, cout), cb, True)
                bs = None if b is None else b.reshape(-1)[cb[self.rank]:cb[self.rank + 1]]
                x = _ClenshawFn.apply(yc, bs, self.op, relu)
                layout, cur_cb = 'cols', cb
            else:
                z = _ChebTermsFn.apply(x, self.op, k1)
                r = self._regroup(z, cur_cb, False)
                x = _OrdersGemmFn.apply(r, w, b, relu)
                layout = 'rows'
        if layout == 'cols':
            x = self._regroup(x.unsqueeze(2), cur_cb, False).squeeze(2)
        return x

    def forward(self, x_rows):
        """-> log-probabilities [1, label_dim, rows_me] of the rank's nodes."""
        return self.head(self.trunk(x_rows))

    def head(self, p):
        """Row-local tail of the network on a trunk output: unwind, dropout, class scores, log-softmax."""
        import torch.nn.functional as F
        x = p.reshape(p.size(0), -1)
        if self.net.dropout > 0:
            x = F.dropout(x, self.net.dropout, training=self.net.training)
        from .layer.sparse_magnet import _HeadFn
        return F.log_softmax(_HeadFn.apply(x, self.net.Conv.weight.squeeze(-1), self.net.Conv.bias), dim=1)

    def all_reduce_grads(self):
        for p in self.net.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if self.world > 1:
                all_reduce(p.grad, group=self.group)

    def epoch(self, opt, x_rows, label_rows, train_rows, val_rows, n_train, n_val):
        """One epoch of the reference loop (src/sparse_Magnet.py:143-176) on this rank's rows:
        label_rows [1, rows_me], train_rows / val_rows index tensors into the local rows (or train.NodeTargets made
        from them: no gather of the predictions), n_train / n_val the GLOBAL numbers of training / validation nodes."""
        import torch.nn.functional as F
        from .train import NodeTargets

        def nll_sum(preds, idx):
            # NodeTargets of the local rows: the sum over the rank's subset without gathering the predictions
            if isinstance(idx, NodeTargets):
                return idx.loss(preds, reduction='sum')
            return F.nll_loss(preds[:, :, idx], label_rows[:, idx], reduction='sum')
        def state():
            return tuple((id(t), t._version) for t in [x_rows] + list(self.net.parameters()))
        kept, self._kept_pass = self._kept_pass, None
        self.net.train()
        # a pass through the convolutions kept by the evaluation of the epoch before (shared_forward): same weights,
        # same input, and nothing in the trunk depends on train / eval mode.  Versions advance alike on every rank,
        # so all ranks take the same branch.
        p = kept[1] if (kept is not None and kept[0] == state()) else self.trunk(x_rows)
        preds = self.head(p)
        loss = nll_sum(preds, train_rows) / n_train
        del p, preds, kept
        opt.zero_grad()
        loss.backward()
        self.all_reduce_grads()
        opt.step()
        self.net.eval()
        if self.shared_forward:
            p = self.trunk(x_rows)                           # recorded: the next epoch backpropagates through it
            self._kept_pass = (state(), p, x_rows)
            with torch.no_grad():
                vloss = nll_sum(self.head(p.detach()), val_rows) / n_val
        else:
            with torch.no_grad():
                preds = self.forward(x_rows)
                vloss = nll_sum(preds, val_rows) / n_val
        stats = torch.stack([loss.detach(), vloss])
        if self.world > 1:
            all_reduce(stats, group=self.group)
        return stats[0], stats[1]


class RowShardedChebNet(ShardedChebNet):
    """The same ChebNet trained on the ROW partition: every rank holds its row block of the Laplacian only
    (build_row_block), its rows of the features, and the halo rows it gathers.  A conv is K propagation orders with
    their halo exchanges (_RowTermsFn), then the row-local weight GEMM over the orders side by side, bias and complex
    ReLU; its backward is the Clenshaw sum over the same partition (Hermitian operator: same blocks, same plan).
    Parameter gradients are summed over the ranks.  Memory per rank: the block's CSR and (n_local + n_halo) feature
    rows per order — nothing of full height."""

    def __init__(self, net, part):
        from .layer.sparse_magnet import ChebConv, complex_relu_layer
        self.net, self.part, self.n = net, part, part.n
        self.rank, self.world, self.group = part.rank, part.world, part.group
        self.rb = list(part.bounds)
        self.r0, self.r1 = part.r0, part.r1
        self.cache_input_propagation = False
        self._input_cache = None
        mods = list(net.Chebs)
        self.layers = []
        i = 0
        while i < len(mods):
            assert isinstance(mods[i], ChebConv)
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], complex_relu_layer)
            self.layers.append((mods[i], relu))
            i += 2 if relu else 1

    def trunk(self, x_rows):
        from .layer.sparse_magnet import _ComplexReluFn
        x = x_rows
        rows_me = self.r1 - self.r0
        for i, (conv, relu) in enumerate(self.layers):
            w, b = conv.weight, conv.bias
            k1, cin, cout = w.shape
            if i == 0 and self.cache_input_propagation and not x.requires_grad:
                # the propagated orders of an input that needs no gradient do not depend on any weight: kept for as
                # long as the same, unmodified tensor is passed (no SpMM and no exchange in this layer afterwards)
                c = self._input_cache
                if c is None or c[0] is not x or c[1] != x._version or c[3] != k1:
                    with torch.no_grad():
                        c = self._input_cache = (x, x._version, _RowTermsFn.apply(x, self.part, k1), k1)
                z = c[2]
            else:
                z = _RowTermsFn.apply(x, self.part, k1)                              # [rows_me, 2, k1, cin]
            x = _OrdersGemmFn.apply(z, w, b, relu)
        return x

