"""The reference drivers' epoch bodies (what "epoch time" means): one full-batch training step
(forward, loss, backward, Adam) followed by one evaluation forward.

    node_epoch  — reference src/sparse_Magnet.py:143-176
    link_epoch  — reference src/Edge_sparseMagnet.py:168-199
    SharedForwardEpochs — the same epochs, the evaluation forward shared with the next training forward
    GraphedSharedEpochs — those replayed as two HIP graphs
Checkpoint writes (src/sparse_Magnet.py:188-197, src/Edge_sparseMagnet.py:210-215) are the caller's
business and are not part of the timed epoch.
"""
import torch
import torch.nn.functional as F


IGNORE = -100          # F.nll_loss's default ignore_index


class NodeTargets:
    """The labels of a node subset in the form the loss reads without gathering: `label` [1, N] with every node
    outside the subset set to IGNORE, and the subset's size as a device scalar.  `index` is a boolean [N] mask or
    an int64 index tensor, as node_epoch takes them.

    The reference computes criterion(preds[:, :, index], label[:, index]) (src/sparse_Magnet.py:155): a gather of
    the predictions, and in backward an index-put (with a sort of the indices) into a zero tensor.  The mean of
    the negative log-likelihood over the subset is the same number as nll_loss over ALL nodes with the other
    nodes ignored — no gather, no index-put, no sort; made once per split, reused every epoch."""

    def __init__(self, label, index):
        lab = torch.full_like(label, IGNORE)
        lab[:, index] = label[:, index]
        self.label = lab
        self.count = (lab != IGNORE).sum()

    def loss(self, preds, reduction='mean'):
        return F.nll_loss(preds, self.label, ignore_index=IGNORE, reduction=reduction)

    def accuracy(self, preds):
        return (preds.max(dim=1)[1] == self.label).sum().float() / self.count


def _loss_acc(preds, label, index, criterion):
    if isinstance(index, NodeTargets):
        return index.loss(preds), index.accuracy(preds)
    loss = criterion(preds[:, :, index], label[:, index])
    acc = (preds.max(dim=1)[1][:, index] == label[:, index]).float().mean()
    return loss, acc


def node_epoch(model, opt, X_real, X_img, label, train_index, val_index, criterion=None):
    """label: [1, N] class ids; train_index / val_index: boolean [N] masks as in the reference, int64 index
    tensors (required under GraphedEpoch: boolean indexing synchronises), or NodeTargets made from either
    (same loss and accuracy without the per-epoch gather and scatter; `criterion` is then NLL).
    Returns (train_loss, train_acc, val_loss, val_acc) as device scalars (no host sync)."""
    criterion = criterion or torch.nn.NLLLoss()
    model.train()
    preds = model(X_real, X_img)
    train_loss, train_acc = _loss_acc(preds, label, train_index, criterion)
    opt.zero_grad()
    train_loss.backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        preds = model(X_real, X_img)
        val_loss, val_acc = _loss_acc(preds, label, val_index, criterion)
    return train_loss.detach(), train_acc, val_loss, val_acc


def link_epoch(model, opt, X_real, X_img, train_index, y_train, val_index, y_val):
    """train_index / val_index: [M, 2] node pairs, y_*: [M] class ids."""
    model.train()
    out = model(X_real, X_img, train_index)
    train_loss = F.nll_loss(out, y_train)
    train_acc = (out.max(dim=1)[1] == y_train).float().mean()
    opt.zero_grad()
    train_loss.backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        out = model(X_real, X_img, val_index)
        val_loss = F.nll_loss(out, y_val)
        val_acc = (out.max(dim=1)[1] == y_val).float().mean()
    return train_loss.detach(), train_acc, val_loss, val_acc


class SharedForwardEpochs:
    """The same epochs with ONE pass through the convolutions per epoch instead of two.

    The reference's epoch is: training forward, backward, Adam step, evaluation forward (src/sparse_Magnet.py:153-176,
    src/Edge_sparseMagnet.py:176-199).  The evaluation forward of epoch e and the training forward of epoch e + 1 run
    under the same weights (nothing steps in between), and train / eval mode only reaches the dropout in front of the
    last linear map: the convolutions (model.trunk) compute the same tensor in both.  So the trunk pass made for the
    evaluation is recorded by autograd and kept; the next epoch applies the head in training mode to it (fresh dropout
    mask, as the reference draws one) and backpropagates through the kept pass.  Every loss, accuracy and parameter is
    what node_epoch / link_epoch give (the first epoch makes one extra trunk pass); full-batch models with K orders
    per layer save half of their forward SpMMs.

    The kept pass is used only while the parameters and the input tensors are the objects and versions it was made
    from (optimizer steps made elsewhere, load_state_dict, in-place writes to X invalidate it); reset() drops it.
    Memory: the saved activations of one forward stay alive between epochs (the peak — one training forward — is
    unchanged).  Not for GraphedEpoch: the first epoch differs from the others."""

    def __init__(self, model, opt):
        self.model, self.opt = model, opt
        self._kept = None

    def reset(self):
        self._kept = None

    def _key(self, X_real, X_img):
        ts = [X_real, X_img] + list(self.model.parameters())
        return tuple((id(t), t._version) for t in ts)

    def _trunk(self, X_real, X_img):
        kept, self._kept = self._kept, None
        if kept is not None and kept[0] == self._key(X_real, X_img):
            return kept[1]
        return self.model.trunk(X_real, X_img)

    def _step(self, loss):
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()

    def _evaluation_trunk(self, X_real, X_img):
        self.model.eval()
        p = self.model.trunk(X_real, X_img)              # recorded: the next epoch backpropagates through it
        self._kept = (self._key(X_real, X_img), p, X_real, X_img)     # the inputs held: their ids stay theirs
        return p.detach()

    def node_epoch(self, X_real, X_img, label, train_index, val_index, criterion=None):
        """Arguments and results of node_epoch."""
        criterion = criterion or torch.nn.NLLLoss()
        model = self.model
        p = self._trunk(X_real, X_img)
        model.train()
        preds = model.head(p)
        train_loss, train_acc = _loss_acc(preds, label, train_index, criterion)
        del p, preds
        self._step(train_loss)
        p = self._evaluation_trunk(X_real, X_img)
        with torch.no_grad():
            val_loss, val_acc = _loss_acc(model.head(p), label, val_index, criterion)
        return train_loss.detach(), train_acc, val_loss, val_acc

    def link_epoch(self, X_real, X_img, train_index, y_train, val_index, y_val):
        """Arguments and results of link_epoch."""
        model = self.model
        p = self._trunk(X_real, X_img)
        model.train()
        out = model.head(p, train_index)
        train_loss = F.nll_loss(out, y_train)
        train_acc = (out.max(dim=1)[1] == y_train).float().mean()
        del p, out
        self._step(train_loss)
        p = self._evaluation_trunk(X_real, X_img)
        with torch.no_grad():
            out = model.head(p, val_index)
            val_loss = F.nll_loss(out, y_val)
            val_acc = (out.max(dim=1)[1] == y_val).float().mean()
        return train_loss.detach(), train_acc, val_loss, val_acc


class GraphedSharedEpochs:
    """SharedForwardEpochs replayed as HIP graphs: the small reference configurations are bound by kernel launches, and
    with the shared forward an epoch has a third fewer of them.  Two graphs in one memory pool:
        A  the evaluation half: trunk pass under the current weights (recorded by autograd), evaluation head, statistics
        B  the training half:   training head on A's trunk output, loss, backward through A's pass, optimizer step
    An epoch is B then A (A once more in front of the first epoch), which is the reference's order of operations
    (src/sparse_Magnet.py:153-176): training step, then the evaluation under the stepped weights.  A's activations live
    at fixed addresses of the shared pool, so B — captured after A, consuming those tensors — finds on every replay what
    the latest replay of A wrote there.

    trunk()            -> the convolutions' output p (model.trunk on the fixed inputs)
    train_stats(p)     -> (loss, accuracy) of the training head on p (called with the model in training mode)
    eval_stats(p)      -> (loss, accuracy) of the evaluation head on p (evaluation mode, no autograd)
    The optimizer must be capturable.  The warm-up epochs (at least one: the optimizer's state must exist before the
    capture, outside the graphs' pool) are real optimizer steps: a caller that wants to start from the initial weights
    restores parameters and optimizer state in place afterwards (the first call refreshes A).  After a call the
    parameters' .grad tensors are not meaningful (they live in the pool, where A's replay writes its temporaries);
    the returned statistics are copied out of the pool inside the graphs."""

    def __init__(self, model, opt, trunk, train_stats, eval_stats, warmup=3):
        if warmup < 1:
            raise ValueError('at least one warm-up epoch is needed: the optimizer state must be allocated before the capture')
        def train_half(p):
            model.train()
            loss, acc = train_stats(p)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss.detach(), acc

        def eval_half():
            model.eval()
            p = trunk()                                   # recorded: the training half backpropagates through it
            with torch.no_grad():
                stats = eval_stats(p.detach())
            return p, stats
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            p, _ = eval_half()
            for _ in range(warmup):                       # builds SpMM plans, autotunes GEMMs, primes the allocator
                train_half(p)
                p, _ = eval_half()
            del p
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # the statistics leave the graphs through buffers outside their pool: what B allocates may lie where A's
        # temporaries lay (the pool hands freed memory on), and A is replayed after B
        self.stats = torch.zeros(4, device=next(model.parameters()).device)
        self.eval_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.eval_graph):
            p, val = eval_half()
            self.stats[2].copy_(val[0]); self.stats[3].copy_(val[1])
        self.train_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.train_graph, pool=self.eval_graph.pool()):
            tr = train_half(p)
            self.stats[0].copy_(tr[0]); self.stats[1].copy_(tr[1])
        del val, tr
        self._p = p                                       # the pool's memory stays what the graphs address
        self._primed = False

    def __call__(self):
        if not self._primed:                              # A under the weights as they are now
            self.eval_graph.replay()
            self._primed = True
        self.train_graph.replay()
        self.eval_graph.replay()
        return self.stats[0], self.stats[1], self.stats[2], self.stats[3]

    @classmethod
    def for_nodes(cls, model, opt, X_real, X_img, label, train_index, val_index, criterion=None, warmup=3):
        """The epoch of node_epoch (same arguments; index tensors or NodeTargets, not boolean masks)."""
        criterion = criterion or torch.nn.NLLLoss()
        return cls(model, opt, lambda: model.trunk(X_real, X_img),
                   lambda p: _loss_acc(model.head(p), label, train_index, criterion),
                   lambda p: _loss_acc(model.head(p), label, val_index, criterion), warmup)

    @classmethod
    def for_links(cls, model, opt, X_real, X_img, train_index, y_train, val_index, y_val, warmup=3):
        """The epoch of link_epoch (same arguments)."""
        def stats(p, index, y):
            out = model.head(p, index)
            return F.nll_loss(out, y), (out.max(dim=1)[1] == y).float().mean()
        return cls(model, opt, lambda: model.trunk(X_real, X_img), lambda p: stats(p, train_index, y_train),
                   lambda p: stats(p, val_index, y_val), warmup)


class GraphedEpoch:
    """Replay an epoch body as one HIP graph.  The small reference configurations (a few thousand
    nodes) spend their epoch in ~100 kernel launches of a few microseconds each; captured once,
    the whole epoch is a single graph launch.  `body` is a zero-argument callable that runs the
    epoch on tensors that stay at fixed addresses (model parameters, features, masks) and returns
    tensors; the optimizer must be created with capturable=True."""

    def __init__(self, body, warmup=3):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):          # builds SpMM plans, autotunes GEMMs, primes the allocator
                body()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = body()

    def __call__(self):
        self.graph.replay()
        return self.out
