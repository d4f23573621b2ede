"""Locality-preserving relabelling for the row partition (no counterpart in the reference, which runs on one device).

The row partition (magnet_amd.dist.RowPartition) gives rank p a contiguous block of vertex ids and exchanges, per
propagation order, every remote feature row the block references.  How many rows that is depends only on how the
vertices are numbered: on a graph whose neighbours have nearby ids a few boundary rows travel, on a randomly
numbered one nearly all of them.  This module renumbers the vertices before the Laplacian is built:

  parts = balanced_label_propagation(src, dst, n, world)     # one part id per vertex
  new_of_old, old_of_new, bounds = order_from_parts(parts, world)
  src2, dst2 = new_of_old[src], new_of_old[dst]              # build the Laplacian / row blocks from these
  x2 = x[old_of_new]                                         # features, labels, masks follow the vertices
  ... out[new_of_old] gives the result in the caller's numbering

The Laplacian of the renumbered graph is P L P^T: results agree with the unpermuted run up to fp32 summation
order (a row's columns are visited in a different order).

Plain tensor code (runs on the device the edges live on, or on CPU tensors); it is set-up work, executed once per
graph.  Any other partitioner's output can be passed to order_from_parts instead."""
import torch


def _hash01(ids, salt):
    """Deterministic pseudo-random numbers in [0, 1) per vertex id and round (no generator state: every rank and
    every device computes the same decisions)."""
    x = (ids.to(torch.int64) * 0x9E3779B1 + (salt + 1) * 0x85EBCA77) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & 0xFFFFFFFF
    x = ((x ^ (x >> 12)) * 0x297A2D39) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    return x.to(torch.float64) / 4294967296.0


def _sum_by(keys, values, n_keys):
    """sum of `values` per key.  With a handful of keys an index_add_ would send every element's atomic to the same few
    addresses (they serialise on a GPU); masked sums are reductions instead."""
    if n_keys <= 16:
        return torch.stack([(values * (keys == k)).sum() for k in range(n_keys)])
    return torch.zeros(n_keys, dtype=values.dtype, device=values.device).index_add_(0, keys, values)


def _propagate(a, b, weight, n, fracs, iters, slack, noise, damping, salt, init=None):
    """Size-constrained synchronous label propagation on the symmetric edge list (a, b): part p may carry at most
    slack * fracs[p] of the total vertex weight."""
    dev = a.device
    n_parts = len(fracs)
    total = float(weight.sum())
    cap = torch.tensor([slack * total * f for f in fracs], dtype=torch.float64, device=dev)
    if init is None:                                   # contiguous id blocks of the wanted weights
        edges = torch.cumsum(torch.tensor(fracs, dtype=torch.float64, device=dev), 0) * total
        part = torch.searchsorted(edges, torch.cumsum(weight, 0) - 0.5 * weight).clamp_(max=n_parts - 1)
    else:
        part = init.long().clone()
    ids = torch.arange(n, device=dev)
    cols = torch.arange(n_parts, device=dev).unsqueeze(0)
    spread = torch.sqrt(weight.to(torch.float32)).unsqueeze(1)
    idle = 0
    for it in range(iters):
        votes = torch.bincount(a * n_parts + part[b], minlength=n * n_parts).view(n, n_parts).to(torch.float32)
        temp = noise * max(0.0, 1.0 - it / (0.7 * iters))
        if temp > 0:
            # annealing: a vertex whose neighbours are split evenly between two parts has no reason to move, and a part
            # holding halves of two communities stays that way; decaying noise on the votes breaks such ties early on
            votes = votes + temp * spread * (_hash01(ids.unsqueeze(1) * n_parts + cols, salt + it).to(torch.float32) - 0.5)
        here = votes.gather(1, part.unsqueeze(1)).squeeze(1)
        best_votes, best = votes.max(dim=1)
        wants = (best_votes > here) & (best != part)
        load = _sum_by(part, weight, n_parts)
        # demand[p, q]: weight that wants to go from part p to part q
        asking_w = torch.where(wants, weight, torch.zeros_like(weight))
        if n_parts <= 4:
            demand = _sum_by(part * n_parts + best, asking_w, n_parts * n_parts).view(n_parts, n_parts)
        else:
            demand = torch.stack([_sum_by(best, asking_w * (part == p), n_parts) for p in range(n_parts)])
        # admitted: what the opposite direction gives back (a swap leaves both loads as they were), plus a share of the
        # room the target has left
        room = (cap - load).clamp_(min=0.0)
        asking = demand.sum(0)
        share = torch.where(asking > 0, room / asking.clamp(min=1e-30), torch.zeros_like(room)).clamp_(max=1.0)
        admitted = torch.minimum(demand, demand.t() + demand * share.unsqueeze(0))
        frac = torch.where(demand > 0, admitted / demand.clamp(min=1e-30), torch.zeros_like(demand))
        move = wants & (_hash01(ids, salt + 7919 * (it + 1)) < damping * frac.view(-1)[part * n_parts + best])
        if bool(move.any()):
            part = torch.where(move, best, part)
            idle = 0
        else:
            idle += 1
            if temp == 0 and idle >= 2:
                break
    return part


def balanced_label_propagation(src, dst, n, n_parts, iters=40, slack=1.05, row_cost=2, damping=0.7, noise=2.0, polish=8):
    """One part id in [0, n_parts) per vertex such that few edges cross parts and every part carries about its share
    of the work, at most `slack` times it (vertex weight = degree + row_cost: what a row of the Laplacian costs a
    rank).  Recursive bisection: the vertex set is cut in two (sized for the numbers of parts either side will hold),
    each side again, until n_parts sets remain; every cut is a size-constrained synchronous label propagation with
    two labels.  In a round every vertex names the side most of its neighbours (in either direction: the
    Laplacian's support is symmetric) are on; movers are admitted at random up to what the opposite direction
    gives back (a swap changes neither load) plus a share of the room the target has left, damped, which breaks the
    two-cycles synchronous propagation is prone to; decaying noise on the votes lets communities that start out
    split evenly settle on one side.  A few rounds over all n_parts labels at once polish the result."""
    dev = src.device
    src, dst = src.long(), dst.long()
    keep = src != dst
    a = torch.cat([src[keep], dst[keep]])
    b = torch.cat([dst[keep], src[keep]])
    weight = torch.bincount(a, minlength=n).to(torch.float64) + float(row_cost)
    levels = max(1, (n_parts - 1).bit_length())
    cut_slack = slack ** (1.0 / (levels + 1))
    part = torch.zeros(n, dtype=torch.int64, device=dev)
    todo = [(torch.arange(n, device=dev), a, b, 0, n_parts, 1)]      # (vertices, their edges in local ids, first part, parts, salt)
    while todo:
        verts, ea, eb, first, k, salt = todo.pop()
        if k == 1 or verts.numel() == 0:
            part[verts] = first
            continue
        k0 = k // 2
        m = verts.numel()
        side = _propagate(ea, eb, weight[verts], m, [k0 / k, (k - k0) / k], iters, cut_slack, noise, damping, salt * 104729)
        local = torch.empty(m, dtype=torch.int64, device=dev)
        for s_id, (f0, kk) in enumerate(((first, k0), (first + k0, k - k0))):
            sel = side == s_id
            sub = verts[sel]
            local[sel] = torch.arange(sub.numel(), device=dev)
            both = sel[ea] & sel[eb]
            todo.append((sub, ea[both], eb[both], f0, kk, 2 * salt + s_id))
        # local ids were written per side above; the edge lists appended before the second side's ids existed are
        # renumbered here, once both sides are known
        for j in (-2, -1):
            sub, sa, sb, f0, kk, sl = todo[j]
            todo[j] = (sub, local[sa], local[sb], f0, kk, sl)
    if polish > 0 and n_parts > 2:
        part = _propagate(a, b, weight, n, [1.0 / n_parts] * n_parts, polish, slack, 0.0, damping, 31337, init=part)
    return part


def order_from_parts(parts, n_parts=None):
    """Vertices grouped by part, original order kept inside a part.
    -> (new_of_old [n] int64, old_of_new [n] int64, bounds: n_parts+1 row boundaries of the parts)."""
    parts = parts.long()
    n_parts = int(parts.max()) + 1 if n_parts is None else n_parts
    old_of_new = torch.sort(parts, stable=True).indices
    new_of_old = torch.empty_like(old_of_new)
    new_of_old[old_of_new] = torch.arange(parts.numel(), device=parts.device)
    counts = torch.bincount(parts, minlength=n_parts)
    bounds = [0] + [int(v) for v in torch.cumsum(counts, 0).tolist()]
    return new_of_old, old_of_new, bounds


def cut_fraction(src, dst, parts):
    """Share of the edges whose endpoints lie in different parts."""
    p = parts.long()
    s, d = src.long(), dst.long()
    return float((p[s] != p[d]).double().mean()) if s.numel() else 0.0
