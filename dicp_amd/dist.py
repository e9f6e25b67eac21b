"""Multi-GPU harness: scan pairs are independent, so a batch shards across the GPUs of one
node with no collective inside the ICP loop; the only exchange is ONE all-gather of the
per-shard poses per call (RCCL over xGMI when the backend is "nccl"; 64 B per cloud, so it
is latency-bound -- SURVEY.md section 8e).  The reference has no distributed code.

One process per GPU (torchrun); every function here also works on the gloo backend with CPU
tensors, which is how tests/test_dist_gloo.py covers it without GPUs.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous split of `total` clouds: rank g gets [lo, hi); sizes differ by at most one."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_bounds(costs, world):
    """Contiguous split of a RAGGED batch by work instead of by count (balance by sum n_i * m_i):
    costs[i] = the work of cloud i (e.g. n_i * m_i).  Returns [(lo, hi)] per rank: rank g ends at the first cloud
    where the running cost reaches (g+1)/world of the total, so no rank exceeds its even share by more than one cloud."""
    costs = [float(c) for c in costs]
    total, n = sum(costs), len(costs)
    cuts, run, i = [0], 0.0, 0
    for g in range(1, int(world)):
        goal = total * g / world
        while i < n and run + 0.5 * costs[i] < goal:        # a cloud goes to the side of the cut its midpoint falls on
            run += costs[i]
            i += 1
        cuts.append(i)
    cuts.append(n)
    return [(cuts[g], cuts[g + 1]) for g in range(int(world))]


def shard(items, rank=None, world=None, bounds=None):
    """Slice a batched tensor (dim 0) or a per-cloud list down to this rank's clouds.
    bounds: optional [(lo, hi)] per rank (balanced_bounds) instead of the even split."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = bounds[rank] if bounds is not None else shard_bounds(len(items), rank, world)
    return items[lo:hi]


def _all_gather_rows(x, counts, group):
    """x (counts[rank], ...) on every rank -> (sum(counts), ...) in rank order.  Equal shards go straight into the result;
    unequal ones are padded to the largest for the collective and trimmed afterwards."""
    world, big = len(counts), max(counts)
    tail = tuple(x.shape[1:])
    if min(counts) == big:
        recv = torch.empty((world * big,) + tail, dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(recv, x.contiguous(), group=group)
        return recv
    send = torch.zeros((big,) + tail, dtype=x.dtype, device=x.device)
    send[:x.shape[0]] = x
    recv = torch.empty((world * big,) + tail, dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world, big) + tail)
    return torch.cat([recv[g, :counts[g]] for g in range(world)], dim=0)


class _GatherPoses(torch.autograd.Function):
    """all-gather with a backward.  grad_mode "slice": every rank holds the same loss on T_all (the usual replicated loss),
    so d loss / d T_local is this rank's slice of the incoming gradient -- no communication.  "reduce_scatter": ranks
    hold DIFFERENT losses on T_all and the total is meant: the slices of all ranks are summed (one all-reduce of
    B_total x 64 B; still latency-bound)."""

    @staticmethod
    def forward(ctx, T_local, counts, group, grad_mode):
        rank = dist.get_rank(group)
        ctx.lo, ctx.n = sum(counts[:rank]), counts[rank]
        ctx.group, ctx.grad_mode = group, grad_mode
        return _all_gather_rows(T_local.detach(), counts, group)

    @staticmethod
    def backward(ctx, g_all):
        if ctx.grad_mode == "reduce_scatter":
            g_all = g_all.contiguous().clone()
            dist.all_reduce(g_all, group=ctx.group)
        return g_all[ctx.lo:ctx.lo + ctx.n], None, None, None


def gather_poses(T_local, total=None, group=None, force=False, grad_mode="slice", counts=None):
    """All-gather per-shard poses (B_local,4,4) -> (B_total,4,4), in cloud order, on every rank.
    Shard sizes: `counts` (per rank) if given, else the even split of `total`, else exchanged (one small all-gather and
    a host sync).  Differentiable: a loss on the result sends its gradient back into this rank's T_local
    (_GatherPoses), from there through the ICP call to source / target -- which stay shard-local."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return T_local
    world = dist.get_world_size(group)
    if counts is None and total is not None:
        counts = [shard_bounds(total, g, world)[1] - shard_bounds(total, g, world)[0] for g in range(world)]
    if counts is None:
        sizes = torch.tensor([T_local.shape[0]], dtype=torch.int64, device=T_local.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    counts = [int(c) for c in counts]
    if T_local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError("gather_poses: this rank holds %d poses but its shard has %d clouds (an empty shard's ICP call "
                         "returns the reference's phony pair, ICP.py:328-346: slice T[:n_local] first, as icp_sharded does)"
                         % (T_local.shape[0], counts[dist.get_rank(group)]))
    if grad_mode not in ("slice", "reduce_scatter"):
        raise ValueError("grad_mode must be 'slice' or 'reduce_scatter'")
    return _GatherPoses.apply(T_local, counts, group, grad_mode)


def gather_poses_async(T_local, total=None, group=None, force=False, counts=None):
    """The same all-gather for a caller that does not differentiate through it and has other work to queue first (the backward of a
    shard-local loss): -> (T_all, work) with the collective in flight on the communicator's stream; work.wait() orders the current
    stream behind it (work is None when nothing was communicated).  Equal shards only (`total` or `counts`)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return T_local, None
    world = dist.get_world_size(group)
    if counts is None:
        if total is None:
            raise ValueError("gather_poses_async needs the shard sizes (total or counts): it does not exchange them")
        counts = [shard_bounds(total, g, world)[1] - shard_bounds(total, g, world)[0] for g in range(world)]
    counts = [int(c) for c in counts]
    if min(counts) != max(counts) or T_local.shape[0] != counts[dist.get_rank(group)]:
        raise ValueError("gather_poses_async: equal shards only, and this rank must hold its own (%s, this rank holds %d)" % (counts, T_local.shape[0]))
    x = T_local.detach().contiguous()
    recv = torch.empty((world * counts[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    work = dist.all_gather_into_tensor(recv, x, group=group, async_op=True)
    return recv, work


def icp_sharded(icp_fn, source, target, T_init, total=None, group=None, counts=None, grad_mode="slice", **kwargs):
    """Run `icp_fn(source, target, T_init, **kwargs)` on this rank's shard (the caller passes the shard, e.g. via
    shard()) and add "T_all": the poses of the whole batch on every rank.  An EMPTY shard (more ranks than clouds) is
    legal: the reference's batching answers it with one phony pair (ICP.py:328-346), which is dropped before the gather."""
    n_local = len(source)
    out = icp_fn(source, target, T_init, **kwargs)
    out["T_all"] = gather_poses(out["T"][:n_local], total=total, group=group, counts=counts, grad_mode=grad_mode)
    return out
