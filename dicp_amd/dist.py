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


def shard(items, rank=None, world=None):
    """Slice a batched tensor (dim 0) or a per-cloud list down to this rank's clouds."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def gather_poses(T_local, total=None, group=None, force=False):
    """All-gather per-shard poses (B_local,4,4) -> (B_total,4,4), in cloud order, on every rank.
    Shards may differ in size by one (shard_bounds): they are padded to the largest for the
    collective and trimmed afterwards.  The result is detached (poses are gathered for the
    consumer's loss/logging; gradients w.r.t. source/target stay shard-local)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return T_local.detach()
    world = dist.get_world_size(group)
    if total is None:
        sizes = torch.tensor([T_local.shape[0]], dtype=torch.int64, device=T_local.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    else:
        counts = [shard_bounds(total, g, world)[1] - shard_bounds(total, g, world)[0] for g in range(world)]
    big = max(counts)
    if min(counts) == big:              # equal shards (the usual case): gather straight into the result, no padding, no trimming
        recv = torch.empty((world * big, 4, 4), dtype=T_local.dtype, device=T_local.device)
        dist.all_gather_into_tensor(recv, T_local.detach().contiguous(), group=group)
        return recv
    send = torch.zeros((big, 4, 4), dtype=T_local.dtype, device=T_local.device)
    send[:T_local.shape[0]] = T_local.detach()
    recv = torch.empty((world * big, 4, 4), dtype=T_local.dtype, device=T_local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, big, 4, 4)
    return torch.cat([recv[g, :counts[g]] for g in range(world)], dim=0)


def icp_sharded(icp_fn, source, target, T_init, total=None, group=None, **kwargs):
    """Run `icp_fn(source, target, T_init, **kwargs)` on this rank's shard (the caller passes the
    shard, e.g. via shard()) and add "T_all": the poses of the whole batch on every rank."""
    out = icp_fn(source, target, T_init, **kwargs)
    out["T_all"] = gather_poses(out["T"], total=total, group=group)
    return out
