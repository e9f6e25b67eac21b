"""N>1 path on CPU: world_size-2 gloo processes shard a batch, run their shard, all-gather poses.
The per-shard compute is the oracle here (no GPU in this suite); the sharding + collective
code is exactly what bench.py and users run over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dicp_amd import dist as ddist
from dicp_amd.synthetic import make_pairs
from oracle import dicp_oracle as O

KW = dict(icp_type="pt2pl", differentiable=True, max_iterations=3, tolerance=1e-12, trim_dist=5.0,
          loss_fn={"name": "huber", "metric": 1.0}, dim=3, const_iter=True)


def oracle_icp(source, target, T_init, **kw):
    return O.icp_batched(source, target, T_init, torch.ones(source.shape[:2], dtype=source.dtype), **kw)


def batching_icp(source, target, T_init, **kw):
    """icp_fn with the product's batching semantics for an EMPTY shard: the reference answers empty input with one phony
    pair of zero weight and identity pose (ICP.py:328-346; dicp_amd/ICP.py:_batch), i.e. T comes back (1,4,4), not (0,4,4)."""
    if len(source) == 0:
        return {"T": torch.eye(4, dtype=T_init.dtype).unsqueeze(0)}
    return oracle_icp(source, target, T_init, **kw)


def grad_worker(rank, world, port, total, out_dir, grad_mode):
    """A loss on the GATHERED poses sends its gradient back through the gather into this rank's shard."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
        T0 = torch.eye(4, dtype=torch.float64).repeat(total, 1, 1)
        cot = torch.rand((total, 4, 4), generator=torch.Generator().manual_seed(7), dtype=torch.float64)
        s = ddist.shard(src).clone().requires_grad_(True)
        t = ddist.shard(tgt).clone().requires_grad_(True)
        out = ddist.icp_sharded(batching_icp, s, t, ddist.shard(T0), total=total, grad_mode=grad_mode, **KW)
        scale = 1.0 if grad_mode == "slice" else (rank + 1.0)      # reduce_scatter: the ranks hold different losses
        loss = (out["T_all"] * cot * scale).sum()
        if loss.requires_grad:                                      # (an empty shard has nothing to differentiate)
            loss.backward()
        lo, hi = ddist.shard_bounds(total, rank, world)
        np.save(os.path.join(out_dir, "gs_%d.npy" % rank), s.grad.numpy() if hi > lo else np.zeros((0, 96, 3)))
        np.save(os.path.join(out_dir, "gt_%d.npy" % rank), t.grad.numpy() if hi > lo else np.zeros((0, 128, 6)))
        np.save(os.path.join(out_dir, "T_all_%d.npy" % rank), out["T_all"].detach().numpy())
    finally:
        dist.destroy_process_group()


def worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
        T0 = torch.eye(4, dtype=torch.float64).repeat(total, 1, 1)
        s, t, T = ddist.shard(src), ddist.shard(tgt), ddist.shard(T0)
        out = ddist.icp_sharded(batching_icp, s, t, T, total=total, **KW)
        out2 = ddist.gather_poses(out["T"][:len(s)])       # size-discovery path
        if len(s) == 0:                                    # the phony pair of an empty shard must not be gathered
            with pytest.raises(ValueError):
                ddist.gather_poses(out["T"], total=total)
        assert torch.equal(out["T_all"], out2)
        np.save(os.path.join(out_dir, "T_all_%d.npy" % rank), out["T_all"].detach().numpy())
        np.save(os.path.join(out_dir, "n_local_%d.npy" % rank), np.array(s.shape[0]))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds():
    assert [ddist.shard_bounds(5, g, 2) for g in range(2)] == [(0, 3), (3, 5)]
    assert [ddist.shard_bounds(2048, g, 8) for g in range(8)] == [(256 * g, 256 * (g + 1)) for g in range(8)]
    assert [ddist.shard_bounds(2, g, 4) for g in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert ddist.gather_poses(torch.eye(4).repeat(3, 1, 1)).shape == (3, 4, 4)    # no process group: identity


@pytest.mark.parametrize("total,world,sizes", [(5, 2, [3, 2]), (4, 2, [2, 2]), (5, 3, [2, 2, 1]), (2, 3, [1, 1, 0]),
                                               (11, 8, [2, 2, 2, 1, 1, 1, 1, 1])])      # BASELINE configs[4]'s rank count, unequal shards
def test_gloo_ranks_match_single_process(tmp_path, total, world, sizes):
    mp.spawn(worker, args=(world, free_port(), total, str(tmp_path)), nprocs=world, join=True)
    src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
    ref = oracle_icp(src, tgt, torch.eye(4, dtype=torch.float64).repeat(total, 1, 1), **KW)["T"].numpy()
    a = np.load(tmp_path / "T_all_0.npy")
    assert [int(np.load(tmp_path / ("n_local_%d.npy" % r))) for r in range(world)] == sizes
    for r in range(1, world):                            # every rank holds the whole batch's poses, in cloud order
        np.testing.assert_array_equal(a, np.load(tmp_path / ("T_all_%d.npy" % r)))
    np.testing.assert_allclose(a, ref, rtol=0, atol=1e-11)       # (the oracle's batched CPU kernels round differently at other batch sizes: 2e-12)


@pytest.mark.parametrize("total,world,grad_mode", [(5, 2, "slice"), (2, 3, "slice"), (4, 2, "reduce_scatter")])
def test_gloo_gradient_through_the_pose_gather(tmp_path, total, world, grad_mode):
    """backward of the all-gather = this rank's slice (replicated loss), or the sum over ranks of the slices
    (reduce_scatter) -- a loss on T_all must give the single-process source / target gradients, shard by shard."""
    mp.spawn(grad_worker, args=(world, free_port(), total, str(tmp_path), grad_mode), nprocs=world, join=True)
    src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
    s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
    cot = torch.rand((total, 4, 4), generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    ref = oracle_icp(s, t, torch.eye(4, dtype=torch.float64).repeat(total, 1, 1), **KW)
    factor = 1.0 if grad_mode == "slice" else sum(r + 1.0 for r in range(world))
    (ref["T"] * cot * factor).sum().backward()
    gs = np.concatenate([np.load(tmp_path / ("gs_%d.npy" % r)) for r in range(world)])
    gt = np.concatenate([np.load(tmp_path / ("gt_%d.npy" % r)) for r in range(world)])
    np.testing.assert_allclose(gs, s.grad.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(gt, t.grad.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(np.load(tmp_path / "T_all_0.npy"), ref["T"].detach().numpy(), rtol=0, atol=1e-12)


def test_balanced_bounds():
    """Ragged batches shard by work (sum n_i * m_i), contiguously."""
    assert ddist.balanced_bounds([1] * 8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    b = ddist.balanced_bounds([100, 1, 1, 1, 1, 100], 2)
    assert b[0][0] == 0 and b[-1][1] == 6 and b[0][1] == b[1][0]
    loads = [sum([100, 1, 1, 1, 1, 100][lo:hi]) for lo, hi in b]
    assert max(loads) <= 104
    costs = list(np.random.RandomState(0).randint(1, 1000, size=200))
    for world in (2, 3, 8):
        bb = ddist.balanced_bounds(costs, world)
        assert bb[0][0] == 0 and bb[-1][1] == 200 and all(bb[g][1] == bb[g + 1][0] for g in range(world - 1))
        loads = [sum(costs[lo:hi]) for lo, hi in bb]
        assert max(loads) <= sum(costs) / world + max(costs)
    assert ddist.balanced_bounds([5, 5], 4)[-1][1] == 2          # more ranks than clouds: empty shards are legal
    assert ddist.shard(list(range(6)), rank=1, world=2, bounds=[(0, 4), (4, 6)]) == [4, 5]


def async_worker(rank, world, port, total, out_dir):
    """gather_poses_async: the collective is issued, other work is queued, wait() -> the same poses as the blocking gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        lo, hi = ddist.shard_bounds(total, rank, world)
        T_local = torch.rand((hi - lo, 4, 4), generator=g, dtype=torch.float64)
        T_all, work = ddist.gather_poses_async(T_local, total=total)
        other = (T_local * 2.0).sum()                    # (what a caller queues while the collective is in flight)
        assert work is not None
        work.wait()
        ref = ddist.gather_poses(T_local, total=total)
        assert torch.equal(T_all, ref) and float(other) == float((T_local * 2.0).sum())
        with pytest.raises(ValueError):
            ddist.gather_poses_async(T_local)            # no sizes given: it does not exchange them
        np.save(os.path.join(out_dir, "async_%d.npy" % rank), T_all.numpy())
    finally:
        dist.destroy_process_group()


def test_gloo_async_pose_gather(tmp_path):
    world, total = 2, 6
    mp.spawn(async_worker, args=(world, free_port(), total, str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(str(tmp_path / ("async_%d.npy" % r))) for r in range(world))
    np.testing.assert_array_equal(a, b)
    assert a.shape == (total, 4, 4)
