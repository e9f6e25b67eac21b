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


def worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
        T0 = torch.eye(4, dtype=torch.float64).repeat(total, 1, 1)
        s, t, T = ddist.shard(src), ddist.shard(tgt), ddist.shard(T0)
        out = ddist.icp_sharded(oracle_icp, s, t, T, total=total, **KW)
        out2 = ddist.gather_poses(out["T"])                # size-discovery path
        assert torch.equal(out["T_all"], out2)
        np.save(os.path.join(out_dir, "T_all_%d.npy" % rank), out["T_all"].numpy())
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


@pytest.mark.parametrize("total,world,sizes", [(5, 2, [3, 2]), (4, 2, [2, 2]), (5, 3, [2, 2, 1]), (2, 3, [1, 1, 0])])
def test_gloo_ranks_match_single_process(tmp_path, total, world, sizes):
    mp.spawn(worker, args=(world, free_port(), total, str(tmp_path)), nprocs=world, join=True)
    src, tgt = make_pairs(total, 96, 128, seed=4, dtype=torch.float64)
    ref = oracle_icp(src, tgt, torch.eye(4, dtype=torch.float64).repeat(total, 1, 1), **KW)["T"].numpy()
    a = np.load(tmp_path / "T_all_0.npy")
    assert [int(np.load(tmp_path / ("n_local_%d.npy" % r))) for r in range(world)] == sizes
    for r in range(1, world):                            # every rank holds the whole batch's poses, in cloud order
        np.testing.assert_array_equal(a, np.load(tmp_path / ("T_all_%d.npy" % r)))
    np.testing.assert_allclose(a, ref, rtol=0, atol=1e-12)
