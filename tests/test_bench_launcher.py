"""bench.py's N > 1 launcher path on two gloo CPU ranks (no GPU, no RCCL): rank environment, process group, weak-scaling
cloud seeds (rank g draws clouds g*B ..), barriers, max-over-ranks timing, the pose all-gather and the ONE JSON line of rank 0.
The compute is a stand-in handed in through bench.main's `make_icp` seam (the oracle: this is a test, bench.py itself never
touches it outside its cpu_baseline leg); everything else is the code the driver launches with torch.distributed.run."""
import hashlib
import json
import os
import socket

import torch
import torch.multiprocessing as mp

B, N_PTS, K = 2, 96, 2


class OracleICP:
    """Same call surface as dicp_amd.ICP.ICP for what bench.py uses; records which clouds it was handed."""

    def __init__(self, log_dir, **kw):
        self.kw, self.log_dir = kw, log_dir
        self.max_iterations, self.const_iter, self.knn_variant, self._tuning = kw["max_iterations"], True, 0, {"timing_events": None}
        self.knn_stats = {}

    def icp(self, source, target, T_init, trim_dist=None, loss_fn=None, dim=3):
        from oracle import dicp_oracle as O
        rank = int(os.environ["RANK"])
        with open(os.path.join(self.log_dir, "clouds_%d.txt" % rank), "w") as f:
            f.write(hashlib.sha256(source.detach().numpy().tobytes()).hexdigest())
        return O.icp_batched(source, target, T_init, torch.ones(source.shape[:2], dtype=source.dtype), icp_type=self.kw["icp_type"],
                             differentiable=True, max_iterations=int(self.max_iterations), tolerance=self.kw["tolerance"],
                             trim_dist=trim_dist, loss_fn=loss_fn, dim=dim, const_iter=bool(self.const_iter), tanh_steepness=5.0)


def worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2 if world <= 2 else 1)
    import bench
    from dicp_amd import dist as ddist
    gathers, real = [], ddist.gather_poses_async

    def counted(*a, **kw):              # the data path's one collective: counted per call of bench.run_call
        gathers.append(1)
        return real(*a, **kw)
    ddist.gather_poses_async = counted
    calls, run = [], bench.run_call

    def counted_call(*a, **kw):
        before = len(gathers)
        res = run(*a, **kw)
        calls.append(len(gathers) - before)
        return res
    bench.run_call = counted_call
    lines = []
    rc = bench.main(["--gpus", str(world), "--steps", str(K), "--warmup", "1", "--batch", str(B), "--points", str(N_PTS), "--reps", "5",
                     "--no-cpu-baseline", "--no-extra-legs"],
                    make_icp=lambda **kw: OracleICP(out_dir, **kw), device=torch.device("cpu"), backend="gloo", emit=lines.append)
    assert rc == 0
    assert calls and set(calls) == {1}, calls                       # ONE pose all-gather per icp() + backward(), nothing else on the data path
    with open(os.path.join(out_dir, "line_%d.json" % rank), "w") as f:
        f.write("".join(lines))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_rank_dry_run(tmp_path):
    from dicp_amd.synthetic import make_pairs
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "line_1.json").read() == ""                  # only rank 0 prints
    text = open(tmp_path / "line_0.json").read()
    assert text.count("\n") == 1                                        # ONE line
    line = json.loads(text)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == K and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["unit"] == "cloud-iterations/s"
    assert line["timed_calls"] == 5 and len(line["call_ms"]) == 5 and line["config"]["K"] == K
    med = sorted(line["call_ms"])[2]
    assert abs(line["value"] - world * B * K / (med * 1e-3)) <= 1e-4 * line["value"]        # whole-job aggregate over both ranks (call_ms is rounded to 0.1 us)
    assert abs(line["ms_per_step"] - med / K) < 1e-4
    assert "x2" in line["config"]["parallelism"] and line["finite"] is True
    assert len(line["call_ms_by_rank"]) == world and max(line["call_ms_by_rank"]) <= med + 1e-3     # every rank's own time of the median call
    assert line["pose_allgather_ms"] is not None and line["pose_allgather_ms"] > 0.0            # the collective, timed alone
    # weak scaling with global cloud seeds: rank g's clouds are clouds g*B .. (g+1)*B of the single-process batch
    for g in range(world):
        want, _ = make_pairs(B, N_PTS, N_PTS, seed=3, dtype=torch.float32, first=g * B)
        assert open(tmp_path / ("clouds_%d.txt" % g)).read() == hashlib.sha256(want.numpy().tobytes()).hexdigest()


def test_bench_eight_rank_dry_run(tmp_path):
    """BASELINE configs[4]'s launch shape -- 8 ranks of one node -- on gloo: the line says 8 ranks were seen, every rank drew its own clouds
    (first = rank * B), `value` is the whole job's, and each timed call made exactly one all-gather (asserted inside the workers)."""
    from dicp_amd.synthetic import make_pairs
    world = 8
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    line = json.loads(open(tmp_path / "line_0.json").read())
    assert all(open(tmp_path / ("line_%d.json" % r)).read() == "" for r in range(1, world))
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and "x8" in line["config"]["parallelism"] and line["scaling"] == "weak"
    med = sorted(line["call_ms"])[2]
    assert abs(line["value"] - world * B * K / (med * 1e-3)) <= 1e-4 * line["value"]
    assert len(line["call_ms_by_rank"]) == world
    seen = set()
    for g in range(world):
        want, _ = make_pairs(B, N_PTS, N_PTS, seed=3, dtype=torch.float32, first=g * B)
        digest = open(tmp_path / ("clouds_%d.txt" % g)).read()
        assert digest == hashlib.sha256(want.numpy().tobytes()).hexdigest()
        seen.add(digest)
    assert len(seen) == world                                           # eight different shards


def test_bench_refuses_a_rank_count_the_node_does_not_have(monkeypatch):
    """--gpus N on a node with fewer devices, or a process group of another size, stops before anything is timed."""
    import pytest
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="launch with"):
        bench.main(["--gpus", "8"], device=torch.device("cpu"))
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="this node shows 1 device"):
        bench.main(["--gpus", "2"], device=None)
