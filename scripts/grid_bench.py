"""Grid kNN vs sorted sweep vs brute force: index equality and kernel time, standalone, at several poses.
    python scripts/grid_bench.py [B] [n]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from dicp_amd import _lib, _ops
from dicp_amd.synthetic import make_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = "cuda"
src, tgt = make_pairs(B, n, n, seed=3)
src, tgt = src.to(dev), tgt.to(dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


ctr = _ops.cloud_center(tgt)
t_build = timeit(lambda: _ops.GridIndex(tgt, sorted_rows=True, center=ctr))
t_sweepb = timeit(lambda: _ops.SweepIndex(tgt, sorted_rows=True, center=ctr))
print("index build: grid %.3f ms   sweep %.3f ms" % (t_build, t_sweepb))
grid = _ops.GridIndex(tgt, sorted_rows=True, center=ctr)
sw = _ops.SweepIndex(tgt, sorted_rows=True, center=ctr)
gi = grid.ginfo[0].cpu()
print("grid of cloud 0: G =", gi[9:12].tolist(), "cells", int(gi[12]), "h =", gi[6:9].tolist())
occ = (grid.cell_start[0, 1:int(gi[12]) + 1] - grid.cell_start[0, :int(gi[12])])
print("cell occupancy: mean %.2f max %d empty %.1f %%" % (occ.float().mean(), occ.max(), 100.0 * float((occ == 0).float().mean())))
tgt4 = _ops.pack_target(tgt, ctr)
# poses: identity (iteration 0: source displaced by up to 0.05 rad / 0.3 m) and near the solution (source := its target picks + noise)
near = tgt[:, :, :3] + 0.01 * torch.randn((B, n, 3), device=dev)
for name, q in (("iteration 0 (far)", src), ("near the pose", near)):
    ps = _ops.search_pose(None, ctr, B)
    brute = _ops.knn(q, ps, tgt4, n, _lib.KNN_VALU)
    got = grid.knn(q, ps)
    qo = sw.query_order(q, ps)
    got_s = sw.knn(q, ps, qo)
    print("%-18s grid == brute: %s (%d differ)   sweep == brute: %s" % (name, torch.equal(got, brute), int((got != brute).sum()), torch.equal(got_s, brute)))
    grid.pair_shards.zero_(); sw.pair_shards.zero_()
    grid.knn(q, ps); sw.knn(q, ps, qo); torch.cuda.synchronize()
    pg, psw = float(grid.pairs.item()) / (B * n), float(sw.pairs.item()) / (B * n)
    spos = torch.empty((B, n), dtype=torch.int32, device=dev)
    tg = timeit(lambda: grid.knn(q, ps, spos=spos))
    ts = timeit(lambda: sw.knn(q, ps, qo, spos=spos))
    tq = timeit(lambda: sw.query_order(q, ps))
    print("%-18s grid %.3f ms (%.1f pairs/query)   sweep %.3f ms (%.1f pairs/query) + query order %.3f ms" % (name, tg, pg, ts, psw, tq))

# the same searches with the queries in CELL order (neighbouring lanes then read the same lines)
gi_all = grid.ginfo
for name, q in (("iteration 0 (far)", src), ("near the pose", near)):
    xq = q - ctr[:, None, :]
    cc = [torch.clamp(torch.floor((xq[:, :, a] - gi_all[:, a:a + 1]) * gi_all[:, 3 + a:4 + a]), min=0).minimum(gi_all[:, 9 + a:10 + a] - 1).long() for a in range(3)]
    cid = (cc[2] * gi_all[:, 10:11].long() + cc[1]) * gi_all[:, 9:10].long() + cc[0]
    order = torch.argsort(cid, dim=1)
    qs = torch.gather(q, 1, order.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ps = _ops.search_pose(None, ctr, B)
    ref = _ops.knn(qs, ps, tgt4, n, _lib.KNN_VALU)
    assert torch.equal(grid.knn(qs, ps), ref)
    print("%-18s grid, queries in cell order: %.3f ms" % (name, timeit(lambda: grid.knn(qs, ps))))
