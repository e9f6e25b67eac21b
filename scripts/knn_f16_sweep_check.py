"""Matrix-core scoring inside the exact sorted sweep (dicp_knn_sweep with f16_image) against the VALU sweep: idx and spos must be identical."""
import sys
import torch
sys.path.insert(0, ".")
from dicp_amd import _lib, _ops
from dicp_amd.synthetic import make_pairs, make_scene_pairs
from dicp_amd.ICP import ICP

dev = "cuda"


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def case(name, src, tgt, pose=None, tgt_rows=None, src_rows=None):
    src, tgt = src.to(dev).contiguous(), tgt.to(dev).contiguous()
    N, n, _ = src.shape
    m = tgt.shape[1]
    frame = _ops.search_frame(tgt, tgt_rows=tgt_rows)
    sw = _ops.SweepIndex(tgt, frame=frame, tgt_rows=tgt_rows)
    ps = _ops.search_pose(pose.to(dev) if pose is not None else None, frame, N)
    qo = sw.query_order(src, ps, src_rows=src_rows)
    out = {}
    for mf in (False, True):
        idx = torch.full((N, n), -7, dtype=torch.int32, device=dev)
        spos = torch.full((N, n), -7, dtype=torch.int32, device=dev)
        sw.pair_shards.zero_()
        sw.knn(src, ps, qo, out=idx, cfg=2, spos=spos, src_rows=src_rows, mfma=mf)
        torch.cuda.synchronize()
        out[mf] = (idx, spos, float(sw.pairs.item()) / (float(N) * n * m))
    bad = int((out[False][0] != out[True][0]).sum()) + int((out[False][1] != out[True][1]).sum())
    again, scan = _ops.f16_counters(sw.img16, N, sw.tgs4.shape[1])
    tmp_i, tmp_s = torch.empty_like(out[True][0]), torch.empty_like(out[True][1])
    t_v = timed(lambda: sw.knn(src, ps, qo, out=tmp_i, cfg=2, spos=tmp_s, src_rows=src_rows))
    t_m = timed(lambda: sw.knn(src, ps, qo, out=tmp_i, cfg=2, spos=tmp_s, src_rows=src_rows, mfma=True))
    print("%-44s N=%4d n=%6d m=%6d  mismatches %7d  pairs %.2f %% / %.2f %%  valu %7.3f ms  mfma %7.3f ms (x%.2f)  pass 2: %.3f %%  scan: %.3f %%"
          % (name, N, n, m, bad, 100 * out[False][2], 100 * out[True][2], t_v, t_m, t_v / t_m, 100.0 * again / (N * n), 100.0 * scan / (N * n)), flush=True)
    return bad


bad = 0
for (N, n, m, seed) in ((3, 500, 600, 1), (8, 4096, 4096, 2), (64, 16384, 16384, 4), (256, 16384, 16384, 5), (16, 65536, 65536, 6), (5, 777, 3001, 7)):
    src, tgt = make_pairs(N, n, m, seed=seed)
    bad += case("random clouds, identity pose", src, tgt)
src, tgt = make_pairs(256, 16384, 16384, seed=11)
poses = {}
for K in (1, 2, 6):
    icp = ICP(icp_type="pt2pl", differentiable=False, max_iterations=K, tolerance=1e-12); icp.const_iter = True
    T = icp.icp(src.to(dev), tgt.to(dev), torch.eye(4, device=dev).repeat(256, 1, 1), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"]
    poses[K] = torch.cat((T[:, :3, :3].reshape(256, 9), T[:, :3, 3]), dim=1).contiguous()
    bad += case("random clouds, pose after %d iteration(s)" % K, src, tgt, pose=poses[K])
src, tgt = src[:64], tgt[:64]
s2, t2 = make_scene_pairs(64, 16384, 16384, seed=3)
bad += case("planar scenes", s2, t2)
far = tgt.clone(); far[:, :, :3] += torch.tensor([2500.0, -1200.0, 300.0]); fs = src + torch.tensor([2500.0, -1200.0, 300.0])
bad += case("clouds 2.8 km from the origin", fs[:16], far[:16])
dup = tgt.clone(); dup[:, 1::2] = dup[:, 0::2]
bad += case("every target twice (exact ties)", src[:16], dup[:16])
trip = tgt.clone(); trip[:, 1::3] = trip[:, 0::3][:, :trip[:, 1::3].shape[1]]; trip[:, 2::3] = trip[:, 0::3][:, :trip[:, 2::3].shape[1]]
bad += case("every target three times", src[:16], trip[:16])
near = tgt.clone(); near[:, 1::2, :3] = near[:, 0::2, :3] + 1e-4
bad += case("every target twice, 0.1 mm apart", src[:16], near[:16])
padded = tgt.clone(); padded[:, -300:] = float(src.max()) * 1000.0
rows = torch.full((64,), 16384 - 299, dtype=torch.int32, device=dev)
bad += case("reference pad rows (x1000) in the target", src[:16], padded[:16])
bad += case("the same with tgt_rows", src[:16], padded[:16], tgt_rows=rows[:16].contiguous())
outl = tgt.clone(); outl[:, 5, :3] = 1e6; outl[:, 77, :3] = -3e5
bad += case("two far outliers", src[:16], outl[:16])
plane = tgt.clone(); plane[:, :, 0] = 1.25
bad += case("all targets on one x plane", src[:8], plane[:8])
tiny = tgt.clone() * 1e-3
bad += case("cloud 2 cm across", src[:8] * 1e-3, tiny[:8])
qfar = src.clone(); qfar[:, ::7] *= 40.0
bad += case("queries far outside the cloud", qfar[:8], tgt[:8])
nanr = tgt.clone(); nanr[:, 100:110, :3] = float("nan"); nanr[:, 200, 0] = float("inf")
bad += case("non-finite target rows", src[:8], nanr[:8])
sr = torch.tensor([16384, 100, 5000, 1, 16000, 9999, 64, 129], dtype=torch.int32, device=dev)
bad += case("ragged sources", src[:8], tgt[:8], src_rows=sr)
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
