"""The exact sorted sweep at the benchmark shape, VALU and matrix-core scoring, under the pose after POSE_ITERS ICP iterations (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib, _ops
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs, make_scene_pairs, make_independent_pairs

B = int(os.environ.get("B", 256)); n = int(os.environ.get("NPTS", 16384)); reps = int(os.environ.get("REPS", 5)); it = int(os.environ.get("POSE_ITERS", 0))
forms = os.environ.get("FORMS", "valu,mfma").split(",")
chunk = min(B, 64)
gen = {"pairs": make_pairs, "scene": make_scene_pairs, "indep": lambda *a, **k: make_independent_pairs(*a, ragged=False, **k)}[os.environ.get("GEN", "pairs")]
parts = [gen(chunk, n, n, seed=3 + i) for i in range(B // chunk)]
src = torch.cat([p[0] for p in parts]).cuda(); tgt = torch.cat([p[1] for p in parts]).cuda()
del parts
pose = None
if it:
    icp = ICP(icp_type="pt2pl", differentiable=False, max_iterations=it, tolerance=1e-12); icp.const_iter = True
    T = icp.icp(src, tgt, torch.eye(4, device="cuda").repeat(B, 1, 1), trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})["T"]
    pose = torch.cat((T[:, :3, :3].reshape(B, 9), T[:, :3, 3]), dim=1).contiguous()
frame = _ops.search_frame(tgt, src=src, T_init=torch.eye(4, device="cuda").repeat(B, 1, 1).contiguous())
sw = _ops.SweepIndex(tgt, frame=frame)
ps = _ops.search_pose(pose, frame, B)
qo = sw.query_order(src, ps)
idx = torch.empty((B, n), dtype=torch.int32, device="cuda"); spos = torch.empty_like(idx)
res = {}
for form in forms:
    ts = []
    for r in range(reps + 1):
        sw.pair_shards.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sw.knn(src, ps, qo, out=idx, cfg=2, spos=spos, mfma=(form == "mfma")); b.record(); torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    res[form] = idx.clone()
    ts.sort()
    extra = ""
    if form == "mfma":
        again, scan = _ops.f16_counters(sw.img16, B, sw.tgs4.shape[1])
        extra = "  second filter pass %.2f %% of the queries, exact scan %.3f %%" % (100.0 * again / (reps + 1) / (B * n), 100.0 * scan / (reps + 1) / (B * n))
    print("%-5s %s sweep B=%d n=m=%d pose after %d iterations: median %.3f ms  min %.3f ms   pairs scored %.2f %%%s" % (form, os.environ.get("GEN", "pairs"), B, n, it, ts[len(ts) // 2], ts[0], 100.0 * sw.pairs.item() / (float(B) * n * n), extra))
if len(res) == 2:
    print("mismatches:", int((res["valu"] != res["mfma"]).sum()))
