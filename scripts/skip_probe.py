"""Where a cloud's reverse sweep ends as a function of bwd_skip_eps, for two losses (T.sum(); T.sum() + 1e-3 |pc|^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd.ICP import ICP
from dicp_amd.synthetic import make_pairs
N, n, K = 40, 16384, 12
src, tgt = make_pairs(N, n, n, seed=61)
for name, lossf in (("T.sum()", lambda o: o["T"].sum()), ("T.sum() + 1e-3 |pc|^2", lambda o: o["T"].sum() + (o["pc"] ** 2).sum() * 1e-3), ("1e-3 |pc|^2", lambda o: (o["pc"] ** 2).sum() * 1e-3)):
    ref = None
    for eps in (0.0, 2.0 ** -22, 1e-4, 1e-2, 1.0, 100.0):
        icp = ICP(icp_type="pt2pl", differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True; icp.bwd_skip_eps = eps
        S, Tg = src.cuda().requires_grad_(True), tgt.cuda().requires_grad_(True)
        Ti = torch.eye(4).repeat(N, 1, 1).cuda().requires_grad_(True)
        out = icp.icp(S, Tg, Ti, trim_dist=5.0, loss_fn={"name": "huber", "metric": 1.0})
        lossf(out).backward()
        g = (S.grad.clone(), Tg.grad.clone(), Ti.grad.clone())
        if ref is None:
            ref = g
            print(name, " |delta| by iteration (cloud 0):", [float("%.2e" % v) for v in out["deltas"][0, :, :, 0].norm(dim=1).tolist()])
            print("    gradient sizes: source %.3e target %.3e T_init %.3e" % tuple(float(x.abs().max()) for x in g))
            continue
        live = icp.knn_stats["bwd_live"][:K].tolist()
        print("    eps %.1e: live %s  rel err source %.1e target %.1e T_init %.1e" % (eps, live, *(float((a - b).abs().max() / max(1e-30, float(b.abs().max()))) for a, b in zip(g, ref))))
