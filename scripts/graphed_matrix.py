"""Captured steps (graphed_icp_step) of other call forms against the eager call, replays behind synchronisations and eager work: deterministic mode, point-to-point,
float64, certificates at small sizes, tolerance refused.  usage: python scripts/graphed_matrix.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dicp_amd import _lib
import dicp_amd._loop as L
from dicp_amd.ICP import ICP
from dicp_amd.graphed import graphed_icp_step
from dicp_amd.synthetic import make_pairs, make_scene_pairs
bad = 0
CASES = [("pt2pl f32", dict()), ("pt2pl f32 deterministic", dict(det=True)), ("pt2pt f32", dict(typ="pt2pt")), ("pt2pl f64", dict(dt=torch.float64)),
         ("pt2pl f32 certificates at every size", dict(certs=True)), ("scene f32", dict(gen="scene")), ("pt2pl f32 brute force", dict(knn=_lib.KNN_VALU)),
         ("pt2pl f32 no tail", dict(tune={"bwd_tail": False})), ("pt2pl f32 cauchy no trim", dict(loss={"name": "cauchy", "metric": 0.5}, trim=None))]
for name, c in CASES:
    for B, n, K in ((8, 8192, 6), (24, 16384, 10)):
        dt = c.get("dt", torch.float32)
        src, tgt = (make_scene_pairs if c.get("gen") == "scene" else make_pairs)(B, n, n, seed=11, dtype=dt)
        if c.get("typ") == "pt2pt":
            tgt = tgt[:, :, :3].contiguous()
        src, tgt = src.cuda(), tgt.cuda()
        T0 = torch.eye(4, device="cuda", dtype=dt).repeat(B, 1, 1)
        kw = dict(trim_dist=c.get("trim", 5.0), loss_fn=c.get("loss", {"name": "huber", "metric": 1.0}))
        icp = ICP(icp_type=c.get("typ", "pt2pl"), differentiable=True, max_iterations=K, tolerance=1e-12); icp.const_iter = True
        icp.deterministic = bool(c.get("det"))
        if "knn" in c:
            icp.knn_variant = c["knn"]
        icp._tuning.update(c.get("tune", {}))
        old = L.CERT_MIN_WORK
        if c.get("certs"):
            L.CERT_MIN_WORK = 0.0
        try:
            a, b = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
            for _ in range(3):
                a.grad = b.grad = None
                o = icp.icp(a, b, T0, **kw); (o["T"].sum() + 1e-3 * (o["pc"] ** 2).sum()).backward()
            gs_e, gt_e, T_e = a.grad.clone(), b.grad.clone(), o["T"].detach().clone()
            s, t = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True)
            step = graphed_icp_step(icp, lambda o_: o_["T"].sum() + 1e-3 * (o_["pc"] ** 2).sum(), s, t, T0, num_warmup_iters=3, **kw)
            worst = 0.0
            same_T = True
            for i in range(6):
                out, grads = step(s, t, T0)
                torch.cuda.synchronize()
                junk = torch.full((1 << 21,), 2.0, device="cuda").sum() + (grads["target"] * 2).sum()
                torch.cuda.synchronize()
                same_T = same_T and torch.equal(out["T"], T_e)
                for g, e in ((grads["source"], gs_e), (grads["target"], gt_e)):
                    worst = max(worst, float((g - e).abs().max() / e.abs().max()))
            step.check_errors()
            ok = same_T and worst <= (1e-9 if dt == torch.float64 else 2e-5) and (not c.get("det") or worst == 0.0)
        finally:
            L.CERT_MIN_WORK = old
        bad += 0 if ok else 1
        print("%-40s %3d x %5d K=%2d: T identical %s, worst gradient difference %.1e   %s" % (name, B, n, K, same_T, worst, "ok" if ok else "FAILED"), flush=True)
print("graphed_matrix:", "ok" if bad == 0 else "%d FAILED" % bad)
